#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ from the CPU oracle.

The reference ships no golden outputs for this path (no integrator,
queue, RNG or lbvh tests; `gt.json` is a config, not a result), it cannot be built here
(CUDA only, every submodule empty) and it is not Python, so these vectors come from the
oracle restatement (oracle/wost_oracle.c) -- they pin the oracle AND the HIP path against
regressions, they are not reference outputs.  Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from elaina_amd import Problem  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def main():
    o = Oracle()
    out = {}
    for scene, depth in (("ladybug", 32), ("fille", 32)):
        p = Problem.load_scene(scene)
        # BASELINE.json configs[0]: 128^2 grid, 16 spp, max_depth 32, eps 1
        r = o.solve(p.as_dict(), 128, 128, 16, depth, 1.0, threads=os.cpu_count(), want_steps=True, want_hist=True)
        out[scene + "_cfg1_field"] = r["field"]
        out[scene + "_cfg1_steps"] = r["steps"]
        out[scene + "_cfg1_hist"] = r["depth_hist"]
        out[scene + "_cfg1_counts"] = np.array(
            [r["walk_steps"], r["walks_started"], r["walks_absorbed"], r["walks_truncated"], r["neumann_hits"]],
            dtype=np.uint64)
        rng = np.random.default_rng(11)
        pts = np.concatenate([rng.uniform(-90, 590, size=(2000, 2)),
                              p.d_verts[rng.integers(0, len(p.d_verts), 2000)] + rng.normal(0, 0.5, (2000, 2))])
        pts = pts.astype(np.float32)
        idx, dist, uv, side = o.closest_point(p.d_verts, p.d_segs, pts, mode=0)  # brute force
        out[scene + "_cp_pts"] = pts
        out[scene + "_cp_idx"] = idx
        out[scene + "_cp_dist"] = dist
        out[scene + "_cp_uv"] = uv
        out[scene + "_cp_side"] = side.astype(np.int8)
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_golden.npz"))


if __name__ == "__main__":
    main()
