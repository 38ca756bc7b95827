#!/usr/bin/env python3
"""Developer check on a GPU box: parity of the HIP path against the oracle + quick timing."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import elaina_amd
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
from oracle.oracle import Oracle


def main():
    spp_big = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    o = Oracle()
    p = Problem.load_scene("ladybug")
    sd = p.as_dict()
    # 1. closest point parity on a grid
    it = UniformIntegrator(p, UniformIntegratorSettings((128, 128), 16, 32, 1.0))
    rng = np.random.default_rng(1)
    pts = np.concatenate([rng.uniform(-90, 590, size=(200000, 2)),
                          p.d_verts[rng.integers(0, len(p.d_verts), 100000)] + rng.normal(0, 0.5, size=(100000, 2))])
    pts = pts.astype(np.float32)
    t = time.time(); gi, gd, gu, gs = it.closest_point(pts); tg = time.time() - t
    t = time.time(); oi, od, ou, os_ = o.closest_point(p.d_verts, p.d_segs, pts, mode=1); to = time.time() - t
    print("closest_point: idx equal %s dist equal %s uv equal %s side equal %s (gpu %.3fs oracle %.3fs)" % (
        np.array_equal(gi, oi), np.array_equal(gd, od), np.array_equal(gu, ou), np.array_equal(gs, os_), tg, to))
    if not np.array_equal(gi, oi):
        bad = np.nonzero(gi != oi)[0]
        print("  mismatches", len(bad), bad[:5], gi[bad[:5]], oi[bad[:5]], gd[bad[:5]], od[bad[:5]])
    # 2. config 1 parity
    ms = it.solve()
    ref = o.solve(sd, 128, 128, 16, 32, 1.0, threads=os.cpu_count())
    f = it.solution
    diff = np.abs(f - ref["field"])
    rel = np.linalg.norm(f - ref["field"]) / np.linalg.norm(ref["field"])
    print("cfg1: gpu %d ms, stats %s" % (ms, it.last_stats))
    print("cfg1: oracle steps %d (%.2fs, %d thr) bit-exact %s rel-L2 %.3e maxabs %.3e steps equal %s" % (
        ref["walk_steps"], ref["seconds"], os.cpu_count(), np.array_equal(f, ref["field"]), rel, diff.max(),
        ref["walk_steps"] == it.last_stats["walk_steps"]))
    for k in ("walks_started", "walks_absorbed", "walks_truncated", "neumann_hits"):
        print("   %s gpu %d oracle %d" % (k, it.last_stats[k], ref[k]))
    it.close()
    # 3. timing at 1024^2
    for spr in (32, 64, 128):
        it = UniformIntegrator(p, UniformIntegratorSettings((1024, 1024), spp_big, 64, 1.0))
        it.set_option("steps_per_round", spr)
        it.solve()  # warm
        ms = it.solve()
        s = it.last_stats
        print("1024^2 spp=%d steps_per_round=%d: wall %.1f ms kernel %.1f ms launches %d steps %d -> %.3e steps/s (kernel %.3e)" % (
            spp_big, spr, s["solve_ms"], s["kernel_ms"], s["kernel_launches"], s["walk_steps"],
            s["walk_steps"] / s["solve_ms"] * 1e3, s["walk_steps"] / s["kernel_ms"] * 1e3))
        it.close()


if __name__ == "__main__":
    main()
