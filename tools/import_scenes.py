#!/usr/bin/env python3
"""Turn the reference's shipped 2-D scenes into compact binary fixtures.

Reads  /root/reference/data/<scene>/{model.obj,boundary.obj,u.json}  (geometry and
settings are DATA the reference ships for this path; `color.json`, the Dirichlet
boundary values, is a missing blob -- /root/reference/.MISSING_LARGE_BLOBS -- and
is synthesized here) and writes

    data/scenes/<scene>.npz   with
        d_verts  float32 [nv,2]   Dirichlet polyline vertices (OBJ x,y)
        d_segs   int32   [ns,2]   0-based vertex indices per segment
        d_colors float32 [nv,6]   synthesized (left rgb, right rgb) per vertex
        n_verts / n_segs          Neumann box (boundary.obj)
        probe    float32 [5]      scale, pos.x, pos.y, up.x, up.y
        aabb     float32 [4]      min.x, min.y, max.x, max.y
        settings int32   [2]      maxWalkingDepth, frame (W == H)
        eps      float32 [1]      epsilonShell

Colour synthesis (deterministic, seed 42): vertices are grouped into polylines
(maximal chains of segments i -> i+1 sharing a vertex); polyline k gets
left = hash01(42, k, 0..2), right = hash01(42, k, 3..5), constant along the chain.

This script only runs in the build container (it needs /root/reference); the
.npz files it writes are committed.  The SHA-256 of each output is printed and
recorded in data/scenes/SHA256SUMS.
"""
import hashlib
import json
import os
import sys

import numpy as np

REF = "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data", "scenes")


def parse_obj(path):
    verts, segs = [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append((float(t[1]), float(t[2])))
            elif t[0] == "l":
                idx = [int(x) - 1 for x in t[1:]]
                for a, b in zip(idx[:-1], idx[1:]):
                    segs.append((a, b))
    return np.asarray(verts, dtype=np.float64).astype(np.float32), np.asarray(segs, dtype=np.int32)


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def hash01(seed, k, c):
    h = splitmix64(splitmix64(splitmix64(seed) ^ k) ^ c)
    return np.float32((h >> 40) / float(1 << 24))


def polyline_ids(n_verts, segs):
    """label vertices by connected chain (union-find over segments)"""
    parent = list(range(n_verts))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for a, b in segs:
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    roots = [find(v) for v in range(n_verts)]
    order = {}
    ids = np.zeros(n_verts, dtype=np.int64)
    for v, r in enumerate(roots):
        if r not in order:
            order[r] = len(order)
        ids[v] = order[r]
    return ids, len(order)


def synth_colors(n_verts, segs, seed=42):
    ids, n_poly = polyline_ids(n_verts, segs)
    table = np.zeros((n_poly, 6), dtype=np.float32)
    for k in range(n_poly):
        for c in range(6):
            table[k, c] = hash01(seed, k, c)
    return table[ids], n_poly


def main():
    os.makedirs(OUT, exist_ok=True)
    sums = []
    for scene in ("ladybug", "fille"):
        dv, ds = parse_obj(os.path.join(REF, scene, "model.obj"))
        nv, ns = parse_obj(os.path.join(REF, scene, "boundary.obj"))
        conf = json.load(open(os.path.join(REF, scene, "u.json")))
        probe = conf["scene"]["evaluation_grid"]["mData"]
        aabb = conf["scene"]["aabb"]
        setting = conf["integrator"]["setting"]
        colors, n_poly = synth_colors(len(dv), ds)
        path = os.path.join(OUT, scene + ".npz")
        np.savez_compressed(
            path,
            d_verts=dv, d_segs=ds, d_colors=colors, n_verts=nv, n_segs=ns,
            probe=np.asarray([probe["scale"], probe["pos"][0], probe["pos"][1], probe["up"][0], probe["up"][1]],
                             dtype=np.float32),
            aabb=np.asarray(aabb["min"] + aabb["max"], dtype=np.float32),
            settings=np.asarray([setting["maxWalkingDepth"], setting["frameSize"][0]], dtype=np.int32),
            eps=np.asarray([setting["epsilonShell"]], dtype=np.float32),
        )
        # content hash over the arrays (npz container bytes depend on zlib build)
        h = hashlib.sha256()
        for a in (dv, ds, colors, nv, ns):
            h.update(np.ascontiguousarray(a).tobytes())
        sums.append("%s  %s (arrays)" % (h.hexdigest(), scene))
        print(scene, "dirichlet", dv.shape, ds.shape, "polylines", n_poly, "neumann", nv.shape, ns.shape,
              h.hexdigest())
    with open(os.path.join(OUT, "SHA256SUMS"), "w") as f:
        f.write("\n".join(sums) + "\n")


if __name__ == "__main__":
    sys.exit(main())
