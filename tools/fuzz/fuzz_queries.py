"""closest-point / silhouette / ray queries of a fuzz scene: HIP against the oracle's brute-force loops and its BVH
(which side is wrong when a fuzz solve differs)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.dirname(__file__))
from fuzz_parity import random_problem
from elaina_amd import UniformIntegrator, UniformIntegratorSettings
from oracle.oracle import Oracle
o = Oracle()
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(seed)
    p, scale, feat = random_problem(rng)
    it = UniformIntegrator(p, UniformIntegratorSettings((8, 8), 1, 4, scale * 1e-3))
    q = np.random.default_rng(1000 + seed)
    for spread in (1.0, 3.0, 50.0, 1e4):
        pts = (q.uniform(-1, 1, (20000, 2)) * scale * spread).astype(np.float32)
        gi, gd, gu, gs = it.closest_point(pts)
        bi, bd, bu, bs = o.closest_point(p.d_verts, p.d_segs, pts, mode=0)
        vi, vd, vu, vs = o.closest_point(p.d_verts, p.d_segs, pts, mode=1)
        print("seed %d spread %g: closest point  HIP!=brute %d (dist %d)  oracleBVH!=brute %d (dist %d)" % (
            seed, spread, int((gi != bi).sum()), int((gd != bd).sum()), int((vi != bi).sum()), int((vd != bd).sum())), flush=True)
        if p.n_segs is not None and len(p.n_segs):
            rmax = (q.uniform(0.01, 2.0, len(pts)) * scale * spread).astype(np.float32)
            gs_ = it.closest_silhouette(pts, rmax)
            os_ = o.closest_silhouette(p.n_verts, p.n_segs, pts, rmax)
            d = q.normal(size=(len(pts), 2)).astype(np.float32)
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            gh, gt, gi2 = it.ray_intersect(pts, d, rmax)
            oh, ot, oi2 = o.ray_intersect(p.n_verts, p.n_segs, pts, d, rmax)
            hit = oh != 0
            print("   silhouette differ %d   ray hit flag differ %d, t differ %d, idx differ %d (of %d hits)" % (
                int((gs_ != os_).sum()), int((gh != oh).sum()), int((gt[hit] != ot[hit]).sum()), int((gi2[hit] != oi2[hit]).sum()), int(hit.sum())), flush=True)
    it.close()
