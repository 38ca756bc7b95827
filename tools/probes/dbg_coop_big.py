import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from conftest import wiggly_problem
from elaina_amd import UniformIntegrator, UniformIntegratorSettings
from oracle.oracle import Oracle
orc = Oracle()
for nn in (3000, 30000):
    p = wiggly_problem(nn, 64)
    it = UniformIntegrator(p, UniformIntegratorSettings((16, 16), 1, 4, 1.0))
    rng = np.random.default_rng(9)
    n = 200000
    V, S = p.n_verts, p.n_segs
    si = rng.integers(0, len(S), n)
    t = rng.uniform(0, 1, (n, 1)).astype(np.float32)
    a, b = V[S[si, 0]], V[S[si, 1]]
    on = (a + (b - a) * t).astype(np.float32)
    e = b - a
    nrm = np.stack([e[:, 1], -e[:, 0]], 1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    off = rng.choice([0.0, 0.05, -0.05, 1e-3, -1e-3, 0.3], n)[:, None].astype(np.float32)
    pts = (on + off * nrm).astype(np.float32)
    for rm in (None, 0.2, 2.0):
        rmax = None if rm is None else np.full(n, rm, np.float32)
        got, ref = it.closest_silhouette(pts, rmax), orc.closest_silhouette(V, S, pts, rmax)
        bad = np.flatnonzero(got != ref)
        print(nn, "silhouette rmax", rm, "mismatches", len(bad), [(pts[i].tolist(), float(got[i]), float(ref[i])) for i in bad[:2]], flush=True)
    ang = rng.uniform(0, 2 * np.pi, size=n)
    d = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)
    for tm in (0.03, 0.5, 10.0):
        tmax = (rng.uniform(0.5, 1.0, n) * tm).astype(np.float32)
        gh, gt, gi = it.ray_intersect(pts, d, tmax)
        rh, rt, ri = orc.ray_intersect(V, S, pts, d, tmax)
        bad = np.flatnonzero((gh != rh) | ((rh == 1) & ((gt != rt) | (gi != ri))))
        print(nn, "ray tmax", tm, "mismatches", len(bad), [(pts[i].tolist(), d[i].tolist(), float(tmax[i]), int(gh[i]), float(gt[i]), int(gi[i]), int(rh[i]), float(rt[i]), int(ri[i])) for i in bad[:3]], flush=True)
    it.close()
