"""developer probe: fine 2-D meshes far from the origin against the flat loops of the oracle -- closest point on a 40 000-segment Dirichlet
circle, an emissive 30 000-segment Neumann curve (sampling sweeps, shadow rays), small solves"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from conftest import wiggly_problem
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
from oracle.oracle import Oracle
orc = Oracle()
for off in ((0.0, 0.0), (800.0, -300.0)):
    p = wiggly_problem(3000, 40000)
    o = np.asarray(off, np.float32)
    p = Problem(d_verts=(p.d_verts + o).astype(np.float32), d_segs=p.d_segs, d_colors=p.d_colors, n_verts=(p.n_verts + o).astype(np.float32), n_segs=p.n_segs,
                n_colors=None, probe=(110.0, float(o[0]), float(o[1]), 0.0, 1.0))
    it = UniformIntegrator(p, UniformIntegratorSettings((16, 16), 1, 4, 1.0))
    rng = np.random.default_rng(5)
    n = 60000
    ang = rng.uniform(0, 2 * np.pi, n)
    r = 15.0 + rng.choice([0.0, 1e-3, -1e-3, 0.05, 3.0, 60.0], n) * rng.uniform(0.5, 1.0, n)
    pts = (np.stack([r * np.cos(ang) + 5.0, r * np.sin(ang) - 3.0], 1) + o).astype(np.float32)
    gi, gd, gu, gs = it.closest_point(pts)
    bi, bd, bu, bs = orc.closest_point(p.d_verts, p.d_segs, pts, mode=0)
    print(off, "closest point: idx differ", int((gi != bi).sum()), "dist differ", int((gd != bd).sum()), "uv differ", int((gu != bu).sum()), "side differ", int((gs != bs).sum()), flush=True)
    it.close()
    for spec in ({}, {"coop": 0}):
        it = UniformIntegrator(p, UniformIntegratorSettings((40, 40), 3, 64, 0.05))
        for k, v in spec.items():
            it.set_option(k, v)
        it.solve()
        ref = orc.solve(p.as_dict(), 40, 40, 3, 64, 0.05, threads=os.cpu_count() or 8)
        print(off, spec, "solve: steps", it.last_stats["walk_steps"], ref["walk_steps"], "field equal", np.array_equal(it.solution, ref["field"]), flush=True)
        it.close()
    pe = wiggly_problem(30000, 400, emissive=True)
    pe = Problem(d_verts=(pe.d_verts + o).astype(np.float32), d_segs=pe.d_segs, d_colors=pe.d_colors, n_verts=(pe.n_verts + o).astype(np.float32), n_segs=pe.n_segs,
                 n_colors=pe.n_colors, probe=(110.0, float(o[0]), float(o[1]), 0.0, 1.0))
    it = UniformIntegrator(pe, UniformIntegratorSettings((32, 32), 3, 48, 0.05))
    it.solve()
    ref = orc.solve(pe.as_dict(), 32, 32, 3, 48, 0.05, threads=os.cpu_count() or 8)
    print(off, "emissive 30000: steps", it.last_stats["walk_steps"], ref["walk_steps"], "field equal", np.array_equal(it.solution, ref["field"]),
          "max diff", float(np.abs(it.solution - ref["field"]).max()), flush=True)
    it.close()
