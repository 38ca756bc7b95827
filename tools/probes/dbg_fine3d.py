import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from elaina_amd import UniformIntegratorSettings
from elaina_amd.integrator3d import Problem3, UniformIntegrator3
from oracle.oracle import Oracle
orc = Oracle()
for subdiv, centre in ((5, (0.0, 0.0, 0.0)), (5, (100.0, 50.0, -70.0))):
    V, T = bench.icosphere(subdiv, 1.0)
    Vi, Ti = bench.icosphere(2, 0.45)
    c = np.asarray(centre, np.float32)
    V = (V + c).astype(np.float32); Vi = (Vi + c).astype(np.float32)
    sd = {"d_verts": Vi, "d_tris": Ti, "d_colors": np.repeat(Vi[:, :1], 6, axis=1).astype(np.float32), "n_verts": V, "n_tris": T,
          "n_colors": np.zeros((len(V), 6), np.float32), "probe": (0.7, tuple(float(x) for x in c), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)),
          "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
    w, h, spp, depth, eps = 24, 24, 3, 64, 2e-3
    it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((w, h), spp, depth, eps))
    it.solve()
    ref = orc.solve3(sd, w, h, spp, depth, eps, threads=os.cpu_count() or 8)
    s = it.last_stats
    print(len(T), centre, "steps", s["walk_steps"], ref["walk_steps"], "hits", s["neumann_hits"], ref["neumann_hits"], "field equal", np.array_equal(it.solution.reshape(-1, 3), ref["field"]), flush=True)
    rng = np.random.default_rng(3)
    n = 20000
    ti = rng.integers(0, len(T), n)
    bc = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    on = (V[T[ti, 0]] * bc[:, :1] + V[T[ti, 1]] * bc[:, 1:2] + V[T[ti, 2]] * bc[:, 2:3]).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = (rng.uniform(0.5, 1.0, n) * rng.choice([0.01, 0.3, 3.0], n)).astype(np.float32)
    got, rf = it.ray_intersect(on, d, tmax), orc.ray_intersect3(V, T, on, d, tmax)
    hit = rf[0] != 0
    print("   rays from the surface: hit flag differ", int((got[0] != rf[0]).sum()), "t differ", int((got[1][hit] != rf[1][hit]).sum()), "idx differ", int((got[2][hit] != rf[2][hit]).sum()), "of", int(hit.sum()), flush=True)
    print("   silhouette differ", int((it.closest_silhouette(on, np.full(n, 0.3, np.float32)) != orc.closest_silhouette3(V, T, on, np.full(n, 0.3, np.float32))).sum()), flush=True)
    it.close()
