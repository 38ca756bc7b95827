"""rates of the batch queries on a Neumann shell (developer scratch)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from test_gpu_3d import _shell_scene
from elaina_amd.integrator3d import Problem3, UniformIntegrator3
from elaina_amd import UniformIntegratorSettings
for subdiv in (2, 3):
    sd = _shell_scene(2, subdiv)
    it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((8, 8), 1, 4, 2e-3))
    rng = np.random.default_rng(1)
    n = 1 << 20
    pts = rng.uniform(-0.7, 0.7, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = np.full(n, 0.5, np.float32)
    rmax = np.full(n, 0.5, np.float32)
    for name, f in (("closest_point", lambda: it.closest_point(pts)), ("silhouette", lambda: it.closest_silhouette(pts, rmax)), ("ray", lambda: it.ray_intersect(pts, d, tmax))):
        f()
        t0 = time.perf_counter(); f(); dt = time.perf_counter() - t0
        print("%d triangles: %s %d queries in %.1f ms" % (len(sd["n_tris"]), name, n, dt * 1e3), flush=True)
    it.close()
