"""walk-steps/s of the 3-D uniform integrator on an icosphere (developer scratch)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from conftest import sphere_scene3
from elaina_amd.integrator3d import Problem3, UniformIntegrator3
from elaina_amd import UniformIntegratorSettings
for subdiv, frame, spp in ((3, 512, 64), (5, 512, 64), (5, 1024, 64)):
    sd = sphere_scene3(subdiv=subdiv, value=lambda x, y, z: x * y + z)
    sd["probe"] = (0.6, (0.0, 0.0, 0.1), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0))
    it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((frame, frame), spp, 128, 2e-3))
    it.solve()
    t0 = time.perf_counter(); it.solve(); dt = time.perf_counter() - t0
    st = it.last_stats
    print("icosphere %d triangles, %dx%d, %d spp: %.3g walk steps in %.1f ms (kernel %.1f ms) -> %.3g steps/s" % (len(sd["d_tris"]), frame, frame, spp, st["walk_steps"], dt * 1e3, st["kernel_ms"], st["walk_steps"] / (st["kernel_ms"] * 1e-3)), flush=True)
    it.close()
