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

# emissive Neumann shell around a Dirichlet ball: the Neumann sampling sweeps runs of consecutive triangle indices
from test_gpu_3d import _shell_scene
for subdiv in (3, 4):
    sd = _shell_scene(2, subdiv, flux=lambda x, y, z: 0.3 * y)
    it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((256, 256), 16, 64, 2e-3))
    it.solve()
    it.solve()
    st = it.last_stats
    print("emissive shell %d triangles, 256x256, 16 spp: %.3g walk steps, kernel %.1f ms -> %.3g steps/s" % (len(sd["n_tris"]), st["walk_steps"], st["kernel_ms"], st["walk_steps"] / (st["kernel_ms"] * 1e-3)), flush=True)
    it.close()
