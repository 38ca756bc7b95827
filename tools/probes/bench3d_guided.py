"""developer probe: GuidedIntegrator<3> on the 3-D bench scenes (Dirichlet icosphere, Neumann shell around a Dirichlet ball), FRAME^2,
SPP samples, half of them trained: walk-steps/s of the whole solve, beside the uniform integrator on the same scene"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from conftest import sphere_scene3
from test_gpu_3d import _shell_scene
from elaina_amd import UniformIntegratorSettings
from elaina_amd.guided import GuidedIntegratorSettings
from elaina_amd.integrator3d import GuidedIntegrator3, Problem3, UniformIntegrator3, default_net_config3
frame, spp = int(os.environ.get("FRAME", "256")), int(os.environ.get("SPP", "16"))
scenes = {"dirichlet_icosphere_1280": sphere_scene3(subdiv=3, radius=1.0, value=lambda x, y, z: x * y), "neumann_shell_1280": _shell_scene(2, 3)}
for name, sd in scenes.items():
    p = Problem3.from_dict(sd)
    it = UniformIntegrator3(p, UniformIntegratorSettings((frame, frame), spp, 64, 2e-3))
    it.solve(); it.solve()
    u = it.last_stats
    print("%-26s uniform: %.4g steps, kernel %.1f ms -> %.3g steps/s" % (name, u["walk_steps"], u["kernel_ms"], u["walk_steps"] / (u["kernel_ms"] * 1e-3)), flush=True)
    it.close()
    st = GuidedIntegratorSettings(frameSize=(frame, frame), samplesPerPixel=spp, trainSppCount=spp // 2, maxWalkingDepth=64, epsilonShell=2e-3)
    gi = GuidedIntegrator3(p, st, ((-1.1, -1.1, -1.1), (1.1, 1.1, 1.1)), network_config=default_net_config3(), seed=7)
    for _ in range(2):
        t0 = time.perf_counter()
        gi.solve()
        dt = time.perf_counter() - t0
    g = gi.last_stats
    print("%-26s guided : %.4g steps (%d guided, %d Adam steps), %.1f ms -> %.3g steps/s" % (
        name, g["walk_steps"], g["guided_steps"], g["optimizer_steps"], dt * 1e3, g["walk_steps"] / dt), flush=True)
    gi.close()
