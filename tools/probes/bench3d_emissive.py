"""Neumann shell around a Dirichlet ball, zero-flux against emissive (flux 0.3 y): what the emissive paths of walk3_kernel (the sample
on the boundary inside the ball, its shadow ray) cost -- VERDICT r3 item 5's ratio.  FRAMES="512:16 1024:4" (frame:spp)"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from test_gpu_3d import _shell_scene
from elaina_amd.integrator3d import Problem3, UniformIntegrator3
from elaina_amd import UniformIntegratorSettings
subdiv = int(os.environ.get("SUBDIV", "3"))
for fs in os.environ.get("FRAMES", "512:16 1024:4").split():
    frame, spp = (int(x) for x in fs.split(":"))
    rate = {}
    for name, flux in (("zero-flux", None), ("emissive", lambda x, y, z: 0.3 * y)):
        sd = _shell_scene(2, subdiv, flux=flux)
        it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((frame, frame), spp, 64, 2e-3))
        it.solve()
        it.solve()
        st = it.last_stats
        rate[name] = st["walk_steps"] / (st["kernel_ms"] * 1e-3)
        print("%s shell %d triangles, %dx%d, %d spp: %.3g walk steps, kernel %.1f ms -> %.3g steps/s" % (
            name, len(sd["n_tris"]), frame, frame, spp, st["walk_steps"], st["kernel_ms"], rate[name]), flush=True)
        it.close()
    print("%dx%d: zero-flux / emissive = %.2f" % (frame, frame, rate["zero-flux"] / rate["emissive"]), flush=True)
