"""emissive / zero-flux Neumann shell around a Dirichlet ball: where the 3-D step time goes (developer scratch)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from test_gpu_3d import _shell_scene
from elaina_amd.integrator3d import Problem3, UniformIntegrator3
from elaina_amd import UniformIntegratorSettings
for subdiv in (2, 3):
    for name, flux in (("zero-flux", None), ("emissive", lambda x, y, z: 0.3 * y)):
        sd = _shell_scene(2, subdiv, flux=flux)
        it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((128, 128), 16, 64, 2e-3))
        it.solve()
        it.solve()
        st = it.last_stats
        print("%s shell %d triangles, 128x128, 16 spp: %.3g walk steps, kernel %.1f ms -> %.3g steps/s  %s" % (name, len(sd["n_tris"]), st["walk_steps"], st["kernel_ms"], st["walk_steps"] / (st["kernel_ms"] * 1e-3), {k: v for k, v in st.items() if k not in ("walk_steps", "kernel_ms")}), flush=True)
        it.close()
