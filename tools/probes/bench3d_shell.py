"""Neumann shell around a Dirichlet ball: scheduler constants of the 3-D walk kernel with tree queries in the step (developer scratch)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from test_gpu_3d import _shell_scene
from elaina_amd.integrator3d import Problem3, UniformIntegrator3
from elaina_amd import UniformIntegratorSettings
subdiv, frame, spp, depth = int(os.environ.get("SUBDIV", "3")), int(os.environ.get("FRAME", "512")), 16, 64
for flux in (None, lambda x, y, z: 0.3 * y):
    sd = _shell_scene(2, subdiv, flux=flux)
    for ww, tb, pc in ((32, 3, 4), (8, 3, 4), (2, 3, 4), (1, 3, 4), (2, 8, 4), (2, 3, 2), (2, 3, 3)):
        os.environ["WOST3_WAIT_WEIGHT"] = str(ww)
        os.environ["WOST3_TRAV_BURST"] = str(tb)
        os.environ["WOST3_BLOCKS_PER_CU"] = str(pc)
        it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((frame, frame), spp, depth, 2e-3))
        it.solve()
        it.solve()
        st = it.last_stats
        print("ww %d tb %d per_cu %d: %s shell %d triangles, %dx%d, %d spp depth %d: %.3g walk steps, kernel %.1f ms -> %.3g steps/s" % (ww, tb, pc, "emissive" if flux else "zero-flux", len(sd["n_tris"]), frame, frame, spp, depth, st["walk_steps"], st["kernel_ms"], st["walk_steps"] / (st["kernel_ms"] * 1e-3)), flush=True)
        it.close()
