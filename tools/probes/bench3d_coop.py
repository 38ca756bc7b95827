"""developer probe: Neumann shell around a Dirichlet ball (the bench scene): the Neumann-side tree queries per lane (WOST3_COOP=0)
against answered by the wave through its task pools (closest_silhouette3_wave / ray_closest3_wave); same field required"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from test_gpu_3d import _shell_scene
from elaina_amd.integrator3d import Problem3, UniformIntegrator3
from elaina_amd import UniformIntegratorSettings
subdiv, frame, spp, depth = int(os.environ.get("SUBDIV", "3")), int(os.environ.get("FRAME", "512")), int(os.environ.get("SPP", "16")), 64
specs = sys.argv[1:] or ["WOST3_COOP=0", "WOST3_COOP=1", "WOST3_COOP=1,WOST3_WAIT_WEIGHT=2", "WOST3_COOP=1,WOST3_WAIT_WEIGHT=4", "WOST3_COOP=1,WOST3_WAIT_WEIGHT=8",
                         "WOST3_COOP=1,WOST3_POOL_CAP=512", "WOST3_COOP=1,WOST3_RAY_TRIGGER=64", "WOST3_COOP=1,WOST3_RAY_TRIGGER=16"]
for flux in (None, lambda x, y, z: 0.3 * y):
    sd = _shell_scene(2, subdiv, flux=flux)
    ref = None
    for spec in specs:
        for k in ("WOST3_COOP", "WOST3_WAIT_WEIGHT", "WOST3_POOL_CAP", "WOST3_RAY_TRIGGER", "WOST3_BLOCKS_PER_CU", "WOST3_TRAV_BURST"):
            os.environ.pop(k, None)
        for kv in spec.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
        it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((frame, frame), spp, depth, 2e-3))
        it.solve()
        it.solve()
        st = it.last_stats
        f = it.solution.copy()
        if ref is None:
            ref = (f, st["walk_steps"])
        print("%-44s %s shell %d triangles %dx%d %d spp: %.4g steps, kernel %.1f ms -> %.3g steps/s, same field %s same steps %s" % (
            spec, "emissive" if flux else "zero-flux", len(sd["n_tris"]), frame, frame, spp, st["walk_steps"], st["kernel_ms"],
            st["walk_steps"] / (st["kernel_ms"] * 1e-3), np.array_equal(ref[0], f), ref[1] == st["walk_steps"]), flush=True)
        it.close()
