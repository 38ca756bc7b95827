"""GuidedIntegrator<3> on the two bench scenes by frame size, one launch per sample against the launches per depth (WOST3_G_FUSED = 1 / 0):
where the default should switch.  ms per solve of 8 samples (4 trained), second of two solves."""
import os, sys, time, json
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import bench
from elaina_amd.guided import GuidedIntegratorSettings
from elaina_amd.integrator3d import GuidedIntegrator3, Problem3, default_net_config3

V, T = bench.icosphere(3, 1.0)
col = np.repeat((V[:, 0] * V[:, 1] + V[:, 2]).astype(np.float32)[:, None], 6, axis=1)
ball = {"d_verts": V, "d_tris": T, "d_colors": col, "n_verts": None, "n_tris": None, "n_colors": None,
        "probe": (0.6, (0.0, 0.0, 0.1), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)), "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
Vi, Ti = bench.icosphere(2, 0.45)
shell = {"d_verts": Vi, "d_tris": Ti, "d_colors": np.repeat(Vi[:, :1], 6, axis=1).astype(np.float32), "n_verts": V, "n_tris": T,
         "n_colors": np.zeros((len(V), 6), np.float32), "probe": (0.7, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)),
         "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
for frame in [int(f) for f in os.environ.get("G3_FRAMES", "256 362 512 724 1024").split()]:
    row = {}
    for name, sd in (("icosphere", ball), ("shell", shell)):
        for form in os.environ.get("G3_FORMS", "1 0").split():
            os.environ["WOST3_G_FUSED"] = form
            st = GuidedIntegratorSettings(frameSize=(frame, frame), samplesPerPixel=8, trainSppCount=4, maxWalkingDepth=64, epsilonShell=2e-3)
            gi = GuidedIntegrator3(Problem3.from_dict(sd), st, ((-1.1, -1.1, -1.1), (1.1, 1.1, 1.1)), network_config=default_net_config3(), seed=7)
            for _ in range(2):
                t0 = time.perf_counter()
                gi.solve()
                dt = time.perf_counter() - t0
            row["%s %s" % (name, "fused" if form == "1" else "per depth")] = round(dt * 1e3, 1)
            gi.close()
    print(frame, json.dumps(row), flush=True)
