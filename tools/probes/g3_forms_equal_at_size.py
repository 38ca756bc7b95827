"""GuidedIntegrator<3>, one launch per sample against the launches per depth on a frame that fills the chip (the live lists hold hundreds of
thousands of walkers, written by thousands of blocks): fields, counters and trained parameters must be equal bit for bit.  FRAME (724), SPP (4)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import bench
from elaina_amd.guided import GuidedIntegratorSettings
from elaina_amd.integrator3d import GuidedIntegrator3, Problem3, default_net_config3

frame, spp = int(os.environ.get("FRAME", "724")), int(os.environ.get("SPP", "4"))
V, T = bench.icosphere(3, 1.0)
Vi, Ti = bench.icosphere(2, 0.45)
shell = {"d_verts": Vi, "d_tris": Ti, "d_colors": np.repeat(Vi[:, :1], 6, axis=1).astype(np.float32), "n_verts": V, "n_tris": T,
         "n_colors": np.zeros((len(V), 6), np.float32), "probe": (0.7, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)),
         "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
res = {}
for form in ("1", "0"):
    os.environ["WOST3_G_FUSED"] = form
    st = GuidedIntegratorSettings(frameSize=(frame, frame), samplesPerPixel=spp, trainSppCount=spp // 2, maxWalkingDepth=64, epsilonShell=2e-3)
    gi = GuidedIntegrator3(Problem3.from_dict(shell), st, ((-1.1, -1.1, -1.1), (1.1, 1.1, 1.1)), network_config=default_net_config3(), seed=7)
    gi.solve()
    res[form] = (gi.solution.copy(), dict(gi.last_stats), gi.network.params().copy())
    gi.close()
a, b = res["1"], res["0"]
keys = ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated", "neumann_hits", "guided_steps", "train_samples", "optimizer_steps")
print("frame %d spp %d: fields equal %s, parameters equal %s, counters %s" % (
    frame, spp, np.array_equal(a[0], b[0]), np.array_equal(a[2], b[2]), {k: (a[1][k], b[1][k]) for k in keys if a[1][k] != b[1][k]} or "equal"))
print("launches fused %d, per depth %d; walk steps %d" % (a[1]["kernel_launches"], b[1]["kernel_launches"], a[1]["walk_steps"]))
