"""create / solve / destroy GuidedIntegrator<3> 150 times (fused and per-depth forms alternating, training on): device memory in use must not grow"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import torch
import bench
from elaina_amd.guided import GuidedIntegratorSettings
from elaina_amd.integrator3d import GuidedIntegrator3, Problem3, default_net_config3

V, T = bench.icosphere(2, 1.0)
col = np.repeat((V[:, 0] * V[:, 1] + V[:, 2]).astype(np.float32)[:, None], 6, axis=1)
ball = {"d_verts": V, "d_tris": T, "d_colors": col, "n_verts": None, "n_tris": None, "n_colors": None,
        "probe": (0.6, (0.0, 0.0, 0.1), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)), "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
used = []
for it in range(150):
    os.environ["WOST3_G_FUSED"] = str(it & 1)
    st = GuidedIntegratorSettings(frameSize=(64, 48), samplesPerPixel=3, trainSppCount=2, maxWalkingDepth=32, epsilonShell=2e-3, batchSize=1024, minBatchSize=128)
    gi = GuidedIntegrator3(Problem3.from_dict(ball), st, ((-1.1, -1.1, -1.1), (1.1, 1.1, 1.1)), network_config=default_net_config3(), seed=7)
    gi.solve()
    gi.close()
    if it in (9, 149):
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        used.append(total - free)
print("device memory in use after 10 handles: %.1f MB, after 150: %.1f MB" % (used[0] / 2**20, used[1] / 2**20))
print("LEAK" if used[1] - used[0] > 64 * 2**20 else "no growth")
