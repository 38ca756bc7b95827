import numpy as np, sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
from conftest import box_problem
prob = box_problem(d_sides=(0, 2), n_sides=(1, 3), value=lambda x, y: y, flux=lambda x, y, s: 0.0, n_per_side=8)
AABB = ((-0.1, -0.1), (1.1, 1.1))
rng = np.random.default_rng(1)
pts = rng.uniform(0, 1, (5000, 2)).astype(np.float32)
for spp in (1, 4, 48):
    res = []
    for k in range(3):
        st = GuidedIntegratorSettings(frameSize=(64, 64), samplesPerPixel=spp, trainSppCount=0, maxWalkingDepth=48, epsilonShell=1e-3, batchSize=4096, minBatchSize=1024)
        gi = GuidedIntegrator(prob, st, AABB, seed=3)
        gi.network.set_option("precision", 16)
        q = gi.queryNetwork(pts)
        gi.solve()
        res.append((gi.solution.copy(), q, dict(gi.last_stats)))
        gi.close()
    print("spp", spp, "q equal", np.array_equal(res[0][1], res[1][1]), "fields 0v1", int((res[0][0] != res[1][0]).sum()), "1v2", int((res[1][0] != res[2][0]).sum()),
          "0v2", int((res[0][0] != res[2][0]).sum()), [r[2]["guided_steps"] for r in res])
    d = np.nonzero((res[0][0] != res[1][0]).any(axis=1))[0]
    if len(d):
        print("  first diffs", d[:8], res[0][0][d[:3], 0], res[1][0][d[:3], 0])
