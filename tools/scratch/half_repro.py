import numpy as np, sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings, GuidingNetwork
from conftest import box_problem
prob = box_problem(d_sides=(0, 2), n_sides=(1, 3), value=lambda x, y: y, flux=lambda x, y, s: 0.0, n_per_side=8)
AABB = ((-0.1, -0.1), (1.1, 1.1))
net = GuidingNetwork(seed=7)
net.set_option("precision", 16)
rng = np.random.default_rng(1)
xy = rng.uniform(0, 1, (100000, 2)).astype(np.float32)
a = net.inference(xy); b = net.inference(xy)
print("inference repeat equal:", np.array_equal(a, b))
net.close()
for train in (0, 24):
    res = []
    for k in range(2):
        st = GuidedIntegratorSettings(frameSize=(64, 64), samplesPerPixel=48, trainSppCount=train, maxWalkingDepth=48, epsilonShell=1e-3, batchSize=4096, minBatchSize=1024)
        gi = GuidedIntegrator(prob, st, AABB, seed=3)
        gi.network.set_option("precision", 16)
        gi.solve()
        res.append((gi.solution.copy(), gi.network.params(), gi.network.inference_params(), dict(gi.last_stats)))
        gi.close()
    print("train", train, "field equal", np.array_equal(res[0][0], res[1][0]), "params equal", np.array_equal(res[0][1], res[1][1]),
          "ema equal", np.array_equal(res[0][2], res[1][2]), "ndiff", int((res[0][0] != res[1][0]).sum()), res[0][3]["walk_steps"], res[1][3]["walk_steps"])
