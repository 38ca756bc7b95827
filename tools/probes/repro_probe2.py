"""narrowing the run-to-run differences of the loss gradient inside the training forward (developer scratch)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
from elaina_amd.scenes import BRIGHT_DISC_AABB, bright_disc_scene
p = bright_disc_scene()
depth, eps = 128, 0.05


def run(w, spp, train, batch, min_batch):
    st = GuidedIntegratorSettings(frameSize=(w, w), samplesPerPixel=spp, trainSppCount=train, maxWalkingDepth=depth, epsilonShell=eps,
                                  batchSize=batch, minBatchSize=min_batch)
    g = GuidedIntegrator(p, st, BRIGHT_DISC_AABB)
    g.network.set_option("precision", 16)
    g.network.set_option("train_precision", 16)
    g.solve()
    out = (g.solution.copy(), g.network.params(), dict(g.last_stats))
    g.close()
    return out


for w, spp, train, batch, mb in ((128, 1, 1, 65536, 8192), (128, 1, 1, 4224, 1024), (128, 2, 2, 65536, 8192), (64, 4, 4, 4224, 1024), (128, 4, 4, 16384, 1024)):
    res = {}
    for fused in ("1", "0"):
        os.environ["WOST_NET_FUSED_LOSS"] = fused
        a, b = run(w, spp, train, batch, mb), run(w, spp, train, batch, mb)
        res[fused] = a
        d = np.abs(a[1] - b[1])
        print("frame %d spp %d batch %d fused %s: weights equal %s (%d differ, max %.3g), field equal %s, optimizer steps %d, train samples %d" % (
            w, spp, batch, fused, np.array_equal(a[1], b[1]), int((d > 0).sum()), float(d.max()), np.array_equal(a[0], b[0]), a[2]["optimizer_steps"],
            a[2]["train_samples"]), flush=True)
    d = np.abs(res["1"][1] - res["0"][1])
    print("   fused against unfused: weights equal %s (%d differ, max %.3g)" % (np.array_equal(res["1"][1], res["0"][1]), int((d > 0).sum()), float(d.max())), flush=True)
