"""reproducibility of the guided solve on the bright-disc scene, half precision, per training order and with / without the loss gradient
inside the training forward (developer scratch): which of field / weights differ between two identical runs"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
from elaina_amd.scenes import BRIGHT_DISC_AABB, bright_disc_scene
p = bright_disc_scene()
w, depth, eps = 128, 128, 0.05


def run(pipeline, group, spp=128, train=64):
    st = GuidedIntegratorSettings(frameSize=(w, w), samplesPerPixel=spp, trainSppCount=train, maxWalkingDepth=depth, epsilonShell=eps,
                                  batchSize=65536, minBatchSize=8192)
    g = GuidedIntegrator(p, st, BRIGHT_DISC_AABB)
    g.network.set_option("precision", 16)
    g.network.set_option("train_precision", 16)
    g.set_option("pipeline", pipeline)
    g.set_option("train_group", group)
    g.solve()
    out = (g.solution.copy(), g.network.params(), dict(g.last_stats))
    g.close()
    return out


for fused in ("1", "0"):
    os.environ["WOST_NET_FUSED_LOSS"] = fused
    for pipeline, group in ((0, 1), (0, 4), (1, 1), (1, 4), (0, 16)):
        a, b = run(pipeline, group), run(pipeline, group)
        print("fused %s pipeline %d group %d: field equal %s, weights equal %s, steps %d / %d, optimizer %d / %d" % (
            fused, pipeline, group, np.array_equal(a[0], b[0]), np.array_equal(a[1], b[1]), a[2]["walk_steps"], b[2]["walk_steps"],
            a[2]["optimizer_steps"], b[2]["optimizer_steps"]), flush=True)
    # trained samples only: is it the training?
    a, b = run(0, 4, spp=16, train=16), run(0, 4, spp=16, train=16)
    print("fused %s (0, 4), 16 trained samples only: field equal %s, weights equal %s" % (fused, np.array_equal(a[0], b[0]), np.array_equal(a[1], b[1])), flush=True)
