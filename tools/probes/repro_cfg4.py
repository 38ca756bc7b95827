"""run-to-run reproducibility of config 4 (ladybug, 1024^2, all samples trained) by network precision on the inference and the training side
(developer scratch): SPP=256 by default"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from elaina_amd import Problem
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
ladybug = Problem.load_scene("ladybug")
aabb = ((-100.0, -100.0), (600.0, 600.0))
spp = int(os.environ.get("SPP", "256"))


def run(prec, tprec, train):
    st = GuidedIntegratorSettings(frameSize=(1024, 1024), samplesPerPixel=spp, trainSppCount=train, maxWalkingDepth=64, epsilonShell=1.0)
    gi = GuidedIntegrator(ladybug, st, aabb)
    gi.network.set_option("precision", prec)
    gi.network.set_option("train_precision", tprec)
    gi.solve()
    out = gi.solution.copy(), dict(gi.last_stats), gi.network.params()
    gi.close()
    return out


cases = ((16, 16, spp), (16, 32, spp), (32, 16, spp), (16, 16, 0), (16, 16, spp))
if os.environ.get("ONLY_F16"):
    cases = ((32, 16, spp),) * int(os.environ["ONLY_F16"]) + ((16, 16, spp),) * int(os.environ.get("BOTH_F16", "0")) + ((16, 32, spp),) * int(os.environ.get("INF_F16", "0"))
for prec, tprec, train in cases:
    a, b = run(prec, tprec, train), run(prec, tprec, train)
    d = np.abs(a[0] - b[0])
    print("inference f%d training f%d, %d of %d samples trained: field equal %s (%d pixels differ), weights equal %s, steps %d / %d" % (
        prec, tprec, train, spp, np.array_equal(a[0], b[0]), int((d.max(axis=1) > 0).sum()), np.array_equal(a[2], b[2]), a[1]["walk_steps"], b[1]["walk_steps"]), flush=True)
