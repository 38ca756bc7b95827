import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from elaina_amd import Problem
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
prob = Problem.load_scene("ladybug")
aabb = ((-100.0, -100.0), (600.0, 600.0))
def run(fused, spp, train, mgd, w=96):
    os.environ["WOST_GUIDED_FUSED"] = fused
    st = GuidedIntegratorSettings(frameSize=(w, w), samplesPerPixel=spp, trainSppCount=train, maxWalkingDepth=64, epsilonShell=1.0,
                                  maxGuidedDepthInTrainingPhase=mgd, maxGuidedDepthInGuidingPhase=mgd, batchSize=4096, minBatchSize=512)
    gi = GuidedIntegrator(prob, st, aabb, seed=11)
    gi.network.set_option("precision", 16)
    gi.network.set_option("train_precision", 16)
    gi.solve()
    r = gi.solution.copy(), dict(gi.last_stats)
    gi.close()
    return r
for name, (spp, train, mgd) in {"uniform 1spp": (1, 0, 0), "uniform 3spp": (3, 0, 0), "frozen 1spp": (1, 0, 5), "frozen 3spp": (3, 0, 5), "train 3": (3, 3, 5)}.items():
    a, sa = run("0", spp, train, mgd)
    b, sb = run("1", spp, train, mgd)
    bad = np.flatnonzero((a != b).any(axis=1))
    print(name, "equal" if len(bad) == 0 else "DIFF %d pixels first %s" % (len(bad), bad[:8]), sa["walk_steps"], sb["walk_steps"], flush=True)
