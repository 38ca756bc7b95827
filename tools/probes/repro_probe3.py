"""the fused loss launched twice on the same inputs (WOST_NET_FUSED_LOSS=3), developer scratch"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
os.environ["WOST_NET_FUSED_LOSS"] = "3"
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
from elaina_amd.scenes import BRIGHT_DISC_AABB, bright_disc_scene
p = bright_disc_scene()
for rep in range(int(os.environ.get("REPS", "6"))):
    st = GuidedIntegratorSettings(frameSize=(128, 128), samplesPerPixel=2, trainSppCount=2, maxWalkingDepth=128, epsilonShell=0.05, batchSize=65536, minBatchSize=8192)
    g = GuidedIntegrator(p, st, BRIGHT_DISC_AABB)
    g.network.set_option("precision", 16)
    g.network.set_option("train_precision", 16)
    g.solve()
    g.close()
