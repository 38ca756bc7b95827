"""WOST_NET_CHECK3=1: config 4 with fp32 inference and half-precision training, every training kernel launched three times on the same
inputs and compared on the device (the counts are printed when the network goes); developer scratch of EXPERIMENTS 20"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
os.environ.setdefault("WOST_NET_CHECK3", "1")
from elaina_amd import Problem
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
ladybug = Problem.load_scene("ladybug")
spp = int(os.environ.get("SPP", "256"))
for rep in range(int(os.environ.get("REPS", "10"))):
    st = GuidedIntegratorSettings(frameSize=(1024, 1024), samplesPerPixel=spp, trainSppCount=spp, maxWalkingDepth=64, epsilonShell=1.0)
    gi = GuidedIntegrator(ladybug, st, ((-100.0, -100.0), (600.0, 600.0)))
    gi.network.set_option("precision", int(os.environ.get("INF", "32")))
    gi.network.set_option("train_precision", 16)
    gi.solve()
    print("solve %d: steps %d" % (rep, gi.last_stats["walk_steps"]), flush=True)
    gi.close()
