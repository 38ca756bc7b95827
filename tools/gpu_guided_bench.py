"""BASELINE config 4 (ladybug, guided integrator with online training, 1024^2, 256 spp) on one
MI355X: wall time, walk-steps/s, share of the training passes.  Prints one JSON line.
Usage: python tools/gpu_guided_bench.py [--frame 1024] [--spp 256] [--train-spp 256] [--depth 64]"""
import argparse
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from elaina_amd import Problem  # noqa: E402
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="ladybug")
ap.add_argument("--frame", type=int, default=1024)
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--train-spp", type=int, default=256)
ap.add_argument("--depth", type=int, default=64)
ap.add_argument("--guided-depth", type=int, default=10)
a = ap.parse_args()

prob = Problem.load_scene(a.scene)
st = GuidedIntegratorSettings(frameSize=(a.frame, a.frame), samplesPerPixel=a.spp, trainSppCount=a.train_spp,
                              maxWalkingDepth=a.depth, epsilonShell=1.0, maxGuidedDepthInTrainingPhase=a.guided_depth,
                              maxGuidedDepthInGuidingPhase=a.guided_depth)
t0 = time.time()
gi = GuidedIntegrator(prob, st, ((-100.0, -100.0), (600.0, 600.0)))
t_create = time.time() - t0
gi.solve()
s = gi.last_stats
print(json.dumps({
    "workload": "%s guided %dx%d %d spp (train %d) depth %d" % (a.scene, a.frame, a.frame, a.spp, a.train_spp, a.depth),
    "solve_s": s["solve_ms"] / 1e3, "train_s": s["train_ms"] / 1e3, "create_s": t_create,
    "walk_steps": s["walk_steps"], "walk_steps_per_s": s["walk_steps"] / (s["solve_ms"] / 1e3),
    "guided_steps": s["guided_steps"], "train_samples": s["train_samples"], "optimizer_steps": s["optimizer_steps"],
    "kernel_launches": s["kernel_launches"], "truncated": s["walks_truncated"], "started": s["walks_started"],
    "mean": float(np.mean(gi.solution)),
}))
