"""Developer probe: what ONE rank of an N-GPU solve does (shard 0 of N on this GPU) -- all there is for the strong-scaling
estimate of bench.py while no multi-GPU node is available.  Config 2 (ladybug, uniform, 1024^2, 256 spp) and config 5 (ladybug,
guided, 2048^2; by default 64 samples, 16 of them trained -- a solve of the full 1024 takes minutes -- scaled to the full job
by samples) for N = 1, 2, 4, 8.  Writes a JSON file when --out is given (profiles/r04_shards.json).
Usage: python tools/gpu_shard_probe.py [--out file] [--configs 2,5] [--spp5 64] [--train5 16] [k=v options of the uniform handle ...]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
from elaina_amd.build import source_id
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--configs", default="2,5")
ap.add_argument("--spp5", type=int, default=64)
ap.add_argument("--train5", type=int, default=16)
ap.add_argument("--net-precision", type=int, default=16)
ap.add_argument("options", nargs="*")
a = ap.parse_args()
p = Problem.load_scene("ladybug")
out = {"source_id": source_id(), "what": "shard 0 of N on ONE MI355X: the time a rank of an N-GPU job spends before the one exchange of the field "
       "(12.6 MB all-reduce at 1024^2, 50 MB all-gather at 2048^2: < 1 ms over xGMI); efficiency = t(1) / (N t(N))"}
if "2" in a.configs.split(","):
    it = UniformIntegrator(p, UniformIntegratorSettings((1024, 1024), 256, p.default_max_depth, 1.0))
    for o in a.options:
        k, v = o.split("=")
        it.set_option(k, float(v))
    field = torch.zeros(1024 * 1024 * 3, dtype=torch.float32, device="cuda")
    rows, base = [], None
    for world in (1, 2, 4, 8):
        best = None
        for _ in range(3):
            field.zero_()
            s = it.solve_sharded(0, world, field.data_ptr())
            if best is None or s["solve_ms"] < best["solve_ms"]:
                best = dict(s)
        base = base or best["solve_ms"]
        rows.append({"ranks": world, "shard_ms": best["solve_ms"], "launches": best["kernel_launches"], "walk_steps": best["walk_steps"],
                     "aggregate_walk_steps_per_s": best["walk_steps"] * world / best["solve_ms"] * 1e3, "efficiency": base / (world * best["solve_ms"])})
        print("config 2, shard 0 of %d: %.1f ms, %d launches -> %.2e steps/s aggregate, efficiency %.0f %%" % (
            world, best["solve_ms"], best["kernel_launches"], rows[-1]["aggregate_walk_steps_per_s"], 100 * rows[-1]["efficiency"]), flush=True)
    out["config2"] = {"workload": "ladybug uniform 1024x1024 256 spp depth 64", "rows": rows}
    it.close()
if "5" in a.configs.split(","):
    frame = 2048
    field = torch.zeros(frame * frame * 3, dtype=torch.float32, device="cuda")
    rows, base = [], None
    for world in (1, 2, 4, 8):
        st = GuidedIntegratorSettings(frameSize=(frame, frame), samplesPerPixel=a.spp5, trainSppCount=a.train5, maxWalkingDepth=64, epsilonShell=1.0)
        gi = GuidedIntegrator(p, st, ((-100.0, -100.0), (600.0, 600.0)))
        if a.net_precision == 16:
            gi.network.set_option("precision", 16)
            gi.network.set_option("train_precision", 16)
        best = None
        for _ in range(2):
            field.zero_()
            torch.cuda.synchronize()
            s = gi.solve_sharded(0, world, field.data_ptr())
            torch.cuda.synchronize()
            if best is None or s["solve_ms"] < best["solve_ms"]:
                best = dict(s)
        gi.close()
        base = base or best["solve_ms"]
        rows.append({"ranks": world, "shard_ms": best["solve_ms"], "train_ms": best["train_ms"], "walk_steps": best["walk_steps"],
                     "aggregate_walk_steps_per_s": best["walk_steps"] * world / best["solve_ms"] * 1e3, "efficiency": base / (world * best["solve_ms"])})
        print("config 5 (%d spp, %d trained), shard 0 of %d: %.1f ms (training %.1f) -> %.2e steps/s aggregate, efficiency %.0f %%" % (
            a.spp5, a.train5, world, best["solve_ms"], best["train_ms"], rows[-1]["aggregate_walk_steps_per_s"], 100 * rows[-1]["efficiency"]), flush=True)
    out["config5"] = {"workload": "ladybug guided 2048x2048, %d spp (%d trained) of the job's 1024 (256), f%d network, per-shard networks" % (
        a.spp5, a.train5, a.net_precision), "rows": rows}
if a.out:
    json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out))
