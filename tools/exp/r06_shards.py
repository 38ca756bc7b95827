"""Round-6 experiment: shard 0 of N of config 2 on one GPU under option sets.  Usage: r06_shards.py 'k=v,...' ..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings

p = Problem.load_scene("ladybug")
frame, spp = 1024, 256
base = {}
for spec in sys.argv[1:]:
    opts = {}
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("=")
            opts[k] = float(v)
    it = UniformIntegrator(p, UniformIntegratorSettings((frame, frame), spp, p.default_max_depth, p.default_eps))
    for k, v in opts.items():
        it.set_option(k, v)
    field = torch.zeros(frame * frame * 3, dtype=torch.float32, device="cuda")
    line = []
    for world in (8, 4, 2):
        ts = []
        for r in range(3):
            field.zero_()
            torch.cuda.synchronize()
            t = time.perf_counter()
            s = it.solve_sharded(0, world, field.data_ptr())
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
        f = field.cpu().numpy().copy()
        if world not in base:
            base[world] = (f, s["walk_steps"])
        ok = np.array_equal(f, base[world][0]) and s["walk_steps"] == base[world][1]
        line.append("N=%d: %6.1f ms %2d launches %s" % (world, min(ts), s["kernel_launches"], "ok" if ok else "DIFFERENT"))
    it.close()
    print("%-50s %s" % (spec or "(defaults)", "   ".join(line)), flush=True)
