"""Round-6 experiment: option sweeps of the persistent solve of config 2 (ladybug) / config 3 (fille).  Usage: r06_sweep.py scene 'k=v,k=v' 'k=v' ..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings

scene = sys.argv[1]
p = Problem.load_scene(scene)
frame, spp = 1024, 256
base = None
for spec in sys.argv[2:]:
    opts = {}
    trace = False
    for kv in spec.split(","):
        if kv == "trace":
            trace = True
        elif kv:
            k, v = kv.split("=")
            opts[k] = float(v)
    it = UniformIntegrator(p, UniformIntegratorSettings((frame, frame), spp, p.default_max_depth, p.default_eps))
    for k, v in opts.items():
        it.set_option(k, v)
    field = torch.zeros(frame * frame * 3, dtype=torch.float32, device="cuda")
    ts = []
    for r in range(3):
        field.zero_()
        torch.cuda.synchronize()
        if trace and r == 2:
            os.environ["WOST_TRACE_LAUNCHES"] = "1"
        t = time.perf_counter()
        s = it.solve_sharded(0, 1, field.data_ptr())
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
        os.environ.pop("WOST_TRACE_LAUNCHES", None)
    f = field.cpu().numpy().copy()
    it.close()
    if base is None:
        base = (f, s["walk_steps"])
    print("%-8s %-60s %8.2f ms (min of 3; %s) %2d launches -> %.3e walk-steps/s  same field %s" % (
        scene, spec, min(ts), " ".join("%.1f" % t for t in ts), s["kernel_launches"], s["walk_steps"] / min(ts) * 1e3,
        np.array_equal(f, base[0]) and s["walk_steps"] == base[1]), flush=True)
