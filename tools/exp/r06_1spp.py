"""Round-6 experiment: time-to-1spp (and 2, 4 spp) of config 2's frame under option sets.  Usage: r06_1spp.py 'k=v,...' ..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings

p = Problem.load_scene("ladybug")
frame = 1024
for spp in (1, 2, 4):
    base = None
    for spec in sys.argv[1:] or [""]:
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
        for r in range(8):
            torch.cuda.synchronize()
            if trace and r == 7:
                os.environ["WOST_TRACE_LAUNCHES"] = "1"
            t = time.perf_counter()
            s = it.solve_sharded(0, 1, field.data_ptr())
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
            os.environ.pop("WOST_TRACE_LAUNCHES", None)
        f = field.cpu().numpy().copy()
        if base is None:
            base = f
        it.close()
        print("spp %d %-46s: steady %.3f ms (median of 7), min %.3f, kernels %.3f ms, %d launches, same field %s" % (
            spp, spec or "(defaults)", sorted(ts[1:])[3], min(ts), s["kernel_ms"], s["kernel_launches"], np.array_equal(f, base)), flush=True)
