"""Round-6 experiment: time-to-1spp (and 2, 4, 8 spp) with the pixels of the one-launch path in longest-chain-first order."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings

p = Problem.load_scene("ladybug")
frame = 1024
for spp in (1, 2, 4, 8, 16):
    res = {}
    for tag, opts in [("default", {}), ("few_order", {"few_order": 1}), ("refill=1", {"refill": 1}), ("refill=1 + order", {"refill": 1, "few_order": 1}),
                      ("persist", {"persist": 1})]:
        it = UniformIntegrator(p, UniformIntegratorSettings((frame, frame), spp, p.default_max_depth, p.default_eps))
        for k, v in opts.items():
            it.set_option(k, v)
        field = torch.zeros(frame * frame * 3, dtype=torch.float32, device="cuda")
        ts = []
        for r in range(6):
            torch.cuda.synchronize()
            t = time.perf_counter()
            s = it.solve_sharded(0, 1, field.data_ptr())
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
        res[tag] = field.cpu().numpy().copy()
        it.close()
        print("spp %2d %-18s: cold %.2f ms, steady %.2f ms (min %.2f), %d launches, identical to default: %s" % (
            spp, tag, ts[0], sorted(ts[1:])[2], min(ts), s["kernel_launches"], np.array_equal(res[tag], res["default"])), flush=True)
