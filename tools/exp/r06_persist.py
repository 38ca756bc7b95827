"""Round-6 experiment: the persistent first launch of the uniform solve (pixels handed to resident lanes, longest expected chain
first) against the rounds of round 5 -- same field, same counters, time per solve.  Usage: python tools/exp/r06_persist.py [scene] [spp]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings

scene = sys.argv[1] if len(sys.argv) > 1 else "ladybug"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
frame = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
p = Problem.load_scene(scene)
KEYS = ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated", "neumann_hits")


def run(tag, opts, world=1, reps=3, trace=False):
    it = UniformIntegrator(p, UniformIntegratorSettings((frame, frame), spp, p.default_max_depth, p.default_eps))
    for k, v in opts.items():
        it.set_option(k, v)
    field = torch.zeros(frame * frame * 3, dtype=torch.float32, device="cuda")
    best = None
    for r in range(reps):
        field.zero_()
        torch.cuda.synchronize()
        if trace and r == reps - 1:
            os.environ["WOST_TRACE_LAUNCHES"] = "1"
        t = time.perf_counter()
        s = it.solve_sharded(0, world, field.data_ptr())
        torch.cuda.synchronize()
        s["wall_ms"] = (time.perf_counter() - t) * 1e3
        os.environ.pop("WOST_TRACE_LAUNCHES", None)
        if best is None or s["wall_ms"] < best["wall_ms"]:
            best = dict(s)
    f = field.cpu().numpy().copy()
    it.close()
    print("%-34s world %d: %8.2f ms wall, %8.2f ms kernels, %3d launches -> %.3e walk-steps/s   trips/step trav %.3f step %.3f" % (
        tag, world, best["wall_ms"], best["kernel_ms"], best["kernel_launches"], best["walk_steps"] / best["wall_ms"] * 1e3,
        best["trav_trips"] * 64.0 / best["walk_steps"], best["step_trips"] * 64.0 / best["walk_steps"]), flush=True)
    return f, best


base_f, base_s = run("rounds (persist=0)", {"persist": 0}, trace=True)
for tag, opts in [("persistent, defaults", {"persist": 1}),
                  ("persistent, round-5 tail", {"persist": 1, "long_steps": 0, "tail_sort": 0}),
                  ("persistent, sorted tail only", {"persist": 1, "long_steps": 0, "tail_sort": 1}),
                  ("persistent, long 1024 priority", {"persist": 1, "long_priority": 1}),
                  ("persistent, long 1024 thin 0", {"persist": 1, "long_thin": 0}),
                  ("persistent, long 1024 thin 8192", {"persist": 1, "long_thin": 8192}),
                  ("persistent, long 1280", {"persist": 1, "long_steps": 1280}),
                  ("persistent, long 1536", {"persist": 1, "long_steps": 1536}),
                  ("persistent, long 2048", {"persist": 1, "long_steps": 2048}),
                  ("persistent, long 896 cap 65536", {"persist": 1, "long_steps": 896, "long_cap": 65536}),
                  ("persistent, long 768 cap 131072", {"persist": 1, "long_steps": 768, "long_cap": 131072}),
                  ("persistent, long 1024 unsorted", {"persist": 1, "tail_sort": 0}),
                  ]:
    f, s = run(tag, opts, trace=tag in ("persistent, defaults", "persistent, long 896 cap 65536"))
    same = np.array_equal(f, base_f)
    cnt = all(s[k] == base_s[k] for k in KEYS)
    print("    field identical: %s, counters identical: %s" % (same, cnt), flush=True)
    if not cnt:
        print("    ", {k: (s[k], base_s[k]) for k in KEYS})
for world in (2,):
    f0, s0 = run("rounds", {"persist": 0}, world=world, reps=2)
    f1, s1 = run("persistent", {"persist": 1}, world=world, reps=2)
    print("    field identical: %s, counters identical: %s" % (np.array_equal(f0, f1), all(s0[k] == s1[k] for k in KEYS)), flush=True)
