#!/bin/bash
mkdir -p gpurun_out/r06_f
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_f/shard_trace.txt
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
p = Problem.load_scene("ladybug")
for world in (8, 4):
    for opts in ({"under_frac": 0}, {"under_frac": 0.125}):
        it = UniformIntegrator(p, UniformIntegratorSettings((1024, 1024), 256, p.default_max_depth, p.default_eps))
        for k, v in opts.items():
            it.set_option(k, v)
        field = torch.zeros(1024 * 1024 * 3, dtype=torch.float32, device="cuda")
        it.solve_sharded(0, world, field.data_ptr())
        field.zero_()
        os.environ["WOST_TRACE_LAUNCHES"] = "1"
        print("world", world, opts, flush=True)
        t = time.perf_counter()
        s = it.solve_sharded(0, world, field.data_ptr())
        torch.cuda.synchronize()
        print("  -> %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
        os.environ.pop("WOST_TRACE_LAUNCHES")
        it.close()
PY
