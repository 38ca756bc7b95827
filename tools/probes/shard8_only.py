"""shard 0 of 8 of config 2 on one GPU, two solves (for rocprofv3 --pmc: what a wave waits for at two waves per SIMD)"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
p = Problem.load_scene("ladybug")
it = UniformIntegrator(p, UniformIntegratorSettings((1024, 1024), 256, p.default_max_depth, p.default_eps))
field = torch.zeros(1024 * 1024 * 3, dtype=torch.float32, device="cuda")
world = int(os.environ.get("SHARDS", "8"))
for _ in range(2):
    field.zero_()
    s = it.solve_sharded(0, world, field.data_ptr())
print("shard 0 of %d: %.1f ms kernels, %d walk steps" % (world, s["kernel_ms"], s["walk_steps"]))
it.close()
