"""Developer probe: what ONE rank of an N-GPU uniform solve does (shard 0 of N on this GPU), for the
strong-scaling estimate of bench.py.  Usage: python tools/gpu_shard_probe.py [scene] [spp] [k=v ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings

scene = sys.argv[1] if len(sys.argv) > 1 else "ladybug"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
p = Problem.load_scene(scene)
it = UniformIntegrator(p, UniformIntegratorSettings((1024, 1024), spp, p.default_max_depth, 1.0))
for a in sys.argv[3:]:
    k, v = a.split("=")
    it.set_option(k, float(v))
field = torch.zeros(1024 * 1024 * 3, dtype=torch.float32, device="cuda")
base = None
for world in (1, 2, 4, 8):
    best = None
    for _ in range(3):
        field.zero_()
        s = it.solve_sharded(0, world, field.data_ptr())
        if best is None or s["solve_ms"] < best["solve_ms"]:
            best = dict(s)
    if base is None:
        base = best["solve_ms"]
    print("shard 0 of %d: %.1f ms, %d launches, %.3e steps -> %d ranks would give %.2e steps/s (efficiency %.0f %%)" % (
        world, best["solve_ms"], best["kernel_launches"], best["walk_steps"], world,
        best["walk_steps"] * world / best["solve_ms"] * 1e3, 100.0 * base / (world * best["solve_ms"])), flush=True)
it.close()
