import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
p = Problem.load_scene("ladybug")
it = UniformIntegrator(p, UniformIntegratorSettings((1024, 1024), 1, p.default_max_depth, p.default_eps))
field = torch.zeros(1024 * 1024 * 3, dtype=torch.float32, device="cuda")
for r in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    s = it.solve_sharded(0, 1, field.data_ptr()); torch.cuda.synchronize()
    print("call %.3f ms kernel %.3f ms" % ((time.perf_counter() - t) * 1e3, s["kernel_ms"]), flush=True)
it.close()
