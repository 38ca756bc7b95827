"""developer probe: time-to-1spp of the 1024^2 ladybug frame under launch options; every variant must give the same field"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch  # noqa
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings

scene = sys.argv[1] if len(sys.argv) > 1 else "ladybug"
p = Problem.load_scene(scene)
it = UniformIntegrator(p, UniformIntegratorSettings((1024, 1024), 1, p.default_max_depth, 1.0))
field = torch.zeros(1024 * 1024 * 3, dtype=torch.float32, device="cuda")
ref = None
for spec in sys.argv[2:] or ["quad=0", "quad=-1"]:
    for kv in spec.split(","):
        k, v = kv.split("=")
        it.set_option(k, float(v))
    best = None
    for _ in range(6):
        field.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = it.solve_sharded(0, 1, field.data_ptr())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        if best is None or ms < best[0]:
            best = (ms, dict(s))
    f = field.cpu().numpy()
    if ref is None:
        ref = f.copy()
    print("%-40s %.3f ms wall, kernel %.3f ms, %d launches, %d steps, same field: %s" % (
        spec, best[0], best[1]["kernel_ms"], best[1]["kernel_launches"], best[1]["walk_steps"], np.array_equal(ref, f)), flush=True)
it.close()
