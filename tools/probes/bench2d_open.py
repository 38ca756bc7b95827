"""an open scene: a Dirichlet polyline alone, seen from three scene sizes away -- every walk that misses it strays
(developer scratch: what the far-walker handling costs when it is the rule, not the exception)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.dirname(__file__))
from fuzz_parity import polyline
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
rng = np.random.default_rng(1)
for nd in (300, 20000):
    dv, ds = polyline(rng, nd, 30.0, (5.0, -3.0), 0.2, True, False)
    p = Problem(d_verts=dv, d_segs=ds, d_colors=rng.uniform(0, 1, (len(dv), 6)).astype(np.float32), probe=(100.0, 0.0, 0.0, 1.0, 0.0))
    for depth in (16, 64):
        it = UniformIntegrator(p, UniformIntegratorSettings((512, 512), 16, depth, 0.05))
        it.solve()
        it.solve()
        st = it.last_stats
        print("%d segments, 512x512, 16 spp, depth %d: %.3g walk steps, solve %.1f ms (kernel %.1f ms, %d launches) -> %.3g steps/s, truncated %d of %d walks" % (
            nd, depth, st["walk_steps"], st["solve_ms"], st["kernel_ms"], st["kernel_launches"], st["walk_steps"] / (st["solve_ms"] * 1e-3), st["walks_truncated"], st["walks_started"]), flush=True)
        it.close()
