"""walk-steps/s of the uniform integrator with a 3000-segment Neumann boundary on the tree (developer scratch):
zero-flux against emissive (the index-ordered sampling sweeps), and the scheduler weight"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from conftest import wiggly_problem
from elaina_amd import UniformIntegrator, UniformIntegratorSettings
for emissive in (False, True):
    p = wiggly_problem(emissive=emissive)
    for ww in (8, 4, 2, 1):
        it = UniformIntegrator(p, UniformIntegratorSettings((512, 512), 64, 64, 0.05))
        it.set_option("wait_weight", ww)
        it.solve()
        it.solve()
        st = it.last_stats
        print("%s 3000-segment boundary, 512x512, 64 spp, weight %d: %.3g walk steps, kernel %.1f ms -> %.3g steps/s" % ("emissive" if emissive else "zero-flux", ww, st["walk_steps"], st["kernel_ms"], st["walk_steps"] / (st["kernel_ms"] * 1e-3)), flush=True)
        it.close()
