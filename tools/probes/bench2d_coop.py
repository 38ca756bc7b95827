"""developer probe: the uniform integrator with a 3000-segment Neumann boundary on the tree: its silhouette and ray queries per lane
(coop=0) against answered by the wave through its task pools (wost_coop.h); same field and counters required"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from conftest import wiggly_problem
from elaina_amd import UniformIntegrator, UniformIntegratorSettings
specs = sys.argv[1:] or ["coop=0", "coop=1", "coop=1,wait_weight=2", "coop=1,wait_weight=8", "coop=1,pool_cap=512", "coop=1,pool_cap=256", "coop=1,ray_slot_trigger=64"]
for emissive in (False, True):
    p = wiggly_problem(int(os.environ.get("N_NEUMANN", "3000")), 400, emissive=emissive)
    ref = None
    for spec in specs:
        it = UniformIntegrator(p, UniformIntegratorSettings((512, 512), 64, 64, 0.05))
        for kv in spec.split(","):
            k, v = kv.split("=")
            it.set_option(k, float(v))
        it.solve()
        it.solve()
        st = it.last_stats
        f = it.solution.copy()
        if ref is None:
            ref = (f, st["walk_steps"])
        print("%-32s %s %d segments 512x512 64 spp: %.4g steps, kernel %.1f ms -> %.3g steps/s, same field %s same steps %s" % (
            spec, "emissive" if emissive else "zero-flux", len(p.n_segs), st["walk_steps"], st["kernel_ms"], st["walk_steps"] / (st["kernel_ms"] * 1e-3),
            np.array_equal(ref[0], f), ref[1] == st["walk_steps"]), flush=True)
        it.close()
