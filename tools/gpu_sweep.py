#!/usr/bin/env python3
"""Developer sweep of kernel knobs on the GPU box (throughput only; parity is in tests/)."""
import itertools
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings


def main():
    spp = int(os.environ.get("SPP", "32"))
    scene = os.environ.get("SCENE", "ladybug")
    p = Problem.load_scene(scene)
    knobs = {}
    for a in sys.argv[1:]:
        k, v = a.split("=")
        knobs[k] = [float(x) for x in v.split(",")]
    keys = list(knobs)
    it = UniformIntegrator(p, UniformIntegratorSettings((1024, 1024), spp, p.default_max_depth, 1.0))
    it.solve()
    for combo in itertools.product(*[knobs[k] for k in keys]):
        for k, v in zip(keys, combo):
            it.set_option(k, v)
        best = None
        for _ in range(2):
            it.solve()
            s = it.last_stats
            if best is None or s["solve_ms"] < best["solve_ms"]:
                best = dict(s)
        print("%s: wall %.1f ms kernel %.1f ms launches %d -> %.3e steps/s  inner/step %.2f leaf/step %.2f" % (
            dict(zip(keys, combo)), best["solve_ms"], best["kernel_ms"], best["kernel_launches"],
            best["walk_steps"] / best["solve_ms"] * 1e3, best["inner_visits"] / best["walk_steps"],
            best["leaf_visits"] / best["walk_steps"]), flush=True)
        vis = best["inner_visits"] + best["leaf_visits"]
        print("     lane utilisation: traversal phase %.3f (trips %d) step phase %.3f (trips %d)" % (
            vis / (64.0 * max(best["trav_trips"], 1)), best["trav_trips"],
            best["walk_steps"] / (64.0 * max(best["step_trips"], 1)), best["step_trips"]), flush=True)
        print("     max stack depth after a visit: %d" % best["reserved"], flush=True)
    it.close()


if __name__ == "__main__":
    main()
