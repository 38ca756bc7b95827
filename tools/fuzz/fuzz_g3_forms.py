"""GuidedIntegrator<3> in its three forms against each other on random far, tree-sized scenes (the scenes of fuzz_far_trees.py) with the
reference's EIGHT-level network -- the shape g3_fused_kernel and the MFMA kernels cover, which the oracle-backed fuzzers avoid (four
levels keep the oracle's dense grid small): one launch per sample with the walkers spread over the lanes, one launch per sample with
64 walkers per wave (four units of the matrices at once), and the launches per depth (walkers spread; 64 per wave).  Frozen and training solves, frames from 80 to
51 200 pixels; fields, counters and (trained) final parameters must agree bit for bit.  GPU only, no oracle.

usage: fuzz_g3_forms.py [first seed] [count [seconds]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.dirname(__file__))
import fuzz_far_trees as F  # noqa: E402

FORMS = (("fused, spread", {}), ("fused, 64 walkers per wave", {"WOST3_G_SHIFT": "0", "WOST3_G_FUSED": "1"}), ("launches per depth", {"WOST3_G_FUSED": "0"}),
         ("launches per depth, 64 walkers per wave", {"WOST3_G_FUSED": "0", "WOST3_G_SHIFT": "0"}))


def case(seed):
    rng = np.random.default_rng(97_000 + seed)
    train = bool(rng.uniform() < 0.4)
    c = F.case_guided3d(seed, train)
    c["cfg"] = F.NetCfg(41, n_levels=8)
    w, h = [(10, 8), (40, 32), (96, 80), (256, 200)][int(rng.integers(0, 4))]
    c.update(w=w, h=h, spp=int(rng.choice([2, 3, 5])), depth=int(rng.choice([8, 20, 40])))
    if train:
        c.update(train=int(rng.choice([1, 2, 3])), batch=(int(rng.choice([256, 1024, 4096])), 128))
        c["spp"] = max(c["spp"], c["train"] + 1)
    n, n_mlp = F.net_counts(c["cfg"], 3)
    c["params"] = F.rand_params(n, n_mlp, rng, *((0.15, 0.5) if train else (0.3, 1.0)))
    c["what"] = "%s | frame %dx%d spp %d train %d depth %d uf %g %g" % (" ".join(c["feat"]), w, h, c["spp"], c["train"], c["depth"], *c["uf"])
    return c


def run_forms(c):
    """None, or the description of the first disagreement"""
    ref = None
    for name, env in FORMS:
        old = {k: os.environ.get(k) for k in ("WOST3_G_SHIFT", "WOST3_G_FUSED")}
        for k in old:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            r = F.hip_run(c)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
        if ref is None:
            ref = r
            continue
        d = F.same(c, ref, r)
        if d:
            return "%s against %s: %s" % (name, FORMS[0][0], d)
    return None


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
    t0, bad, done = time.time(), 0, 0
    for seed in range(first, first + count):
        if time.time() - t0 > seconds:
            break
        c = case(seed)
        d = run_forms(c)
        done += 1
        if d:
            bad += 1
            print("seed %d MISMATCH %s | %s" % (seed, d, c["what"]), flush=True)
    print("fuzz_g3_forms: seeds %d..%d, %d run in %.0f s, %d mismatches" % (first, first + done - 1, done, time.time() - t0, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
