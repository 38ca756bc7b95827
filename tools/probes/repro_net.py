"""the half-precision network kernels alone, the same call repeated on the same state: inference outputs and training gradients compared
across repeats (developer scratch).  N=524288 REPS=40"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from elaina_amd.guided import GuidingNetwork
n, reps = int(os.environ.get("N", "524288")), int(os.environ.get("REPS", "40"))
rng = np.random.default_rng(1)
xy = rng.uniform(0.02, 0.98, (n, 2)).astype(np.float32)
dl = (rng.normal(size=(n, 33)) * 1e-4).astype(np.float32)
for prec in (16, 32):
    net = GuidingNetwork(seed=7)
    net.set_option("precision", prec)
    net.set_option("train_precision", prec)
    ref = net.inference(xy)
    bad = 0
    for r in range(reps):
        out = net.inference(xy)
        d = out != ref
        if d.any():
            bad += 1
            rows = np.nonzero(d.any(axis=1))[0]
            print("  f%d inference repeat %d: %d values in %d points differ, first points %s" % (prec, r, int(d.sum()), len(rows), rows[:8]), flush=True)
    print("f%d inference: %d of %d repeats differ from the first" % (prec, bad, reps), flush=True)
    net.train_step(xy, dl, apply_update=False)
    g0 = net.gradients()
    bad = 0
    for r in range(reps):
        net.train_step(xy, dl, apply_update=False)
        g = net.gradients()
        d = g != g0
        if d.any():
            bad += 1
            print("  f%d training repeat %d: %d of %d gradient entries differ (mlp part: %d), max |diff| %.3g" % (prec, r, int(d.sum()), g.size, int(d[:net.n_mlp_params].sum()),
                                                                                                                   float(np.abs(g - g0).max())), flush=True)
    print("f%d training gradients: %d of %d repeats differ from the first" % (prec, bad, reps), flush=True)
    net.close()
