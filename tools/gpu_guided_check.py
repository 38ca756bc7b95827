"""Developer check of the guided integrator against the CPU oracle (run on the GPU box)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import box_problem  # noqa: E402
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings  # noqa: E402
from oracle.oracle import Oracle, default_net_config, guided_settings  # noqa: E402

o = Oracle()
cfg = default_net_config()
prob = box_problem(d_sides=(0, 2), n_sides=(1, 3), value=lambda x, y: y, flux=lambda x, y, s: 0.0, n_per_side=8)
W = H = 48
aabb = ((-0.1, -0.1), (1.1, 1.1))


def run(spp, train_spp, uf=(0.5, 0.5), min_batch=512, batch=2048, params=None, depth=32, mgd=(10, 10)):
    st = GuidedIntegratorSettings(frameSize=(W, H), samplesPerPixel=spp, trainSppCount=train_spp, maxWalkingDepth=depth,
                                  epsilonShell=1e-3, uniformFractionInTrainingPhase=uf[0], uniformFractionInGuidingPhase=uf[1],
                                  maxGuidedDepthInTrainingPhase=mgd[0], maxGuidedDepthInGuidingPhase=mgd[1],
                                  batchSize=batch, minBatchSize=min_batch)
    gi = GuidedIntegrator(prob, st, aabb, seed=7)
    if params is not None:
        gi.network.set_params(params)
    p0 = gi.network.params()
    t = time.time()
    gi.solve()
    tg = time.time() - t
    gs = guided_settings(W, H, spp, depth, 1e-3, aabb[0], aabb[1], train_spp_count=train_spp, uniform_fraction=uf,
                         max_guided_depth=mgd, batch_size=batch, min_batch_size=min_batch)
    po = p0.copy()
    t = time.time()
    r = o.solve_guided(prob.as_dict(), gs, cfg, po, threads=16, dump_spp=min(train_spp, spp) - 1 if train_spp > 0 else -1)
    to = time.time() - t
    return gi, r, tg, to


def report(name, gi, r, tg, to):
    f, fo = gi.solution, r["field"]
    st = gi.last_stats
    print("==", name, "gpu %.2fs oracle %.2fs" % (tg, to))
    print("  gpu   ", {k: st[k] for k in ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated", "neumann_hits",
                                         "guided_steps", "train_samples", "optimizer_steps", "kernel_launches")})
    print("  oracle", {k: v for k, v in r.items() if k not in ("field", "train_set")})
    d = np.abs(f - fo)
    rel = d / np.maximum(np.abs(fo), 1e-3)
    print("  field: exact %.4f  rel<1e-4 %.4f  rel<1e-2 %.4f  max %.3e  mean gpu %.6f oracle %.6f"
          % (np.mean(f == fo), np.mean(rel < 1e-4), np.mean(rel < 1e-2), d.max(), f.mean(), fo.mean()))


# 1. no guiding at all: oneStepWalk path only
gi, r, tg, to = run(4, 0, mgd=(0, 0))
report("unguided (max guided depth 0)", gi, r, tg, to)
# 2. first training pass, no optimizer step: records must match
gi, r, tg, to = run(1, 1, min_batch=10 ** 9)
report("one pass, records only", gi, r, tg, to)
ts, to_ = gi.train_set(), r["train_set"]
print("  train set n gpu %d oracle %d" % (len(ts["xy"]), len(to_["xy"])))
if len(ts["xy"]) == len(to_["xy"]):
    for k in ts:
        a, b = ts[k].astype(np.float64), to_[k].astype(np.float64)
        print("   %-10s exact %.4f close %.4f" % (k, np.mean(a == b), np.mean(np.isclose(a, b, rtol=1e-4, atol=1e-6))))
# 3. frozen random network with peaked lobes
rng = np.random.default_rng(3)
n = o.net_n_params(cfg)
p = rng.uniform(-0.3, 0.3, n).astype(np.float32)
p[13312:] = rng.uniform(-1, 1, n - 13312).astype(np.float32)
gi, r, tg, to = run(8, 0, params=p)
report("frozen random network", gi, r, tg, to)
# 4. training end to end
gi, r, tg, to = run(32, 16)
report("training 16 + guiding 16", gi, r, tg, to)
ys = np.array([o.lib and 0 for _ in range(0)])
