"""Tree-sized Neumann meshes FAR from the origin -- the class of scene the ray / box bug of round 3 (c5eaca1) came from -- through
the kernels the earlier fuzzers hardly reached with it: the guided 2-D integrator, the uniform 3-D integrator and the guided 3-D
integrator, HIP against the oracle bit for bit (fields and counters).

  scene   a Dirichlet curve / bumpy sphere inside a Neumann boundary of 5 000 .. 30 000 segments (2-D) or 5 120 / 20 480
          triangles (3-D), closed or open (a run of primitives removed: boundary ends / edges are always silhouettes),
          emissive or not; the WHOLE scene shifted by 10 .. 300 scene sizes, so that the rounding of the coordinates themselves
          (node records, probe, walk positions) is no longer small against the primitives;
  solve   small frames and few samples (the oracle answers every query on the CPU), a frozen random network with pronounced
          lobes for the guided integrators (training on scenes of this size is the business of the other fuzzers).

usage: fuzz_far_trees.py <guided2d | uniform3d | guided3d> [first seed] [count]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.dirname(__file__))
import bench  # noqa: E402
from fuzz_parity import polyline  # noqa: E402
from oracle.oracle import Oracle, default_net_config, default_net_config3, guided_settings, guided_settings3  # noqa: E402

COUNTERS = ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated", "neumann_hits")


def offset_of(rng, scale, dims):
    return (scale * rng.choice([10.0, 30.0, 100.0, 300.0]) * rng.uniform(0.5, 1.0, dims) * rng.choice([-1.0, 1.0], dims)).astype(np.float32)


def scene2d(rng):
    scale = 10.0 ** rng.uniform(-1, 2)
    nn = int(rng.choice([5000, 8000, 15000, 30000]))
    closed = rng.uniform() < 0.6
    emissive = rng.uniform() < 0.5
    off = offset_of(rng, scale, 2)
    dv, ds = polyline(rng, int(rng.choice([40, 300])), 0.3 * scale, (0.1 * scale, -0.05 * scale), rng.uniform(0, 0.3), True, rng.uniform() < 0.5)
    nv, ns = polyline(rng, nn, scale, (0.0, 0.0), rng.uniform(0, 0.2), closed, rng.uniform() < 0.5)
    nc = (0.05 * rng.normal(size=(len(nv), 6))).astype(np.float32) if emissive else None
    ang = rng.uniform(0, 2 * np.pi)
    view = scale * float(rng.choice([0.9, 1.2]))
    kw = dict(d_verts=(dv + off).astype(np.float32), d_segs=ds, d_colors=rng.uniform(0, 1, (len(dv), 6)).astype(np.float32),
              n_verts=(nv + off).astype(np.float32), n_segs=ns, n_colors=nc,
              probe=(view, float(off[0] + rng.uniform(-0.1, 0.1) * scale), float(off[1] + rng.uniform(-0.1, 0.1) * scale), np.cos(ang), np.sin(ang)))
    feat = ["N %d%s%s" % (len(ns), "" if closed else " open", " emissive" if emissive else ""), "offset %.3g %.3g" % tuple(off / scale)]
    return kw, scale, off, feat


def rand_params(n, n_mlp, rng, wscale=0.3, gscale=1.0):
    p = rng.uniform(-wscale, wscale, n).astype(np.float32)
    p[n_mlp:] = rng.uniform(-gscale, gscale, n - n_mlp).astype(np.float32)
    return p


def guided2d(oracle, seed):
    from elaina_amd import Problem
    from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
    rng = np.random.default_rng(70_000 + seed)
    kw, scale, off, feat = scene2d(rng)
    p = Problem(**kw)
    w, h, spp, depth = 16, 12, int(rng.choice([2, 3])), int(rng.choice([12, 32]))
    eps = scale * 10.0 ** rng.uniform(-3.5, -2)
    aabb = ((float(off[0] - 1.3 * scale), float(off[1] - 1.3 * scale)), (float(off[0] + 1.3 * scale), float(off[1] + 1.3 * scale)))
    uf = float(rng.choice([0.0, 0.5]))
    st = GuidedIntegratorSettings(frameSize=(w, h), samplesPerPixel=spp, trainSppCount=0, maxWalkingDepth=depth, epsilonShell=eps,
                                  uniformFractionInTrainingPhase=uf, uniformFractionInGuidingPhase=uf)
    gi = GuidedIntegrator(p, st, aabb, seed=7)
    cfg = default_net_config()
    prm = rand_params(gi.network.n_params, gi.network.n_mlp_params, rng)
    gi.network.set_params(prm)
    gi.solve()
    gs = guided_settings(w, h, spp, depth, eps, aabb[0], aabb[1], train_spp_count=0, uniform_fraction=(uf, uf))
    ref = oracle.solve_guided(p.as_dict(), gs, cfg, prm.copy(), threads=os.cpu_count() or 8, dump_spp=-1)
    s = gi.last_stats
    keys = COUNTERS + ("guided_steps",)
    ok = all(s[k] == ref[k] for k in keys) and np.array_equal(gi.solution, ref["field"], equal_nan=True)
    gi.close()
    return ok, "scale %.3g frame %dx%d spp %d depth %d uf %g" % (scale, w, h, spp, depth, uf), feat, {k: (s[k], ref[k]) for k in keys if s[k] != ref[k]}


def scene3d(rng):
    scale = 10.0 ** rng.uniform(-1, 2)
    subdiv = int(rng.choice([4, 4, 5]))                    # 5 120 / 20 480 triangles
    closed = rng.uniform() < 0.6
    emissive = rng.uniform() < 0.5
    off = offset_of(rng, scale, 3)
    V, T = bench.icosphere(subdiv, 1.0)
    V = V.astype(np.float64)
    V *= 1.0 + rng.uniform(0, 0.15) * np.sin(rng.integers(2, 6) * V[:, :1] + rng.uniform(0, 6)) * np.cos(rng.integers(2, 6) * V[:, 1:2])
    if not closed:
        keep = np.ones(len(T), bool)
        keep[rng.choice(len(T), len(T) // 25, replace=False)] = False
        T = np.ascontiguousarray(T[keep])
    nV = (V * scale + off).astype(np.float32)
    dV, dT = bench.icosphere(int(rng.choice([1, 2])), 0.35 * scale)
    dV = (dV.astype(np.float64) + np.asarray([0.05, 0.0, -0.03]) * scale + off).astype(np.float32)
    sd = {"d_verts": dV, "d_tris": dT, "d_colors": rng.uniform(0, 1, (len(dV), 6)).astype(np.float32), "n_verts": nV, "n_tris": T,
          "n_colors": (0.05 * rng.normal(size=(len(nV), 6))).astype(np.float32) if emissive else np.zeros((len(nV), 6), np.float32),
          "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
    up = rng.normal(size=3)
    up /= np.linalg.norm(up)
    right = np.cross(up, rng.normal(size=3))
    right /= np.linalg.norm(right)
    pos = rng.uniform(-0.1, 0.1, 3) * scale + off
    sd["probe"] = (float(rng.choice([0.8, 1.1])) * scale, tuple(float(x) for x in pos), tuple(up), tuple(right))
    feat = ["N %d%s%s" % (len(T), "" if closed else " open", " emissive" if emissive else ""), "offset %.3g %.3g %.3g" % tuple(off / scale)]
    return sd, scale, off, feat


def uniform3d(oracle, seed):
    from elaina_amd import UniformIntegratorSettings
    from elaina_amd.integrator3d import Problem3, UniformIntegrator3
    rng = np.random.default_rng(80_000 + seed)
    sd, scale, off, feat = scene3d(rng)
    w, h, spp, depth = 12, 8, int(rng.choice([1, 2])), int(rng.choice([8, 24]))
    eps = scale * 10.0 ** rng.uniform(-3.5, -2)
    it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((w, h), spp, depth, eps))
    it.solve()
    ref = oracle.solve3(sd, w, h, spp, depth, eps, threads=os.cpu_count() or 8)
    s = it.last_stats
    ok = all(s[k] == ref[k] for k in COUNTERS) and np.array_equal(it.solution.reshape(-1, 3), ref["field"], equal_nan=True)
    it.close()
    return ok, "scale %.3g frame %dx%d spp %d depth %d" % (scale, w, h, spp, depth), feat, {k: (s[k], ref[k]) for k in COUNTERS if s[k] != ref[k]}


def guided3d(oracle, seed):
    from elaina_amd import capi
    from elaina_amd.guided import GuidedIntegratorSettings
    from elaina_amd.integrator3d import GuidedIntegrator3, Problem3
    rng = np.random.default_rng(90_000 + seed)
    sd, scale, off, feat = scene3d(rng)
    w, h, spp, depth = 10, 8, 2, int(rng.choice([8, 20]))
    eps = scale * 10.0 ** rng.uniform(-3.5, -2)
    aabb = (tuple(float(o - 1.3 * scale) for o in off), tuple(float(o + 1.3 * scale) for o in off))
    uf = float(rng.choice([0.0, 0.5]))
    cfg = default_net_config3(n_levels=4)                # four levels keep the dense 3-D grid small; the code path is the same for eight
    hip_cfg = capi.NetConfig(cfg.n_levels, cfg.n_features, cfg.base_resolution, cfg.per_level_scale, cfg.n_neurons, cfg.n_hidden_layers,
                             cfg.n_output, cfg.learning_rate, cfg.beta1, cfg.beta2, cfg.epsilon, cfg.l2_reg, cfg.ema_decay)
    st = GuidedIntegratorSettings(frameSize=(w, h), samplesPerPixel=spp, trainSppCount=0, maxWalkingDepth=depth, epsilonShell=eps,
                                  uniformFractionInTrainingPhase=uf, uniformFractionInGuidingPhase=uf)
    gi = GuidedIntegrator3(Problem3.from_dict(sd), st, aabb, network_config=hip_cfg, seed=7)
    prm = rand_params(gi.network.n_params, gi.network.n_mlp_params, rng)
    gi.network.set_params(prm)
    gi.solve()
    gs = guided_settings3(w, h, spp, depth, eps, aabb[0], aabb[1], train_spp_count=0, uniform_fraction=(uf, uf))
    ref = oracle.solve_guided3(sd, gs, cfg, prm.copy(), threads=os.cpu_count() or 8, dump_spp=-1)
    s = gi.last_stats
    keys = COUNTERS + ("guided_steps",)
    ok = all(s[k] == ref[k] for k in keys) and np.array_equal(gi.solution, ref["field"], equal_nan=True)
    gi.close()
    return ok, "scale %.3g frame %dx%d spp %d depth %d uf %g" % (scale, w, h, spp, depth, uf), feat, {k: (s[k], ref[k]) for k in keys if s[k] != ref[k]}


def main():
    """fuzz_far_trees.py MODE [FIRST [COUNT [SECONDS [LOG]]]]: seeds FIRST .. FIRST + COUNT - 1; with SECONDS > 0 the run stops at the
    first seed that starts after that many seconds (the summary line names the seeds that ran); LOG = a file that receives one line
    per seed as it finishes (a run that is cut off still leaves its record)"""
    import time
    mode = sys.argv[1]
    first, count = int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 40
    seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    log = open(sys.argv[5], "a") if len(sys.argv) > 5 else None
    run = {"guided2d": guided2d, "uniform3d": uniform3d, "guided3d": guided3d}[mode]
    oracle = Oracle()
    bad, done, t0 = 0, 0, time.time()
    for seed in range(first, first + count):
        if seconds > 0 and time.time() - t0 > seconds:
            break
        ok, what, feat, diff = run(oracle, seed)
        done += 1
        if log:
            log.write("%s seed %d %s %.1fs\n" % (mode, seed, "ok" if ok else "MISMATCH", time.time() - t0))
            log.flush()
        if not ok:
            bad += 1
            print("seed %d MISMATCH (%s): %s" % (seed, mode, what), feat, diff, flush=True)
    print("fuzz far trees %s %d..%d: %d mismatches" % (mode, first, first + done - 1, bad), flush=True)
    if log:
        log.write("fuzz far trees %s %d..%d: %d mismatches\n" % (mode, first, first + done - 1, bad))
        log.close()


if __name__ == "__main__":
    main()
