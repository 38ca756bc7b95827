"""Tree-sized Neumann meshes FAR from the origin -- the class of scene the ray / box bug of round 3 (c5eaca1) came from -- through
the kernels the earlier fuzzers hardly reached with it: the guided 2-D integrator, the uniform 3-D integrator and the guided 3-D
integrator, HIP against the oracle bit for bit (fields and counters).

  scene   a Dirichlet curve / bumpy sphere inside a Neumann boundary of 5 000 .. 30 000 segments (2-D) or 5 120 / 20 480
          triangles (3-D), closed or open (a run of primitives removed: boundary ends / edges are always silhouettes),
          emissive or not; the WHOLE scene shifted by 10 .. 300 scene sizes, so that the rounding of the coordinates themselves
          (node records, probe, walk positions) is no longer small against the primitives;
  solve   small frames and few samples (the oracle answers every query on the CPU), a frozen random network with pronounced
          lobes for the guided integrators (training on scenes of this size is the business of the other fuzzers).

  trained the *_train modes run the guided integrators with trainSppCount >= 2 on the same class of scene: training records, the
          Adam / EMA steps and the trained network's walks (rows a26 - a27), the final parameters compared as well.

Every mode is a pair of halves over one `case` (scene + settings + initial parameters, all from the seed): oracle_run() needs no
GPU, hip_run() needs no oracle.  `golden` writes the oracle half of the seeds to tests/golden/far_trees_<mode>.npz in the build
container; tests/test_gpu_far_trees.py runs the HIP half on the GPU box against that file -- the same seeds, none of the oracle's
CPU time inside the GPU suite.

usage: fuzz_far_trees.py <mode> [first seed] [count [seconds [log]]]       both halves live (needs GPU + oracle)
       fuzz_far_trees.py golden <mode> [count]                             write the fixture (oracle only)
       fuzz_far_trees.py check <mode>                                      HIP against the fixture (GPU only)
modes: guided2d uniform3d guided3d guided2d_train guided3d_train guided3d_l8 guided3d_l8_train"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.dirname(__file__))
import bench  # noqa: E402
from fuzz_parity import polyline  # noqa: E402

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


def net_counts(cfg, dims):
    """(n_params, n_mlp_params) of the guiding network: the layout of oracle/wost_net.c layout_d restated on the host in fp32
    (checked against the library's own count on both sides: Oracle.net*_n_params when a fixture is written, wost_net_n_params
    when HIP runs)"""
    f32 = np.float32
    log2s = np.log2(f32(cfg.per_level_scale), dtype=f32)
    entries = 0
    for i in range(cfg.n_levels):
        scale = f32(np.exp2(f32(i) * log2s, dtype=f32) * f32(cfg.base_resolution)) - f32(1.0)
        res = int(np.ceil(scale)) + 1
        entries += (res ** dims + 7) // 8 * 8
    enc = cfg.n_levels * cfg.n_features
    n_mlp = cfg.n_neurons * enc + (cfg.n_hidden_layers - 1) * cfg.n_neurons * cfg.n_neurons + cfg.n_output_padded * cfg.n_neurons
    return n_mlp + entries * cfg.n_features, n_mlp


class NetCfg:
    """the network configuration as plain numbers (data/ladybug/n.json:49-81 of the reference + guided/parameters.h:16-33); both
    libraries' NetConfig structs are made from it, so neither half of a case imports the other's"""
    def __init__(self, n_output, n_levels=8):
        self.n_levels, self.n_features, self.base_resolution, self.per_level_scale = n_levels, 4, 8, 1.4049999713897705
        self.n_neurons, self.n_hidden_layers, self.n_output, self.n_output_padded = 64, 3, n_output, 48
        self.learning_rate, self.beta1, self.beta2 = 0.00800000037997961, 0.8999999761581421, 0.9900000095367432
        self.epsilon, self.l2_reg, self.ema_decay = 1.0000000036274937e-15, 9.999999974752427e-07, 0.949999988079071

    def oracle(self):
        from oracle.oracle import NetConfig
        return NetConfig(self.n_levels, self.n_features, self.base_resolution, self.per_level_scale, self.n_neurons, self.n_hidden_layers,
                         self.n_output, self.n_output_padded, self.learning_rate, self.beta1, self.beta2, self.epsilon, self.l2_reg,
                         self.ema_decay)

    def hip(self):
        from elaina_amd import capi
        return capi.NetConfig(self.n_levels, self.n_features, self.base_resolution, self.per_level_scale, self.n_neurons,
                              self.n_hidden_layers, self.n_output, self.learning_rate, self.beta1, self.beta2, self.epsilon, self.l2_reg,
                              self.ema_decay)


GUIDED_KEYS = COUNTERS + ("guided_steps",)
TRAIN_KEYS = GUIDED_KEYS + ("train_samples", "optimizer_steps")


def case_guided2d(seed, train=False):
    rng = np.random.default_rng((75_000 if train else 70_000) + seed)
    kw, scale, off, feat = scene2d(rng)
    c = dict(dims=2, scene=kw, scale=scale, feat=feat, w=16, h=12, spp=int(rng.choice([2, 3])), depth=int(rng.choice([12, 32])))
    c["eps"] = scale * 10.0 ** rng.uniform(-3.5, -2)
    c["aabb"] = ((float(off[0] - 1.3 * scale), float(off[1] - 1.3 * scale)), (float(off[0] + 1.3 * scale), float(off[1] + 1.3 * scale)))
    uf = float(rng.choice([0.0, 0.5]))
    c["uf"], c["train"], c["keys"], c["cfg"] = (uf, uf), 0, GUIDED_KEYS, NetCfg(33)
    if train:
        c.update(w=24, h=16, spp=int(rng.choice([3, 4])), train=int(rng.choice([2, 3])), keys=TRAIN_KEYS,
                 uf=(float(rng.choice([0.5, 1.0])), float(rng.choice([0.0, 0.5]))), batch=(512, 128))
    n, n_mlp = net_counts(c["cfg"], 2)
    c["params"] = rand_params(n, n_mlp, rng, *((0.15, 0.5) if train else (0.3, 1.0)))
    c["what"] = "scale %.3g frame %dx%d spp %d train %d depth %d uf %g %g" % (scale, c["w"], c["h"], c["spp"], c["train"], c["depth"], *c["uf"])
    return c


def case_uniform3d(seed):
    rng = np.random.default_rng(80_000 + seed)
    sd, scale, off, feat = scene3d(rng)
    c = dict(dims=3, scene=sd, scale=scale, feat=feat, w=12, h=8, spp=int(rng.choice([1, 2])), depth=int(rng.choice([8, 24])), keys=COUNTERS,
             params=None, train=0)
    c["eps"] = scale * 10.0 ** rng.uniform(-3.5, -2)
    c["what"] = "scale %.3g frame %dx%d spp %d depth %d" % (scale, c["w"], c["h"], c["spp"], c["depth"])
    return c


def case_guided3d(seed, train=False, n_levels=4):
    rng = np.random.default_rng((95_000 if train else 90_000) + seed + (0 if n_levels == 4 else 3_000))
    sd, scale, off, feat = scene3d(rng)
    c = dict(dims=3, scene=sd, scale=scale, feat=feat, w=10, h=8, spp=2, depth=int(rng.choice([8, 20])))
    c["eps"] = scale * 10.0 ** rng.uniform(-3.5, -2)
    c["aabb"] = (tuple(float(o - 1.3 * scale) for o in off), tuple(float(o + 1.3 * scale) for o in off))
    uf = float(rng.choice([0.0, 0.5]))
    # four levels keep the dense 3-D grid small and run the scalar network kernels with the launches per depth; eight levels (the
    # *_l8 modes, the reference's network) run the matrix-core kernels and, on these small frames, g3_fused_kernel: a sample in one launch
    c["uf"], c["train"], c["keys"], c["cfg"] = (uf, uf), 0, GUIDED_KEYS, NetCfg(41, n_levels=n_levels)
    if n_levels == 8:
        c.update(w=14, h=11, spp=3)
    if train:
        c.update(w=16, h=12, spp=int(rng.choice([3, 4])), train=int(rng.choice([2, 3])), keys=TRAIN_KEYS,
                 uf=(float(rng.choice([0.5, 1.0])), float(rng.choice([0.0, 0.5]))), batch=(256, 128))
    n, n_mlp = net_counts(c["cfg"], 3)
    c["params"] = rand_params(n, n_mlp, rng, *((0.15, 0.5) if train else (0.3, 1.0)))
    c["what"] = "scale %.3g frame %dx%d spp %d train %d depth %d uf %g %g" % (scale, c["w"], c["h"], c["spp"], c["train"], c["depth"], *c["uf"])
    return c


CASES = {"guided2d": case_guided2d, "uniform3d": case_uniform3d, "guided3d": case_guided3d,
         "guided2d_train": lambda seed: case_guided2d(seed, True), "guided3d_train": lambda seed: case_guided3d(seed, True),
         "guided3d_l8": lambda seed: case_guided3d(seed, False, 8), "guided3d_l8_train": lambda seed: case_guided3d(seed, True, 8)}
DEFAULT_COUNT = {"guided2d": 40, "uniform3d": 40, "guided3d": 40, "guided2d_train": 10, "guided3d_train": 10, "guided3d_l8": 24, "guided3d_l8_train": 8}


def _result(c, field, stats, params):
    """what the two halves are compared on: the field (bits), the counters, and for a trained case the final parameters"""
    r = {"field": np.ascontiguousarray(field, np.float32).reshape(-1, 3).copy(), "counters": np.asarray([stats[k] for k in c["keys"]], np.uint64)}
    if c["train"]:
        r["params"] = np.ascontiguousarray(params, np.float32).copy()
    return r


def oracle_run(oracle, c):
    from oracle.oracle import guided_settings, guided_settings3
    threads = os.cpu_count() or 8
    if c["params"] is None:
        ref = oracle.solve3(c["scene"], c["w"], c["h"], c["spp"], c["depth"], c["eps"], threads=threads)
        return _result(c, ref["field"], ref, None)
    cfg = c["cfg"].oracle()
    n = (oracle.net_n_params if c["dims"] == 2 else oracle.net3_n_params)(cfg)
    assert n == c["params"].size, (n, c["params"].size)
    kw = dict(train_spp_count=c["train"], uniform_fraction=c["uf"])
    if c["train"]:
        kw.update(batch_size=c["batch"][0], min_batch_size=c["batch"][1])
    prm = c["params"].copy()
    if c["dims"] == 2:
        from elaina_amd import Problem
        gs = guided_settings(c["w"], c["h"], c["spp"], c["depth"], c["eps"], c["aabb"][0], c["aabb"][1], **kw)
        ref = oracle.solve_guided(Problem(**c["scene"]).as_dict(), gs, cfg, prm, threads=threads, dump_spp=-1)
    else:
        gs = guided_settings3(c["w"], c["h"], c["spp"], c["depth"], c["eps"], c["aabb"][0], c["aabb"][1], **kw)
        ref = oracle.solve_guided3(c["scene"], gs, cfg, prm, threads=threads, dump_spp=-1)
    return _result(c, ref["field"], ref, prm)


def hip_run(c):
    from elaina_amd import UniformIntegratorSettings
    if c["params"] is None:
        from elaina_amd.integrator3d import Problem3, UniformIntegrator3
        it = UniformIntegrator3(Problem3.from_dict(c["scene"]), UniformIntegratorSettings((c["w"], c["h"]), c["spp"], c["depth"], c["eps"]))
        it.solve()
        r = _result(c, it.solution, it.last_stats, None)
        it.close()
        return r
    from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
    kw = dict(frameSize=(c["w"], c["h"]), samplesPerPixel=c["spp"], trainSppCount=c["train"], maxWalkingDepth=c["depth"], epsilonShell=c["eps"],
              uniformFractionInTrainingPhase=c["uf"][0], uniformFractionInGuidingPhase=c["uf"][1])
    if c["train"]:
        kw.update(batchSize=c["batch"][0], minBatchSize=c["batch"][1])
    st = GuidedIntegratorSettings(**kw)
    if c["dims"] == 2:
        from elaina_amd import Problem
        gi = GuidedIntegrator(Problem(**c["scene"]), st, c["aabb"], seed=7)
    else:
        from elaina_amd.integrator3d import GuidedIntegrator3, Problem3
        gi = GuidedIntegrator3(Problem3.from_dict(c["scene"]), st, c["aabb"], network_config=c["cfg"].hip(), seed=7)
    assert (gi.network.n_params, gi.network.n_mlp_params) == net_counts(c["cfg"], c["dims"])
    gi.network.set_params(c["params"])
    gi.solve()
    r = _result(c, gi.solution, gi.last_stats, gi.network.params() if c["train"] else None)
    gi.close()
    return r


def same(c, a, b):
    """the mismatch description of two results, or None"""
    diff = {k: (int(x), int(y)) for k, x, y in zip(c["keys"], a["counters"], b["counters"]) if x != y}
    if not np.array_equal(a["field"], b["field"], equal_nan=True):
        diff["field"] = int((a["field"].view(np.uint32) != b["field"].view(np.uint32)).sum())
    if c["train"] and not np.array_equal(a["params"], b["params"], equal_nan=True):
        diff["params"] = int((a["params"].view(np.uint32) != b["params"].view(np.uint32)).sum())
    return diff or None


def golden_path(mode):
    return os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "far_trees_%s.npz" % mode))


def write_golden(mode, count):
    """the oracle half of seeds 0 .. count-1 -> tests/golden/far_trees_<mode>.npz (fields are a few KB per seed; a trained case
    keeps its final parameters as 64-bit sums of their bit patterns per 1024-parameter block, plus the first 256 values)"""
    import time
    from oracle.oracle import Oracle
    oracle = Oracle()
    out = {"count": np.asarray(count), "mode": np.asarray(mode)}
    t0 = time.time()
    for seed in range(count):
        c = CASES[mode](seed)
        r = oracle_run(oracle, c)
        out["field_%d" % seed], out["counters_%d" % seed] = r["field"], r["counters"]
        if c["train"]:
            out["params_sums_%d" % seed], out["params_head_%d" % seed] = param_sums(r["params"]), r["params"][:256].copy()
        print("%s seed %d: %s %s counters %s (%.1fs)" % (mode, seed, c["what"], c["feat"], r["counters"].tolist(), time.time() - t0), flush=True)
    np.savez_compressed(golden_path(mode), **out)
    print("wrote %s (%d bytes)" % (golden_path(mode), os.path.getsize(golden_path(mode))))


def param_sums(p):
    bits = np.ascontiguousarray(p, np.float32).view(np.uint32).astype(np.uint64)
    pad = (-len(bits)) % 1024
    return np.concatenate([bits, np.zeros(pad, np.uint64)]).reshape(-1, 1024).sum(1)


def check_golden(mode, seeds=None):
    """the HIP half of the fixture's seeds against the fixture: the list of (seed, what, features, differences)"""
    g = np.load(golden_path(mode))
    bad = []
    for seed in (range(int(g["count"])) if seeds is None else seeds):
        c = CASES[mode](seed)
        r = hip_run(c)
        ref = {"field": g["field_%d" % seed], "counters": g["counters_%d" % seed]}
        diff = {k: (int(x), int(y)) for k, x, y in zip(c["keys"], r["counters"], ref["counters"]) if x != y}
        if not np.array_equal(r["field"], ref["field"], equal_nan=True):
            diff["field"] = int((r["field"].view(np.uint32) != ref["field"].view(np.uint32)).sum())
        if c["train"] and not (np.array_equal(param_sums(r["params"]), g["params_sums_%d" % seed]) and
                               np.array_equal(r["params"][:256], g["params_head_%d" % seed], equal_nan=True)):
            diff["params"] = int((param_sums(r["params"]) != g["params_sums_%d" % seed]).sum())
        if diff:
            bad.append((seed, c["what"], c["feat"], diff))
    return bad, int(g["count"]) if seeds is None else len(list(seeds))


def main():
    """fuzz_far_trees.py MODE [FIRST [COUNT [SECONDS [LOG]]]]: seeds FIRST .. FIRST + COUNT - 1; with SECONDS > 0 the run stops at the
    first seed that starts after that many seconds (the summary line names the seeds that ran); LOG = a file that receives one line
    per seed as it finishes (a run that is cut off still leaves its record)"""
    import time
    mode = sys.argv[1]
    if mode == "golden":
        return write_golden(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else DEFAULT_COUNT[sys.argv[2]])
    if mode == "check":
        bad, n = check_golden(sys.argv[2])
        for b in bad:
            print("seed %d MISMATCH (%s): %s" % (b[0], sys.argv[2], b[1]), b[2], b[3], flush=True)
        print("fuzz far trees %s against the fixture, %d seeds: %d mismatches" % (sys.argv[2], n, len(bad)), flush=True)
        return
    from oracle.oracle import Oracle
    first, count = int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else DEFAULT_COUNT[mode]
    seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    log = open(sys.argv[5], "a") if len(sys.argv) > 5 else None
    oracle = Oracle()
    bad, done, t0 = 0, 0, time.time()
    for seed in range(first, first + count):
        if seconds > 0 and time.time() - t0 > seconds:
            break
        c = CASES[mode](seed)
        diff = same(c, hip_run(c), oracle_run(oracle, c))
        done += 1
        if log:
            log.write("%s seed %d %s %.1fs\n" % (mode, seed, "ok" if diff is None else "MISMATCH", time.time() - t0))
            log.flush()
        if diff is not None:
            bad += 1
            print("seed %d MISMATCH (%s): %s" % (seed, mode, c["what"]), c["feat"], diff, flush=True)
    print("fuzz far trees %s %d..%d: %d mismatches" % (mode, first, first + done - 1, bad), flush=True)
    if log:
        log.write("fuzz far trees %s %d..%d: %d mismatches\n" % (mode, first, first + done - 1, bad))
        log.close()


if __name__ == "__main__":
    main()
