"""random 2-D scenes through the guided integrator (fp32 network, online training), HIP against the oracle bit for bit
: the scenes of fuzz_parity.py without sources at moderate scales, a few trained and a few guiding samples"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.dirname(__file__))
from fuzz_parity import polyline
from elaina_amd import Problem
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
from oracle.oracle import Oracle, default_net_config, guided_settings


def main():
    first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 20
    # "half": the half-precision network mode has no bit-exact oracle; its fused launch (the network inside the walk kernel)
    # must equal its own one-launch-per-depth path bit for bit instead
    half = len(sys.argv) > 3 and sys.argv[3] == "half"
    oracle = None if half else Oracle()
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(10_000 + seed)
        scale = 10.0 ** rng.uniform(-1, 2)
        nd, nn = int(rng.choice([4, 40, 65, 400])), int(rng.choice([0, 4, 30, 65, 300]))
        dv, ds = polyline(rng, nd, 0.3 * scale, (0.1 * scale, -0.05 * scale), rng.uniform(0, 0.3), rng.uniform() < 0.8, rng.uniform() < 0.5)
        kw = dict(d_verts=dv, d_segs=ds, d_colors=rng.uniform(0, 1, (len(dv), 6)).astype(np.float32))
        feat = ["D %d" % len(ds)]
        if nn:
            nv, ns = polyline(rng, nn, scale, (0.0, 0.0), rng.uniform(0, 0.25), rng.uniform() < 0.6, rng.uniform() < 0.5)
            nc = (0.05 * rng.normal(size=(len(nv), 6))).astype(np.float32) if rng.uniform() < 0.5 else None
            kw.update(n_verts=nv, n_segs=ns, n_colors=nc)
            feat.append("N %d%s" % (len(ns), " emissive" if nc is not None else ""))
        ang = rng.uniform(0, 2 * np.pi)
        view = scale * float(rng.choice([0.5, 1.1, 3.0]))
        kw["probe"] = (view, rng.uniform(-0.2, 0.2) * scale, rng.uniform(-0.2, 0.2) * scale, np.cos(ang), np.sin(ang))
        feat.append("view %g" % (view / scale))
        p = Problem(**kw)
        w, h = int(rng.choice([16, 24])), int(rng.choice([16, 24]))
        spp, train, depth = int(rng.choice([3, 5])), int(rng.choice([0, 2, 3])), int(rng.choice([8, 32]))
        eps = scale * 10.0 ** rng.uniform(-3.5, -2)
        aabb = ((-1.3 * scale, -1.3 * scale), (1.3 * scale, 1.3 * scale))
        uf = (float(rng.choice([0.0, 0.5, 1.0])), float(rng.choice([0.0, 0.5])))
        mgd = (int(rng.choice([10, 2])), int(rng.choice([10, 3, 0])))
        stride = int(rng.choice([1, 1, 2]))
        offset = int(rng.integers(0, stride))
        if rng.uniform() < 0.25:
            feat.append("mask")
            p.mask = (rng.uniform(size=w * h) < 0.75).astype(np.uint8)
        feat.append("mgd %s stride %d+%d" % (mgd, stride, offset))
        st = GuidedIntegratorSettings(frameSize=(w, h), samplesPerPixel=spp, trainSppCount=train, maxWalkingDepth=depth, epsilonShell=eps,
                                      uniformFractionInTrainingPhase=uf[0], uniformFractionInGuidingPhase=uf[1],
                                      maxGuidedDepthInTrainingPhase=mgd[0], maxGuidedDepthInGuidingPhase=mgd[1], batchSize=1024, minBatchSize=256,
                                      trainPixelStride=stride, trainPixelOffset=offset)
        if half:
            out = []
            for fused in ("1", "0"):
                os.environ["WOST_GUIDED_FUSED"] = fused
                gi = GuidedIntegrator(p, st, aabb, seed=7)
                gi.network.set_option("precision", 16)
                gi.network.set_option("train_precision", 16)
                gi.solve()
                out.append((gi.solution.copy(), dict(gi.last_stats), gi.network.params().copy()))
                gi.close()
            os.environ.pop("WOST_GUIDED_FUSED", None)
            keys = ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated", "neumann_hits", "guided_steps")
            ok = all(out[0][1][k] == out[1][1][k] for k in keys) and np.array_equal(out[0][0], out[1][0], equal_nan=True) and \
                np.array_equal(out[0][2], out[1][2])
            if not ok:
                bad += 1
                print("seed %d MISMATCH (half precision, fused against per-depth): scale %.3g frame %dx%d spp %d train %d depth %d" % (
                    seed, scale, w, h, spp, train, depth), feat, flush=True)
            continue
        gi = GuidedIntegrator(p, st, aabb, seed=7)
        p0 = gi.network.params()
        gi.solve()
        gs = guided_settings(w, h, spp, depth, eps, aabb[0], aabb[1], train_spp_count=train, uniform_fraction=uf, max_guided_depth=mgd,
                             batch_size=1024, min_batch_size=256, train_pixel_stride=stride, train_pixel_offset=offset)
        ref = oracle.solve_guided(p.as_dict(), gs, default_net_config(), p0.copy(), threads=os.cpu_count() or 8, dump_spp=-1)
        s = gi.last_stats
        keys = ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated", "neumann_hits", "guided_steps")
        ok = all(s[k] == ref[k] for k in keys) and np.array_equal(gi.solution, ref["field"], equal_nan=True)
        if not ok:
            bad += 1
            d = np.abs(gi.solution - ref["field"])
            print("seed %d MISMATCH: scale %.3g frame %dx%d spp %d train %d depth %d eps %.3g uf %s maxdiff %s" % (
                seed, scale, w, h, spp, train, depth, eps, uf, np.nanmax(d)), feat, {k: (s[k], ref[k]) for k in keys if s[k] != ref[k]},
                "pixels differing: %d of %d" % (int((d.max(axis=1) > 0).sum()), len(d)), flush=True)
        gi.close()
    print("fuzz guided %d..%d: %d mismatches" % (first, first + count - 1, bad), flush=True)


if __name__ == "__main__":
    main()
