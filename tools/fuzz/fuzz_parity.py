"""random 2-D scenes, HIP against the oracle bit for bit (some seeds run as tests):
closed and open polylines of 3 .. 2000 segments on either boundary kind, emissive or not, degenerate and doubled
segments, scales from 1e-3 to 1e4, probes that look at the scene from far away, source terms"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
from oracle.oracle import Oracle


def polyline(rng, n, radius, centre, wobble, closed, jitter):
    t = np.sort(rng.uniform(0, 2 * np.pi, n)) if jitter else np.linspace(0, 2 * np.pi, n, endpoint=False)
    r = radius * (1.0 + wobble * np.sin(rng.integers(2, 9) * t + rng.uniform(0, 6)) + 0.3 * wobble * np.sin(rng.integers(9, 40) * t))
    v = np.stack([centre[0] + r * np.cos(t), centre[1] + r * np.sin(t)], 1)
    s = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1)
    if not closed:
        k = rng.integers(1, max(2, n // 4))
        s = np.roll(s, -rng.integers(0, n), axis=0)[:-k]
    return v, s


def random_problem(rng):
    feat = []
    scale = 10.0 ** rng.uniform(-3, 4)
    nd = int(rng.choice([3, 5, 40, 64, 65, 300, 2000]))
    nn = int(rng.choice([0, 4, 30, 64, 65, 200, 1500]))
    dv, ds = polyline(rng, nd, 0.3 * scale, (0.1 * scale, -0.05 * scale), rng.uniform(0, 0.3), rng.uniform() < 0.8, rng.uniform() < 0.5)
    dc = rng.uniform(0, 1, (len(dv), 6)).astype(np.float32)
    kw = dict(d_verts=dv, d_segs=ds, d_colors=dc)
    if rng.uniform() < 0.3:                       # degenerate and doubled segments
        feat.append('degenerate')
        ds2 = np.concatenate([ds, ds[:3], np.stack([ds[:2, 0], ds[:2, 0]], 1)])
        kw.update(d_segs=ds2)
    if nn:
        nv, ns = polyline(rng, nn, scale, (0.0, 0.0), rng.uniform(0, 0.25), rng.uniform() < 0.6, rng.uniform() < 0.5)
        nc = None
        if rng.uniform() < 0.5:
            feat.append('emissive')
            nc = (0.05 * rng.normal(size=(len(nv), 6))).astype(np.float32)
        kw.update(n_verts=nv, n_segs=ns, n_colors=nc)
    ang = rng.uniform(0, 2 * np.pi)
    view = scale * rng.choice([0.5, 1.1, 3.0, 50.0])
    kw["probe"] = (view, rng.uniform(-0.2, 0.2) * scale, rng.uniform(-0.2, 0.2) * scale, np.cos(ang), np.sin(ang))
    if rng.uniform() < 0.25:
        feat.append('source')
        g = rng.uniform(-1, 1, (9, 7, 3)).astype(np.float32)
        kw["source"] = {"rgb": g, "index_scale": (3.0 / scale, 4.0 / scale), "index_offset": (3.0, 4.0), "intensity": 0.1 * scale ** -2}
    feat.append('view %.3g' % (view / scale))
    if rng.uniform() < 0.3:
        # the whole scene far from the origin: coordinates 10 .. 300 scene sizes large, so the rounding of the coordinates themselves
        # (node records, probe, walk positions) is no longer small against the segments (found a ray / box miss in round 3)
        off = (scale * rng.choice([10.0, 100.0, 300.0]) * rng.uniform(0.5, 1.0, 2) * rng.choice([-1.0, 1.0], 2)).astype(np.float32)
        feat.append('offset %.3g %.3g' % (off[0] / scale, off[1] / scale))
        kw["d_verts"] = (kw["d_verts"] + off).astype(np.float32)
        if kw.get("n_verts") is not None:
            kw["n_verts"] = (kw["n_verts"] + off).astype(np.float32)
        pr = kw["probe"]
        kw["probe"] = (pr[0], pr[1] + float(off[0]), pr[2] + float(off[1]), pr[3], pr[4])
        if "source" in kw:
            so = kw["source"]
            so["index_offset"] = (so["index_offset"][0] - float(off[0]) * so["index_scale"][0], so["index_offset"][1] - float(off[1]) * so["index_scale"][1])
    return Problem(**kw), scale, feat


def main():
    first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40
    oracle = Oracle()
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        p, scale, feat = random_problem(rng)
        w, h, spp, depth = int(rng.choice([8, 24, 40])), int(rng.choice([8, 16, 24])), int(rng.choice([1, 3, 6, 19])), int(rng.choice([4, 24, 64]))
        eps = scale * 10.0 ** rng.uniform(-4, -1.5)
        if rng.uniform() < 0.2:
            feat.append('mask')
            p.mask = (rng.uniform(size=w * h) < 0.7).astype(np.uint8)
        it = UniformIntegrator(p, UniformIntegratorSettings((w, h), spp, depth, eps))
        if rng.uniform() < 0.3:
            feat.append('refill')
            it.set_option("refill", 1 if p.source is None else 0)
        if rng.uniform() < 0.4:            # launch shapes and scheduler constants: none of them may change a bit
            opts = {"steps_per_round": int(rng.choice([1, 3, 17, 256])), "block_size": int(rng.choice([64, 128, 256])),
                    "wait_weight": int(rng.choice([1, 8, 64])), "trav_burst": int(rng.choice([1, 3, 5])), "thin_waves": int(rng.choice([0, 1])),
                    "quad": int(rng.choice([0, 1])),       # (quad: four lanes per walker in every ordinary round)
                    # a Neumann mesh on the tree: its step queries per lane, by the wave, or by a wave whose pools are too small for a trip
                    "coop": int(rng.choice([0, 1, 1])), "pool_cap": int(rng.choice([96, 200, 384])), "ray_slot_trigger": int(rng.choice([1, 32, 64]))}
            feat.append(str(opts))
            for k, v in opts.items():
                it.set_option(k, v)
        if rng.uniform() < 0.35 and 'refill' not in feat:
            # the persistent first launch of round 6 on a frame a few resident blocks drain: pixels taken longest-first (or in queue order),
            # the hand-over with long remainders beside the rounds (every pixel long, none, a capped number, all thin or none), sorted or not
            popts = {"persist": 1, "resident_blocks": int(rng.choice([1, 2, 3, 7])), "persist_order": int(rng.choice([0, 1, 1])),
                     "long_steps": int(rng.choice([0, 8, 24, 64, 1024])), "long_cap": int(rng.choice([5, 64, 32768])), "long_thin": int(rng.choice([0, 3, 2048])),
                     "tail_sort": int(rng.choice([0, 1]))}
            feat.append(str(popts))
            for k, v in popts.items():
                it.set_option(k, v)
        it.solve()
        ref = oracle.solve(p.as_dict(), w, h, spp, depth, eps, threads=os.cpu_count() or 8)
        s = it.last_stats
        ok = all(s[k] == ref[k] for k in ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated", "neumann_hits")) and \
            np.array_equal(it.solution, ref["field"], equal_nan=True)
        if ok and rng.uniform() < 0.15 and (w % 8 == 0 and h % 8 == 0):
            # the tile-sharded solve of the multi-GPU path: the shards' fields add up to the field, bit for bit
            import torch
            n_sh = int(rng.choice([2, 3, 5]))
            acc = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
            for r in range(n_sh):
                part = torch.zeros_like(acc)
                it.solve_sharded(r, n_sh, part.data_ptr(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                acc += part
            feat.append("sharded x%d" % n_sh)
            ok = np.array_equal(acc.cpu().numpy().reshape(-1, 3), ref["field"], equal_nan=True)
        if not ok:
            bad += 1
            d = np.abs(it.solution - ref["field"])
            print("seed %d MISMATCH: scale %.3g D %d N %s frame %dx%d spp %d depth %d eps %.3g steps %d/%d maxdiff %s" % (
                seed, scale, len(p.d_segs), 0 if p.n_segs is None else len(p.n_segs), w, h, spp, depth, eps, s["walk_steps"], ref["walk_steps"],
                np.nanmax(d)), feat, {k: (s[k], ref[k]) for k in ("walks_absorbed", "walks_truncated", "neumann_hits") if s[k] != ref[k]},
                "pixels differing: %d of %d" % (int((d.max(axis=1) > 0).sum()), len(d)), flush=True)
        it.close()
    print("fuzz %d..%d: %d mismatches" % (first, first + count - 1, bad), flush=True)


if __name__ == "__main__":
    main()
