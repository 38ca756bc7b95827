"""random 3-D scenes, HIP against the oracle bit for bit : bumpy icospheres of 20 .. 1280 triangles on
either boundary kind, holes (boundary edges), emissive or not, scales 1e-3 .. 1e3, probes from afar, doubled and zero-area
triangles"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from elaina_amd import UniformIntegratorSettings
from elaina_amd.integrator3d import Problem3, UniformIntegrator3
from oracle.oracle import Oracle


def shell(rng, subdiv, radius, bump, holes, centre):
    V, T = bench.icosphere(subdiv, 1.0)
    V = V.astype(np.float64)
    V *= 1.0 + bump * np.sin(rng.integers(2, 6) * V[:, :1] + rng.uniform(0, 6)) * np.cos(rng.integers(2, 6) * V[:, 1:2])
    V = (V * radius + np.asarray(centre)).astype(np.float32)
    if holes and len(T) > 30:
        keep = np.ones(len(T), bool)
        keep[rng.choice(len(T), max(1, len(T) // 20), replace=False)] = False
        T = np.ascontiguousarray(T[keep])
    return V, T


def random_scene(rng):
    scale = 10.0 ** rng.uniform(-3, 3)
    feat = []
    dV, dT = shell(rng, int(rng.choice([0, 1, 2, 3])), 0.4 * scale, rng.uniform(0, 0.25), rng.uniform() < 0.3, (0.05 * scale, 0.0, -0.03 * scale))
    if rng.uniform() < 0.3:
        feat.append("degenerate")
        dT = np.concatenate([dT, dT[:4], np.stack([dT[:3, 0], dT[:3, 0], dT[:3, 1]], 1)]).astype(np.int32)
    sd = {"d_verts": dV, "d_tris": dT, "d_colors": rng.uniform(0, 1, (len(dV), 6)).astype(np.float32), "n_verts": None, "n_tris": None,
          "n_colors": None, "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
    if rng.uniform() < 0.7:
        nV, nT = shell(rng, int(rng.choice([1, 2, 3])), scale, rng.uniform(0, 0.2), rng.uniform() < 0.5, (0.0, 0.0, 0.0))
        sd["n_verts"], sd["n_tris"] = nV, nT
        sd["n_colors"] = np.zeros((len(nV), 6), np.float32)
        if rng.uniform() < 0.5:
            feat.append("emissive")
            sd["n_colors"] = (0.05 * rng.normal(size=(len(nV), 6))).astype(np.float32)
        feat.append("N %d" % len(nT))
    view = float(rng.choice([0.4, 0.9, 3.0, 40.0]))
    feat.append("view %g" % view)
    up = rng.normal(size=3); up /= np.linalg.norm(up)
    right = np.cross(up, rng.normal(size=3)); right /= np.linalg.norm(right)
    pos = rng.uniform(-0.1, 0.1, 3) * scale
    sd["offset"] = np.zeros(3, np.float32)
    if rng.uniform() < 0.3:
        # the whole scene far from the origin: the rounding of the coordinates is no longer small against the triangles
        off = (scale * rng.choice([10.0, 100.0, 300.0]) * rng.uniform(0.5, 1.0, 3) * rng.choice([-1.0, 1.0], 3)).astype(np.float32)
        feat.append("offset %.3g %.3g %.3g" % tuple(off / scale))
        sd["d_verts"] = (sd["d_verts"] + off).astype(np.float32)
        if sd["n_verts"] is not None:
            sd["n_verts"] = (sd["n_verts"] + off).astype(np.float32)
        pos = pos + off
        sd["offset"] = off
    sd["probe"] = (view * scale, tuple(float(x) for x in pos), tuple(up), tuple(right))
    return sd, scale, feat


def main():
    first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40
    oracle = Oracle()
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        sd, scale, feat = random_scene(rng)
        w, h, spp, depth = int(rng.choice([8, 16, 24])), int(rng.choice([8, 16])), int(rng.choice([1, 3])), int(rng.choice([4, 24, 64]))
        eps = scale * 10.0 ** rng.uniform(-3.5, -1.5)
        if rng.uniform() < 0.25:
            feat.append("source")
            isc = (2.5 / scale, 2.0 / scale, 1.5 / scale)
            sd["source"] = {"rgb": rng.uniform(-1, 1, (4, 5, 6, 3)).astype(np.float32), "index_scale": isc,
                            "index_offset": tuple(float(o - f * k) for o, f, k in zip((3.0, 2.5, 2.0), sd["offset"], isc)), "intensity": 0.1 * scale ** -2}
        if rng.uniform() < 0.2:
            feat.append("mask")
            sd["mask"] = (rng.uniform(size=w * h) < 0.7).astype(np.uint8)
        # how the tree queries are answered -- by the wave through its task pools (default), per lane, pools too small for a trip
        # (the waves then answer the old way), slot tasks served eagerly -- must not change a bit
        for k in ("WOST3_WAVE", "WOST3_COOP", "WOST3_POOL_CAP", "WOST3_RAY_TRIGGER", "WOST3_CP_TRIGGER"):
            os.environ.pop(k, None)
        if rng.uniform() < 0.5:
            knobs = {"WOST3_WAVE": str(int(rng.choice([0, 1, 1]))), "WOST3_COOP": str(int(rng.choice([0, 1, 1]))),
                     "WOST3_POOL_CAP": str(int(rng.choice([96, 200, 512]))), "WOST3_RAY_TRIGGER": str(int(rng.choice([1, 32]))),
                     "WOST3_CP_TRIGGER": str(int(rng.choice([1, 64])))}
            feat.append(str(knobs))
            os.environ.update(knobs)
        it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((w, h), spp, depth, eps))
        it.solve()
        ref = oracle.solve3(sd, w, h, spp, depth, eps, threads=os.cpu_count() or 8)
        s = it.last_stats
        ok = all(s[k] == ref[k] for k in ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated", "neumann_hits")) and \
            np.array_equal(it.solution.reshape(-1, 3), ref["field"], equal_nan=True)
        if not ok:
            bad += 1
            d = np.abs(it.solution.reshape(-1, 3) - ref["field"])
            print("seed %d MISMATCH: scale %.3g D %d frame %dx%d spp %d depth %d eps %.3g steps %d/%d maxdiff %s" % (
                seed, scale, len(sd["d_tris"]), w, h, spp, depth, eps, s["walk_steps"], ref["walk_steps"], np.nanmax(d)), feat,
                {k: (s[k], ref[k]) for k in ("walks_absorbed", "walks_truncated", "neumann_hits") if s[k] != ref[k]},
                "pixels differing: %d of %d" % (int((d.max(axis=1) > 0).sum()), len(d)), flush=True)
        it.close()
    print("fuzz3d %d..%d: %d mismatches" % (first, first + count - 1, bad), flush=True)


if __name__ == "__main__":
    main()
