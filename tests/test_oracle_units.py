"""CPU tests of the oracle's building blocks against the known-answer vectors that pin them
(SURVEY.md 8c): canonical PCG32, per-pixel seeding, deterministic math accuracy, and the
geometric queries against O(N) brute force / analytic answers."""
import math

import numpy as np
import pytest


def test_pcg32_canonical_kat(oracle):
    # the reference generator is the canonical pcg32 (core/sampler.h:20-27,65-72):
    # pcg32_srandom(42, 54) known-answer vector from the PCG distribution
    r = oracle.pcg_seed(42, 54)
    got = [oracle.pcg_uint(r) for _ in range(6)]
    assert got == [0xA15C02B7, 0x7B47F409, 0xBA1D3330, 0x83D2F293, 0xBFA4784B, 0xCBED606E]


@pytest.mark.parametrize("pid,width,state,floats", [
    (0, 1024, 0x5851F42D4C957F2E, (0.8935741186, 0.2172300816, 0.3605147600)),
    (1, 1024, 0xC5AC726072C6F55B, (0.5327365398, 0.7743020058, 0.9744403362)),
    (389, 128, 0xC1CB8C682C9F63ED, (0.4668478966, 0.5740761757, 0.7844222784)),
])
def test_pixel_seeding_vectors(oracle, pid, width, state, floats):
    # prepareSolve seeding (integrator.cu:73-76 + hash.h:13-28 + sampler.h:46-62)
    r = oracle.pcg_seed_pixel(pid, width)
    assert r.state == state and r.inc == 1
    for f in floats:
        assert abs(oracle.pcg_float(r) - f) < 5e-10


def test_pcg_advance_matches_stepping(oracle):
    a = oracle.pcg_seed(7, 0)
    b = oracle.pcg_seed(7, 0)
    for _ in range(1000):
        oracle.pcg_uint(a)
    oracle.pcg_advance(b, 1000)
    assert a.state == b.state


def test_pcg_float_range_and_double(oracle):
    r = oracle.pcg_seed(1, 0)
    xs = np.array([oracle.pcg_float(r) for _ in range(20000)])
    assert xs.min() >= 0.0 and xs.max() < 1.0
    assert abs(xs.mean() - 0.5) < 0.01
    d = oracle.pcg_double(r)
    assert 0.0 <= d < 1.0


def test_interleave(oracle):
    assert oracle.lib.wo_interleave_32bit(0xFFFF, 0) == 0x55555555
    assert oracle.lib.wo_interleave_32bit(0, 0xFFFF) == 0xAAAAAAAA
    assert oracle.lib.wo_interleave_32bit(5, 3) == 0b011011


def test_sincos_accuracy(oracle):
    ks = np.unique(np.concatenate([np.arange(0, 1 << 23, 997), np.arange(0, 4096), (1 << 23) - 1 - np.arange(0, 4096),
                                   np.arange(8) * (1 << 20) + np.array([0, 1, -1, 5, 0, 1, 2, 3])]))
    ks = ks[(ks >= 0) & (ks < (1 << 23))]
    worst = 0.0
    for k in ks[::7]:
        u = float(np.float32(k / float(1 << 23)))
        c, s = oracle.sincos_2pi(u)
        tc, ts = math.cos(2 * math.pi * u), math.sin(2 * math.pi * u)
        worst = max(worst, abs(c - tc), abs(s - ts))
    assert worst < 2.0e-7


def test_sincos_exact_axes(oracle):
    assert oracle.sincos_2pi(0.0) == (1.0, 0.0)
    assert oracle.sincos_2pi(0.25) == (0.0, 1.0) or oracle.sincos_2pi(0.25) == (-0.0, 1.0)
    c, s = oracle.sincos_2pi(0.5)
    assert c == -1.0 and abs(s) == 0.0
    c, s = oracle.sincos_2pi(0.125)
    assert abs(c - s) < 1.5e-7 and abs(c - math.sqrt(0.5)) < 1.5e-7


def test_logf_accuracy(oracle):
    xs = np.concatenate([np.logspace(-30, 30, 2000), np.linspace(0.5, 2.0, 2000), [1.0, 1.0000001, 0.9999999]])
    for x in xs:
        xf = float(np.float32(x))
        got = oracle.logf(xf)
        ref = math.log(xf)
        assert abs(got - ref) <= 4e-7 * max(1.0, abs(ref)), (xf, got, ref)
    assert oracle.logf(1.0) == 0.0


def _random_mesh(rng, n, scale=100.0, seg_len=1.0):
    a = rng.uniform(0, scale, size=(n, 2))
    d = rng.normal(0, seg_len, size=(n, 2))
    verts = np.concatenate([a, a + d]).astype(np.float32)
    segs = np.stack([np.arange(n), np.arange(n) + n], axis=1).astype(np.int32)
    return verts, segs


def test_closest_point_bvh_equals_brute_force(oracle):
    rng = np.random.default_rng(0)
    verts, segs = _random_mesh(rng, 3000)
    pts = rng.uniform(-50, 150, size=(4000, 2)).astype(np.float32)
    b = oracle.closest_point(verts, segs, pts, mode=0)
    t = oracle.closest_point(verts, segs, pts, mode=1)
    for x, y in zip(b, t):
        assert np.array_equal(x, y)


def test_closest_point_ties_pick_lowest_index(oracle):
    # duplicated segments + symmetric configurations: exact ties must resolve to the lowest index
    verts = np.array([[0, 0], [1, 0], [0, 0], [1, 0], [0, 2], [1, 2]], dtype=np.float32)
    segs = np.array([[2, 3], [0, 1], [4, 5]], dtype=np.int32)
    pts = np.array([[0.5, 1.0], [0.5, 0.25], [0.5, 1.75]], dtype=np.float32)
    for mode in (0, 1):
        idx, dist, uv, side = oracle.closest_point(verts, segs, pts, mode=mode)
        assert list(idx) == [0, 0, 2]
        assert np.allclose(dist, [1.0, 0.25, 0.25])
        assert np.allclose(uv, 0.5)
        assert list(side) == [1, 1, -1]


def test_closest_point_analytic(oracle):
    verts = np.array([[0, 0], [2, 0]], dtype=np.float32)
    segs = np.array([[0, 1]], dtype=np.int32)
    pts = np.array([[1, 1], [-1, 0], [3, 4], [1, -2]], dtype=np.float32)
    idx, dist, uv, side = oracle.closest_point(verts, segs, pts, mode=0)
    assert np.allclose(dist, [1, 1, math.sqrt(17), 2])
    assert np.allclose(uv, [0.5, -0.5, 1.5, 0.5])
    assert list(side) == [1, 0, 1, -1]


def test_closest_point_ladybug_subset(oracle, ladybug):
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-90, 590, size=(300, 2)),
                          ladybug.d_verts[rng.integers(0, len(ladybug.d_verts), 300)] + rng.normal(0, 0.3, (300, 2))])
    b = oracle.closest_point(ladybug.d_verts, ladybug.d_segs, pts.astype(np.float32), mode=0)
    t = oracle.closest_point(ladybug.d_verts, ladybug.d_segs, pts.astype(np.float32), mode=1)
    for x, y in zip(b, t):
        assert np.array_equal(x, y)


def test_silhouette_box_inside_has_none(oracle, ladybug):
    # a convex CCW box seen from inside has no silhouette vertex
    rng = np.random.default_rng(5)
    pts = rng.uniform(-89, 589, size=(500, 2)).astype(np.float32)
    d = oracle.closest_silhouette(ladybug.n_verts, ladybug.n_segs, pts)
    assert np.all(np.isinf(d))


def test_silhouette_box_outside_sees_corners(oracle, ladybug):
    # from outside, beyond one side, the two corners of that side are silhouettes
    pts = np.array([[250, -200], [700, 250], [-300, -300]], dtype=np.float32)
    d = oracle.closest_silhouette(ladybug.n_verts, ladybug.n_segs, pts)
    assert abs(d[0] - math.hypot(340, 110)) < 1e-3
    assert abs(d[1] - math.hypot(110, 340)) < 1e-3
    # diagonal view of a corner: both incident faces are front facing -> that corner is not a
    # silhouette, its two neighbours are
    assert abs(d[2] - math.hypot(210, 890)) < 1e-2


def test_silhouette_open_polyline_endpoints(oracle):
    verts = np.array([[0, 0], [1, 0], [2, 0]], dtype=np.float32)
    segs = np.array([[0, 1], [1, 2]], dtype=np.int32)
    pts = np.array([[1, 1], [-1, 0.5]], dtype=np.float32)
    d = oracle.closest_silhouette(verts, segs, pts)
    # interior vertex of a straight line is never a silhouette; the open ends always are
    assert abs(d[0] - math.sqrt(2)) < 1e-6
    assert abs(d[1] - math.hypot(1, 0.5)) < 1e-6
    # bounded search radius
    d = oracle.closest_silhouette(verts, segs, pts, rmax=np.array([1.0, 5.0], dtype=np.float32))
    assert np.isinf(d[0]) and np.isfinite(d[1])


def test_ray_intersect_box(oracle, ladybug):
    o = np.array([[250, 250], [250, 250], [250, 250], [0, 0]], dtype=np.float32)
    d = np.array([[1, 0], [0, -1], [1, 0], [-1, 0]], dtype=np.float32)
    tmax = np.array([1000, 1000, 100, 1000], dtype=np.float32)
    hit, t, idx = oracle.ray_intersect(ladybug.n_verts, ladybug.n_segs, o, d, tmax)
    assert list(hit) == [1, 1, 0, 1]
    assert t[0] == 340.0 and idx[0] == 2      # right edge  (l 2 4)
    assert t[1] == 340.0 and idx[1] == 1      # bottom edge (l 1 2)
    assert t[3] == 90.0 and idx[3] == 0       # left edge   (l 3 1)


def test_eval_point_matches_survey_formula(oracle, ladybug):
    # pixel (px,py) -> (250 - s*ndc.y, 250 + s*ndc.x), s = 250
    import ctypes as C
    sc = oracle.make_scene(ladybug.as_dict())
    x, y = C.c_float(), C.c_float()
    for px, py in ((0, 0), (1023, 0), (512, 512), (100, 900)):
        oracle.lib.wo_eval_point(C.byref(sc), px, py, 1024, 1024, C.byref(x), C.byref(y))
        ndcx, ndcy = 2 * px / 1024 - 1, 2 * py / 1024 - 1
        assert abs(x.value - (250 - 250 * ndcy)) < 1e-3 and abs(y.value - (250 + 250 * ndcx)) < 1e-3


def test_unit_throughput_identity():
    # thp' = thp / pdf / alpha / 2pi (integrator.cu:521) is exactly 1.0f for
    # thp = 1 on both branches of the uniform path, with the constants of krrmath/constants.h
    # and M_PI taken from <cmath> as a double.  The HIP kernel relies on this identity.
    f = np.float32
    two_pi = f(6.28318530717958647693)
    sphere_pdf = f(1.0) / two_pi
    hemi_pdf = f(np.float64(1.0) / np.float64(3.14159265358979323846))
    assert f(f(f(1.0) / sphere_pdf) / f(1.0)) / two_pi == f(1.0)
    assert f(f(f(1.0) / hemi_pdf) / f(0.5)) / two_pi == f(1.0)


def test_pcg_skip2_constants(oracle):
    # two discarded draws == one LCG jump x -> M^2 x + (M+1) inc (used by the HIP kernel when
    # the Neumann boundary does not emit)
    M = 0x5851F42D4C957F2D
    mask = (1 << 64) - 1
    r = oracle.pcg_seed(123, 0)
    s0, inc = r.state, r.inc
    oracle.pcg_uint(r)
    oracle.pcg_uint(r)
    assert r.state == ((s0 * ((M * M) & mask)) + ((M + 1) & mask) * inc) & mask
