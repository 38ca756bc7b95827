"""CPU tests of the oracle's solver loop: invariants the reference's design implies
 and analytic solutions of the Laplace problem, which pin the
pieces inferred at the snch-lbvh boundary (sign conventions, Neumann estimator)."""
import numpy as np
import pytest

from conftest import box_problem


def _solve(oracle, problem, w, h, spp, depth, eps, **kw):
    return oracle.solve(problem.as_dict(), w, h, spp, depth, eps, **kw)


def test_determinism_across_thread_counts(oracle, ladybug):
    a = _solve(oracle, ladybug, 32, 32, 4, 32, 1.0, threads=1, want_steps=True)
    b = _solve(oracle, ladybug, 32, 32, 4, 32, 1.0, threads=8, want_steps=True)
    assert np.array_equal(a["field"], b["field"]) and np.array_equal(a["steps"], b["steps"])
    assert a["walk_steps"] == b["walk_steps"] == int(a["steps"].sum())


def test_pixel_shards_concatenate_exactly(oracle, ladybug):
    full = _solve(oracle, ladybug, 32, 32, 4, 32, 1.0)
    parts = [_solve(oracle, ladybug, 32, 32, 4, 32, 1.0, pixel_begin=b, pixel_end=e)
             for b, e in ((0, 100), (100, 517), (517, 1024))]
    assert np.array_equal(np.concatenate([p["field"] for p in parts]), full["field"])
    assert sum(p["walk_steps"] for p in parts) == full["walk_steps"]


def test_depth_histogram_and_counts(oracle, ladybug):
    r = _solve(oracle, ladybug, 32, 32, 8, 16, 1.0, want_hist=True)
    h = r["depth_hist"]
    assert h[0] == 32 * 32 * 8 == r["walks_started"]
    assert np.all(np.diff(h.astype(np.int64)) <= 0)
    assert int(h.sum()) == r["walk_steps"]
    assert r["walks_absorbed"] + r["walks_truncated"] == r["walks_started"]
    assert r["walk_steps"] <= 32 * 32 * 8 * 16


def test_constant_colour_gives_absorption_probability(oracle, ladybug):
    # constant Dirichlet colour c=1 => field*spp = number of absorbed walks,
    # an exact integer because thp stays exactly 1.0f on the uniform path
    p = ladybug
    sd = p.as_dict()
    sd["d_colors"] = np.ones_like(p.d_colors)
    spp = 8
    r = oracle.solve(sd, 32, 32, spp, 24, 1.0)
    f = r["field"] * spp
    assert np.array_equal(f, np.round(f))
    assert np.all(r["field"] <= 1.0) and np.all(r["field"] >= 0.0)
    assert abs(float(f[:, 0].sum()) - r["walks_absorbed"]) < 0.5
    assert np.array_equal(r["field"][:, 0], r["field"][:, 1])


def test_mask_zeroes_pixels_and_keeps_the_rest(oracle, ladybug):
    full = _solve(oracle, ladybug, 16, 16, 4, 32, 1.0)
    sd = ladybug.as_dict()
    mask = np.ones(256, dtype=np.uint8)
    mask[::3] = 0
    sd["mask"] = mask
    r = oracle.solve(sd, 16, 16, 4, 32, 1.0)
    assert np.all(r["field"][mask == 0] == 0)
    assert np.array_equal(r["field"][mask == 1], full["field"][mask == 1])
    assert r["walks_started"] == int(mask.sum()) * 4


def test_max_depth_one_only_shell_pixels_contribute(oracle, ladybug):
    r = _solve(oracle, ladybug, 32, 32, 2, 1, 1.0, want_steps=True)
    assert r["walk_steps"] == 32 * 32 * 2
    assert np.all(r["steps"] == 2)


def test_laplace_dirichlet_linear_solution(oracle):
    # u = x is harmonic; Dirichlet data g = x on all four sides of [0,100]^2
    p = box_problem(0.0, 100.0, 25, d_sides=(0, 1, 2, 3), value=lambda x, y: x, probe=(40.0, 50.0, 50.0, 0.0, 1.0))
    w = 8
    r = _solve(oracle, p, w, w, 3000, 512, 0.25)
    xs = np.array([40.0 * (2 * (i % w) / w - 1) + 50.0 for i in range(w * w)])
    err = np.abs(r["field"][:, 0] - xs)
    assert err.mean() < 0.6 and err.max() < 2.5, (err.mean(), err.max())
    assert r["walks_truncated"] == 0


def test_laplace_mixed_zero_flux(oracle):
    # Dirichlet g = x on the left/right sides, zero-flux Neumann on top/bottom: u = x
    p = box_problem(0.0, 100.0, 25, d_sides=(1, 3), n_sides=(0, 2), value=lambda x, y: x,
                    probe=(40.0, 50.0, 50.0, 0.0, 1.0))
    w = 8
    r = _solve(oracle, p, w, w, 3000, 2048, 0.25)
    xs = np.array([40.0 * (2 * (i % w) / w - 1) + 50.0 for i in range(w * w)])
    err = np.abs(r["field"][:, 0] - xs)
    assert r["neumann_hits"] > 0
    assert err.mean() < 0.8 and err.max() < 3.5, (err.mean(), err.max())


def test_laplace_mixed_nonzero_flux_sign_convention(oracle):
    # u = y with Dirichlet g = y on left/right and Neumann data on top/bottom.  The reference
    # SUBTRACTS colour*G/alpha/pdf (integrator.cu:441-442), so the Neumann "colour" is the
    # derivative along the INWARD normal: -1 on the top side (y=hi), +1 on the bottom side.
    flux = lambda x, y, side: -1.0 if side == 2 else 1.0
    p = box_problem(0.0, 100.0, 25, d_sides=(1, 3), n_sides=(0, 2), value=lambda x, y: y, flux=flux,
                    probe=(40.0, 50.0, 50.0, 0.0, 1.0))
    w = 8
    r = _solve(oracle, p, w, w, 4000, 2048, 0.25)
    ys = np.array([40.0 * (2 * (i // w) / w - 1) + 50.0 for i in range(w * w)])
    err = np.abs(r["field"][:, 0] - ys)
    assert err.mean() < 1.5 and err.max() < 6.0, (err.mean(), err.max())


def test_libm_variant_agrees_statistically(oracle, oracle_libm, ladybug):
    # the literal std::cos/std::sin restatement and the deterministic-math oracle are different
    # trajectories of the same estimator: their fields agree within Monte-Carlo noise
    a = _solve(oracle, ladybug, 16, 16, 256, 64, 1.0)
    b = _solve(oracle_libm, ladybug, 16, 16, 256, 64, 1.0)
    rel = np.linalg.norm(a["field"] - b["field"]) / np.linalg.norm(a["field"])
    assert 0 < rel < 0.08, rel
    assert abs(a["walk_steps"] - b["walk_steps"]) / a["walk_steps"] < 0.03


def _poisson_disc(n_seg=256, cells_per_unit=16, f=(1.0, 0.5, 0.0)):
    """unit disc, u = 0 on the boundary, constant source f: u(r) = f (1 - r^2) / 4"""
    from elaina_amd import Problem
    t = np.linspace(0, 2 * np.pi, n_seg, endpoint=False)
    dv = np.stack([np.cos(t), np.sin(t)], 1).astype(np.float32)
    ds = np.stack([np.arange(n_seg), (np.arange(n_seg) + 1) % n_seg], 1).astype(np.int32)
    g = cells_per_unit
    n = 3 * g + 1
    rgb = np.ones((n, n, 3), np.float32) * np.asarray(f, np.float32)
    src = {"rgb": rgb, "index_scale": (g, g), "index_offset": (1.5 * g, 1.5 * g), "intensity": 1.0}
    return Problem(d_verts=dv, d_segs=ds, d_colors=np.zeros((n_seg, 6), np.float32), probe=(0.7, 0, 0, 0, 1), source=src)


def test_source_term_solves_the_poisson_equation(oracle):
    """sampleSource (reference integrator/uniform/integrator.cu:235-316): laplace(u) = -f; with a
    constant f on the unit disc and u = 0 on the rim, u = f (1 - r^2) / 4 -- fixes sign and scale"""
    p = _poisson_disc()
    w = h = 20
    r = oracle.solve(p.as_dict(), w, h, 192, 64, 1e-3)
    ys, xs = np.mgrid[0:h, 0:w]
    x, y = (xs * 2 / w - 1) * 0.7, (ys * 2 / h - 1) * 0.7
    want = (1 - x ** 2 - y ** 2) / 4
    f = r["field"].reshape(h, w, 3)
    assert abs(float(np.mean(f[..., 0] - want))) < 2e-3 and float(np.sqrt(np.mean((f[..., 0] - want) ** 2))) < 0.015
    np.testing.assert_allclose(f[..., 1], 0.5 * f[..., 0], rtol=1e-5, atol=1e-7)
    assert np.all(f[..., 2] == 0)
    # the SOURCE channel is the grid itself, bilinear, zero outside
    p2 = _poisson_disc()
    p2.probe = np.asarray((2.0, 0, 0, 0, 1), np.float32)
    img = oracle.render_source(p2.as_dict(), 16, 16).reshape(16, 16, 3)
    assert np.allclose(img[8, 8], (1.0, 0.5, 0.0)) and np.all(img[0, 0] == 0)


def test_source_term_bilinear_interpolation_and_intensity(oracle):
    from elaina_amd import Problem
    rgb = np.zeros((2, 3, 3), np.float32)
    rgb[0, :, 0] = (0, 1, 2)
    rgb[1, :, 0] = (10, 11, 12)
    src = {"rgb": rgb, "index_scale": (1, 1), "index_offset": (0, 0), "intensity": 2.0}
    p = Problem(d_verts=np.zeros((2, 2), np.float32), d_segs=np.zeros((0, 2), np.int32), probe=(1, 1, 0.5, 0, 1), source=src)
    # pixel centres of a 2x1 frame with this probe: x = 0 and 1, y = -0.5 ... use render at known points
    img = oracle.render_source(p.as_dict(), 4, 4).reshape(4, 4, 3)[..., 0]
    # eval point of pixel (px, py): (1 + (2px/4 - 1), 0.5 + (2py/4 - 1)) -> x in {0, .5, 1, 1.5}, y in {-.5, 0, .5, 1}
    assert img[1, 0] == 0.0 and img[1, 2] == 2.0 and img[1, 1] == 1.0          # y = 0 row, intensity 2
    assert img[2, 1] == 2.0 * (0.5 * 0.5 + 0.5 * 10.5)                          # (0.5, 0.5): mean of the four
    assert img[0, 1] == 2.0 * 0.5 * 0.5                                          # y = -0.5: half outside
