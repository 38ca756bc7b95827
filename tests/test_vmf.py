"""von Mises-Fisher lobe on the sphere (reference util/vmf.h), the distribution layer of the 3-D guided integrator
(`vmf.h` = 3D only, next; the integrator itself is not built): the oracle against the closed forms of the
distribution, the HIP entry points against the oracle."""
import math

import numpy as np
import pytest


def test_density_is_normalised_and_matches_the_closed_form(oracle):
    # int over the sphere = 2 pi int_{-1}^{1} f(c) dc = 1; f(1) = kappa / (2 pi (1 - exp(-2 kappa)))
    c = np.linspace(-1.0, 1.0, 400001)
    for kappa in (1e-6, 0.3, 1.45, 12.0, 80.0):
        f = oracle.vmf_eval(np.full(len(c), kappa, np.float32), c.astype(np.float32)).astype(np.float64)
        assert 2.0 * math.pi * np.trapezoid(f, c) == pytest.approx(1.0, rel=2e-4), kappa
        peak = 1.0 / (4.0 * math.pi) if kappa < 1e-5 else kappa / (2.0 * math.pi * (1.0 - math.exp(-2.0 * kappa)))
        assert f[-1] == pytest.approx(peak, rel=1e-5)
        if kappa >= 1e-5:
            # exp(kappa (c - 1)) kappa / (2 pi (1 - exp(-2 kappa))): the usual kappa exp(kappa c) / (4 pi sinh kappa)
            ref = kappa * np.exp(kappa * c[::50000]) / (4.0 * math.pi * math.sinh(kappa))
            assert np.allclose(f[::50000], ref, rtol=2e-5)


def test_a_cosine_beyond_one_is_clamped_and_tiny_kappa_is_uniform(oracle):
    # min(0, cosTheta - 1): rounding of a dot product of unit vectors must not raise the density above its peak
    assert oracle.vmf_eval([5.0, 5.0], [1.0, 1.0000002])[1] == oracle.vmf_eval([5.0, 5.0], [1.0, 1.0000002])[0]
    assert np.all(oracle.vmf_eval([0.0, 9e-6], [0.3, -1.0]) == np.float32(1.0 / (4.0 * math.pi)))


@pytest.mark.parametrize("kappa", [0.0, 0.5, 1.45, 20.0, 300.0])
def test_samples_follow_the_distribution(oracle, kappa):
    """unit vectors; mean direction mu with resultant length coth(kappa) - 1/kappa; the cosine to mu has the cdf
    (exp(kappa (c - 1)) - exp(-2 kappa)) / (1 - exp(-2 kappa))"""
    rng = np.random.default_rng(3)
    mu = rng.normal(size=3)
    mu /= np.linalg.norm(mu)
    n = 40000
    d = oracle.vmf_sample(np.full(n, kappa, np.float32), np.tile(mu.astype(np.float32), (n, 1)), rng.integers(0, 2**62, n).astype(np.uint64), 4)
    d = d.reshape(-1, 3).astype(np.float64)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=2e-6)
    mean = d.mean(axis=0)
    a = 0.0 if kappa == 0.0 else 1.0 / math.tanh(kappa) - 1.0 / kappa
    assert np.linalg.norm(mean - a * mu) < 4.0 / math.sqrt(len(d))
    c = np.sort(d @ mu)
    if kappa > 0.0:
        cdf = (np.exp(kappa * (c - 1.0)) - math.exp(-2.0 * kappa)) / (1.0 - math.exp(-2.0 * kappa))
    else:
        cdf = 0.5 * (c + 1.0)
    emp = (np.arange(len(c)) + 0.5) / len(c)
    assert np.abs(cdf - emp).max() < 1.7 / math.sqrt(len(c))         # Kolmogorov-Smirnov at ~1 %
    # the azimuth about mu is uniform: no preferred side
    t = np.cross(mu, [1.0, 0.0, 0.0])
    t /= np.linalg.norm(t)
    assert abs((d @ t).mean()) < 4.0 / math.sqrt(len(d))


def test_the_stream_advances_by_two_draws_per_sample(oracle):
    seed = np.array([11, 12], np.uint64)
    mu = np.array([[0, 0, 1], [0, 1, 0]], np.float32)
    one = oracle.vmf_sample([2.0, 0.0], mu, seed, 1)
    three = oracle.vmf_sample([2.0, 0.0], mu, seed, 3)
    assert np.array_equal(one[:, 0], three[:, 0]) and not np.array_equal(three[:, 0], three[:, 1])


@pytest.mark.gpu
def test_hip_vmf_matches_oracle(oracle):
    from elaina_amd import integrator3d
    rng = np.random.default_rng(8)
    n = 50000
    kappa = np.exp(rng.uniform(-14, 7, n)).astype(np.float32)
    kappa[:100] = 0.0
    cos_t = rng.uniform(-1.0, 1.0000002, n).astype(np.float32)
    assert np.array_equal(integrator3d.vmf_eval(kappa, cos_t), oracle.vmf_eval(kappa, cos_t))
    mu = rng.normal(size=(n, 3)).astype(np.float32)
    mu /= np.linalg.norm(mu, axis=1, keepdims=True)
    mu[:300] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 300)] * rng.choice([-1.0, 1.0], (300, 1)).astype(np.float32)
    seed = rng.integers(0, 2**62, n).astype(np.uint64)
    assert np.array_equal(integrator3d.vmf_sample(kappa, mu, seed, 3), oracle.vmf_sample(kappa, mu, seed, 3))


# ---- VMM<3,8>: the mixture of eight lobes (integrator/guided/distribution.h:279-436, train.h:492-553 with common3d) ----
def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _random_vmm3(rng, n):
    raw = rng.normal(0, 1.5, size=(n, 40)).astype(np.float32)
    raw[:, 1::5] = rng.uniform(-3, 5, size=(n, 8))      # log kappa
    wi = _unit(rng.normal(size=(n, 3))).astype(np.float32)
    return raw, wi


def _sphere_quadrature(n_c=1200, n_p=600):
    c = (np.arange(n_c) + 0.5) * (2.0 / n_c) - 1.0
    p = (np.arange(n_p) + 0.5) * (2.0 * math.pi / n_p)
    cc, pp = np.meshgrid(c, p, indexing="ij")
    s = np.sqrt(1.0 - cc * cc)
    return np.stack([s * np.cos(pp), s * np.sin(pp), cc], -1).reshape(-1, 3).astype(np.float32), (2.0 / n_c) * (2.0 * math.pi / n_p)


def test_mixture_density_integrates_to_one(oracle):
    rng = np.random.default_rng(0)
    raw, _ = _random_vmm3(rng, 3)
    raw[:, 1::5] = np.clip(raw[:, 1::5], -3, 3.5)          # lobes the quadrature can resolve
    wi, dw = _sphere_quadrature()
    for k in range(3):
        pdf, _ = oracle.vmm3_pdf_sample(np.tile(raw[k], (len(wi), 1)), wi, np.zeros(len(wi), np.uint64))
        assert pdf.astype(np.float64).sum() * dw == pytest.approx(1.0, rel=2e-3)


def test_mixture_samples_have_the_mean_of_the_mixture(oracle):
    """E[w] = sum_i weight_i (coth kappa_i - 1/kappa_i) mu_i; one draw picks the lobe, two more the direction"""
    rng = np.random.default_rng(1)
    raw, _ = _random_vmm3(rng, 1)
    n = 200000
    _, d = oracle.vmm3_pdf_sample(np.tile(raw[0], (n, 1)), np.zeros((n, 3), np.float32), rng.integers(0, 2**62, n).astype(np.uint64))
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=2e-6)
    lam = np.exp(np.clip(raw[0, 0::5].astype(np.float64), -10, 15))
    kap = np.exp(np.clip(raw[0, 1::5].astype(np.float64), -10, 15))
    mu = _unit(raw[0].reshape(8, 5)[:, 2:].astype(np.float64))
    a = 1.0 / np.tanh(kap) - 1.0 / kap
    mean = ((lam / lam.sum() * a)[:, None] * mu).sum(0)
    assert np.linalg.norm(d.astype(np.float64).mean(0) - mean) < 4.0 / math.sqrt(n)


def test_a_zero_mean_vector_keeps_the_density_finite(oracle):
    """Eigen normalized() leaves a zero vector as it is: every cosine is 0 and the density stays finite (a direction
    sampled about a zero mean has no frame -- frameFromNormal divides by a zero length -- and is not asserted)"""
    rng = np.random.default_rng(2)
    raw, wi = _random_vmm3(rng, 50)
    for c in (2, 3, 4):
        raw[:, c::5] = 0.0
    pdf, _ = oracle.vmm3_pdf_sample(raw, wi, rng.integers(0, 2**62, 50).astype(np.uint64))
    assert np.isfinite(pdf).all() and (pdf > 0).all()


def _random_training_batch3(rng, n):
    raw = rng.normal(0, 1, size=(n, 41)).astype(np.float32)
    raw[:, 1:40:5] = rng.uniform(0.2, 4, size=(n, 8))       # kappa >= 1.2: below 1 the reference uses a fitted parabola
    dirs = _unit(rng.normal(size=(n, 3))).astype(np.float32)
    li = rng.uniform(0.0, 1.0, n).astype(np.float32)
    dir_pdf = rng.uniform(0.05, 0.6, n).astype(np.float32)
    on_n = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    normal = _unit(rng.normal(size=(n, 3))).astype(np.float32)
    return raw, dirs, li, dir_pdf, on_n, normal


def test_mixture_loss_gradients_are_the_gradient_of_the_likelihood(oracle):
    # the analytic chain (distribution.h:348-421 x train.h:518-540) = d/d(raw) of the likelihood term -Li/q log p(raw)
    # the same function reports (train.h:520)
    rng = np.random.default_rng(0)
    n = 8
    raw, dirs, li, dir_pdf, on_n, normal = _random_training_batch3(rng, n)
    g, _ = oracle.vmm3_loss_gradients(raw, dirs, li, dir_pdf, on_n, normal, loss_scale=float(n))
    eps = 1e-3
    for j in range(40):
        hi, lo = raw.copy(), raw.copy()
        hi[:, j] += eps
        lo[:, j] -= eps
        _, lh = oracle.vmm3_loss_gradients(hi, dirs, li, dir_pdf, on_n, normal, float(n))
        _, ll = oracle.vmm3_loss_gradients(lo, dirs, li, dir_pdf, on_n, normal, float(n))
        num = (lh - ll) / (2 * eps)
        assert np.allclose(num, g[:, j], rtol=3e-2, atol=3e-3), j


def test_the_fitted_parabola_below_kappa_one(oracle):
    """1/kappa - coth(kappa) for kappa < 1 is the reference's parabola 0.000962 - 0.344883 kappa + 0.030147 kappa^2
    (distribution.h:389-392), not the closed form: the kappa gradient of a single-lobe mixture shows which one is used"""
    raw = np.zeros((1, 41), np.float32)
    raw[0, 0:40:5] = -10.0
    raw[0, 0] = 10.0
    kappa = 0.5
    raw[0, 1] = math.log(kappa)
    raw[0, 2:5] = [0.0, 0.0, 1.0]
    w = np.array([[0.6, 0.0, 0.8]], np.float32)
    g, _ = oracle.vmm3_loss_gradients(raw, w, [1.0], [1.0 - 1e-5], [0], np.zeros((1, 3), np.float32), loss_scale=1.0)
    p = float(oracle.vmf_eval([kappa], [0.8])[0])
    fitted = 0.000962 - 0.344883 * kappa + 0.030147 * kappa * kappa
    assert g[0, 1] == pytest.approx(-1.0 / (p + 1e-5) * p * (0.8 + fitted) * kappa, rel=2e-4)
    assert abs(fitted - (1.0 / kappa - 1.0 / math.tanh(kappa))) < 3e-3


@pytest.mark.gpu
def test_hip_mixture3_matches_oracle(oracle):
    from elaina_amd import integrator3d
    rng = np.random.default_rng(5)
    n = 30000
    raw, wi = _random_vmm3(rng, n)
    raw[:100, 2::5] = 0.0
    raw[:100, 3::5] = 0.0
    raw[:100, 4::5] = 0.0
    seed = rng.integers(0, 2**62, n).astype(np.uint64)
    gp, gd = integrator3d.vmm3_pdf_sample(raw, wi, seed)
    rp, rd = oracle.vmm3_pdf_sample(raw, wi, seed)
    assert np.array_equal(gp, rp) and np.array_equal(gd, rd, equal_nan=True) and np.isfinite(rd[100:]).all()
    batch = _random_training_batch3(rng, 20000)
    batch[0][:, 1:40:5] = rng.uniform(-3, 4, size=(20000, 8))          # both branches of the kappa derivative
    gg, gl = integrator3d.vmm3_loss_gradients(*batch)
    rg, rl = oracle.vmm3_loss_gradients(*batch)
    assert np.array_equal(gg, rg) and np.array_equal(gl, rl)
