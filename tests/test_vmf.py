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
