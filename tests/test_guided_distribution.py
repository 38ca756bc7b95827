"""Guided path, distribution layer: the oracle against the reference's
own known-answer constants (test/vonmises_test.cu, commented out there but numerically valid),
and the HIP entry points against the oracle within the reference tests' 1e-5 tolerance."""
import math

import numpy as np
import pytest

REL = 1e-5  # Catch::Matchers::WithinRel(..., 1e-5f) in the reference tests


def test_eval_poly_kat(oracle):
    # test/vonmises_test.cu:5-9
    assert oracle.eval_poly_large0(1.14514) == pytest.approx(0.4184690292340133, rel=REL)


def test_log_bessel_kat(oracle):
    # test/vonmises_test.cu:11-22
    r = oracle.vonmises_eval([1.0, 2.0, 3.0, 4.0], [0, 0, 0, 0])
    assert np.allclose(r["log_i0"], [0.23591432, 0.82399356, 1.58530772, 2.42497277], rtol=REL)


def test_von_mises_log_pdf_kat(oracle):
    # test/vonmises_test.cu:57-59 (kappa 4.2; today's API takes cos(theta): util/vonmises.h:128-133)
    th = np.array([-2.0, -1.0, 0.0, 1.0, 2.0])
    r = oracle.vonmises_eval([4.2] * 5, np.cos(th))
    assert np.allclose(r["log_pdf"], [-6.18411160, -2.16702533, -0.23629522, -2.16702533, -6.18411160], rtol=REL)
    assert np.allclose(np.exp(r["log_pdf"]), [0.00206193, 0.11451776, 0.78954756, 0.11451776, 0.00206193], rtol=1e-4)


def test_von_mises_dlog_dkappa_kat(oracle):
    # test/vonmises_test.cu:124-148: small-kappa and large-kappa branches
    r = oracle.vonmises_eval([1.45, 14.5], np.cos([0.5, 0.5]))
    assert r["dlog_dkappa"][0] == pytest.approx(0.29405486583709717, rel=REL)
    assert r["dlog_dkappa"][1] == pytest.approx(-0.08729398250579834, rel=REL)


def test_von_mises_kernel_kats_of_the_distribution_test(oracle):
    """test/distribution_test.cu:49-56,112-128 (commented out there, written for an older kernel class, but the numbers
    do not depend on its parametrisation): the von Mises density with kappa 1.45 and mean pi/4 at angle 0 -- also with
    the mean given as pi/4 + 2 pi -- and its derivatives by kappa and by the mean, within that file's 1e-5"""
    for mean in (math.pi / 4, math.pi / 4 + 2 * math.pi):
        r = oracle.vonmises_eval([1.45], [math.cos(0.0 - mean)])
        pdf = math.exp(r["log_pdf"][0])
        assert pdf == pytest.approx(0.27751895785331726, abs=1e-5)
        assert pdf * r["dlog_dkappa"][0] == pytest.approx(0.034295544028282166, abs=1e-5)
        # d/d(mean) exp(kappa cos(theta - mean)) = kappa sin(theta - mean) pdf
        assert pdf * 1.45 * math.sin(0.0 - mean) == pytest.approx(-0.284541517496109, abs=1e-5)


def _mixture_kat_raw():
    """the three-component mixture of test/distribution_test.cu:178-185 in today's parametrisation.  That (older) class
    read (lambda, kappa, mean) per component as exp(x), exp(x), 2 pi sigmoid(x) -- the only reading of its nine numbers
    that reproduces its own expected density, to 4e-9 -- and today's takes exp(x), exp(x) and a mean VECTOR; the
    weights lambda / sum(lambda) do not change when all lambdas are scaled, so the three are raised by 13 and the five
    components the old class did not have sit at the lower clamp: 1e-10 of the weight"""
    d = np.array([-0.3391095697879791, 1.3653955459594727, -0.11165934801101685, 0.7329881191253662, 1.1205719709396362,
                  -1.145609736442566, 1.5198860168457031, -0.962236225605011, 1.4103161096572876]).reshape(3, 3)
    raw = np.zeros((1, 33), np.float32)
    raw[0, 0:32:4] = -10.0
    for i in range(3):
        mean = 2.0 * math.pi / (1.0 + math.exp(-d[i, 2]))
        raw[0, 4 * i:4 * i + 4] = [d[i, 0] + 13.0, d[i, 1], math.cos(mean), math.sin(mean)]
    return raw


def test_mixture_density_kat_of_the_distribution_test(oracle):
    # test/distribution_test.cu:186-188: density of the mixture at angle 0 (the value gradients_probability returns)
    pdf, _ = oracle.vmm_pdf_sample(_mixture_kat_raw(), np.array([[1.0, 0.0]], np.float32), np.array([1], np.uint64))
    assert pdf[0] == pytest.approx(0.11850630, abs=1e-5)


_MIXTURE_KAT_GRADIENTS = [-0.016046222299337387, -5.7009561714949086e-05, -2.110011519107502e-05,
                          -0.011129779741168022, -0.007846416905522346, -0.031608663499355316,
                          0.00756735447794199, 0.015586040914058685, 0.0389787033200264]


def _mixture_kat_gradients(loss_gradients):
    """d(density) / d(lambda, kappa, mean angle) of the three components at angle 0, recovered from the gradient of the
    training loss: with one sample, Li = 1, dirPdf + eps = 1 and loss scale 1 the loss gradient of a raw output is
    -1 / (density + eps) times the parameter gradient times the derivative of the activation
    (integrator/guided/train.h:81-105,492-553); the mean angle moves the unit mean vector along its tangent"""
    raw = _mixture_kat_raw()
    g, _ = loss_gradients(raw, np.array([[1.0, 0.0]], np.float32), np.array([1.0], np.float32), np.array([1.0 - 1e-5], np.float32),
                          np.array([0], np.uint8), np.zeros((1, 2), np.float32), loss_scale=1.0)
    prefix = -1.0 / (0.11850630 + 1e-5)
    out = []
    for i in range(3):
        lam, kap = math.exp(float(raw[0, 4 * i]) - 13.0), math.exp(float(raw[0, 4 * i + 1]))
        mx, my = float(raw[0, 4 * i + 2]), float(raw[0, 4 * i + 3])
        out += [g[0, 4 * i] / (prefix * lam), g[0, 4 * i + 1] / (prefix * kap), (g[0, 4 * i + 2] * -my + g[0, 4 * i + 3] * mx) / prefix]
    return out


def test_mixture_gradient_kat_of_the_distribution_test(oracle):
    # test/distribution_test.cu:186-196: the nine expected gradients of the density, within that file's 1e-5
    assert np.allclose(_mixture_kat_gradients(oracle.vmm_loss_gradients), _MIXTURE_KAT_GRADIENTS, rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_hip_mixture_gradient_kat_of_the_distribution_test():
    from elaina_amd import guided
    assert np.allclose(_mixture_kat_gradients(guided.vmm_loss_gradients), _MIXTURE_KAT_GRADIENTS, rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_hip_mixture_density_kat_of_the_distribution_test():
    from elaina_amd import guided
    pdf, _ = guided.vmm_pdf_sample(_mixture_kat_raw(), np.array([[1.0, 0.0]], np.float32), np.array([1], np.uint64))
    assert pdf[0] == pytest.approx(0.11850630, abs=1e-5)


@pytest.mark.parametrize("kappa,n,eps", [(1.45, 200000, 0.02), (145.0, 20000, 0.05), (1e-4, 50000, None)])
def test_von_mises_sampling_moments(oracle, kappa, n, eps):
    # test/vonmises_test.cu:72-122: circular mean ~ 0 and circular variance 1 - I1/I0, seed 42
    th = oracle.vonmises_sample([kappa], [42], n)[0]
    if eps is None:   # kappa < 1e-3: uniform on [0, 2pi)
        assert th.min() >= 0 and th.max() < 2 * math.pi and abs(th.mean() - math.pi) < 0.05
        return
    assert abs(th.mean()) < 0.1
    R = math.hypot(np.cos(th).mean(), np.sin(th).mean())
    r = oracle.vonmises_eval([kappa], [0.0])
    theory = 1.0 - math.exp(float(r["log_i1"][0]) - float(r["log_i0"][0]))
    assert (1.0 - R) == pytest.approx(theory, rel=eps)


def _random_vmm(rng, n):
    raw = rng.normal(0, 1.5, size=(n, 32)).astype(np.float32)
    raw[:, 1::4] = rng.uniform(-3, 5, size=(n, 8))      # log kappa
    ang = rng.uniform(0, 2 * np.pi, size=n)
    wi = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)
    return raw, wi


def test_vmm_pdf_integrates_to_one(oracle):
    rng = np.random.default_rng(0)
    raw, _ = _random_vmm(rng, 4)
    ang = (np.arange(4096) + 0.5) * (2 * np.pi / 4096)
    wi = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)
    for k in range(4):
        pdf, _ = oracle.vmm_pdf_sample(np.repeat(raw[k:k + 1], 4096, 0), wi, np.zeros(4096, np.uint64))
        assert pdf.sum() * (2 * np.pi / 4096) == pytest.approx(1.0, abs=2e-3)


def test_vmm_with_a_zero_mean_vector_stays_finite(oracle):
    """mu_original.normalized() is Eigen's normalized(): a zero vector is returned unchanged, not divided by its
    zero norm (distribution.h:160).  Half-precision network outputs do underflow to exact zeros."""
    rng = np.random.default_rng(1)
    raw, wi = _random_vmm(rng, 64)
    raw[:, 2:32:4] = 0.0          # mu.x of every lobe
    raw[:, 3:32:4] = 0.0          # mu.y
    pdf, d = oracle.vmm_pdf_sample(raw, wi, np.arange(64, dtype=np.uint64))
    assert np.isfinite(pdf).all() and np.isfinite(d).all() and (pdf > 0).all()
    assert not d.any()            # a lobe without a direction samples the zero vector, as the reference's frame does


def test_vmm_samples_follow_the_pdf(oracle):
    rng = np.random.default_rng(1)
    raw, _ = _random_vmm(rng, 1)
    n = 60000
    _, d = oracle.vmm_pdf_sample(np.repeat(raw, n, 0), np.zeros((n, 2), np.float32), np.arange(n, dtype=np.uint64))
    assert np.allclose(np.hypot(d[:, 0], d[:, 1]), 1.0, atol=1e-5)
    bins = 16
    hist, _ = np.histogram(np.arctan2(d[:, 1], d[:, 0]), bins=bins, range=(-np.pi, np.pi))
    ang = (np.arange(4096) + 0.5) * (2 * np.pi / 4096) - np.pi
    wi = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)
    pdf, _ = oracle.vmm_pdf_sample(np.repeat(raw, 4096, 0), wi, np.zeros(4096, np.uint64))
    expect = pdf.reshape(bins, -1).sum(1) * (2 * np.pi / 4096) * n
    assert np.all(np.abs(hist - expect) < 5 * np.sqrt(expect + 1) + 0.01 * n / bins)


@pytest.mark.gpu
def test_hip_vonmises_eval_matches_oracle_and_kats(oracle):
    from elaina_amd import guided
    rng = np.random.default_rng(3)
    kappa = np.concatenate([[1.0, 2.0, 3.0, 4.0, 4.2, 1.45, 14.5], np.exp(rng.uniform(-9, 9, 20000))]).astype(np.float32)
    cos = np.concatenate([[0, 0, 0, 0, 1, math.cos(0.5), math.cos(0.5)], rng.uniform(-1, 1, 20000)]).astype(np.float32)
    got = guided.vonmises_eval(kappa, cos)
    ref = oracle.vonmises_eval(kappa, cos)
    assert np.allclose(got["log_i0"][:4], [0.23591432, 0.82399356, 1.58530772, 2.42497277], rtol=REL)
    assert got["dlog_dkappa"][5] == pytest.approx(0.29405486583709717, rel=REL)
    assert got["dlog_dkappa"][6] == pytest.approx(-0.08729398250579834, rel=REL)
    for k in ("log_i0", "log_i1", "log_pdf"):
        assert np.allclose(got[k], ref[k], rtol=REL, atol=2e-6 * np.maximum(1.0, np.abs(kappa))), k
    # the derivative cancels to ~1/kappa^2 at large kappa: compare on the scale of its terms
    assert np.allclose(got["dlog_dkappa"], ref["dlog_dkappa"], rtol=REL, atol=3e-6)


@pytest.mark.gpu
def test_hip_vonmises_sampling_matches_oracle(oracle):
    from elaina_amd import guided
    rng = np.random.default_rng(4)
    kappa = np.exp(rng.uniform(-8, 7, 4000)).astype(np.float32)
    seed = rng.integers(0, 2**62, 4000).astype(np.uint64)
    got = guided.vonmises_sample(kappa, seed, 8)
    ref = oracle.vonmises_sample(kappa, seed, 8)
    # identical PCG streams and double-precision acceptance tests: the same trials are accepted
    # except where libm/ocml differ in the last bits of a borderline comparison (vanishingly rare)
    close = np.isclose(got, ref, rtol=0, atol=2e-6)
    assert close.mean() > 0.9995, close.mean()


@pytest.mark.gpu
def test_hip_vmm_pdf_and_sample_match_oracle(oracle):
    from elaina_amd import guided
    rng = np.random.default_rng(5)
    raw, wi = _random_vmm(rng, 30000)
    seed = rng.integers(0, 2**62, 30000).astype(np.uint64)
    gp, gd = guided.vmm_pdf_sample(raw, wi, seed)
    rp, rd = oracle.vmm_pdf_sample(raw, wi, seed)
    assert np.allclose(gp, rp, rtol=1e-4, atol=1e-7)     # 1e-4 agreement of the VMM sub-kernels
    close = np.isclose(gd, rd, rtol=0, atol=1e-5).all(1)
    assert close.mean() > 0.999, close.mean()
    raw[:100, 2:32:4] = 0.0       # zero mean vectors (Eigen normalized(): left as they are): finite on both sides
    raw[:100, 3:32:4] = 0.0
    gp, gd = guided.vmm_pdf_sample(raw[:100], wi[:100], seed[:100])
    rp, rd = oracle.vmm_pdf_sample(raw[:100], wi[:100], seed[:100])
    assert np.isfinite(gp).all() and np.allclose(gp, rp, rtol=1e-4, atol=1e-7) and np.array_equal(gd, rd) and not gd.any()


def _random_training_batch(rng, n):
    raw = rng.normal(0, 1, size=(n, 33)).astype(np.float32)
    raw[:, 1:32:4] = rng.uniform(-2, 4, size=(n, 8))
    ang = rng.uniform(0, 2 * np.pi, n)
    dirs = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)
    li = rng.uniform(0.0, 1.0, n).astype(np.float32)
    dir_pdf = rng.uniform(0.05, 0.6, n).astype(np.float32)
    on_n = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    na = rng.uniform(0, 2 * np.pi, n)
    normal = np.stack([np.cos(na), np.sin(na)], 1).astype(np.float32)
    return raw, dirs, li, dir_pdf, on_n, normal


def test_vmm_loss_gradients_are_the_gradient_of_the_likelihood(oracle):
    # the analytic chain (distribution.h:201-264 x train.h:518-538) must equal d/draw of the
    # likelihood term -Li/q * log p(raw) that the same kernel reports (train.h:520)
    rng = np.random.default_rng(0)
    n = 8
    raw, dirs, li, dir_pdf, on_n, normal = _random_training_batch(rng, n)
    g, _ = oracle.vmm_loss_gradients(raw, dirs, li, dir_pdf, on_n, normal, loss_scale=float(n))
    eps = 1e-3
    for j in range(32):
        hi, lo = raw.copy(), raw.copy()
        hi[:, j] += eps
        lo[:, j] -= eps
        _, lh = oracle.vmm_loss_gradients(hi, dirs, li, dir_pdf, on_n, normal, float(n))
        _, ll = oracle.vmm_loss_gradients(lo, dirs, li, dir_pdf, on_n, normal, float(n))
        num = (lh - ll) / (2 * eps)
        assert np.allclose(num, g[:, j], rtol=3e-2, atol=3e-3), j


@pytest.mark.gpu
def test_hip_vmm_loss_gradients_match_oracle(oracle):
    from elaina_amd import guided
    rng = np.random.default_rng(6)
    batch = _random_training_batch(rng, 20000)
    gg, gl = guided.vmm_loss_gradients(*batch)
    rg, rl = oracle.vmm_loss_gradients(*batch)
    scale = np.abs(rg).max(axis=1, keepdims=True) + 1e-12
    assert np.all(np.abs(gg - rg) <= 2e-4 * scale + 1e-9)     # 1e-4-level agreement
    assert np.allclose(gl, rl, rtol=1e-4, atol=1e-6)
