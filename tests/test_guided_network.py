"""Guiding network (SURVEY.md 8a rows a22/a23 + the optimizer of a27).

CPU part: the oracle (oracle/wost_net.c) against the published tiny-cuda-nn layout numbers of the
reference configuration (data/ladybug/n.json:49-81), against finite differences and against the
closed-form first Adam/EMA step.  GPU part: the HIP network (elaina_amd/csrc/wost_net.hip) through
the C-ABI against the oracle.  PARITY UNPINNED w.r.t. tiny-cuda-nn itself (submodule absent)."""
import numpy as np
import pytest

from oracle.oracle import Oracle, default_net_config


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def _rand_params(orc, cfg, seed=5, wscale=0.25, gscale=0.5):
    n = orc.net_n_params(cfg)
    rng = np.random.default_rng(seed)
    p = rng.uniform(-wscale, wscale, n).astype(np.float32)
    n_mlp = 64 * 32 + 2 * 64 * 64 + 48 * 64
    p[n_mlp:] = rng.uniform(-gscale, gscale, n - n_mlp).astype(np.float32)
    return p


def test_layout_matches_reference_configuration(orc):
    cfg = default_net_config()
    # ~13.3k MLP + ~61k grid parameters
    n_mlp = 64 * 32 + 2 * 64 * 64 + 48 * 64
    assert n_mlp == 13312
    assert orc.net_n_params(cfg) == 74848
    res, scale, enc = orc.net_levels(cfg)
    assert enc == 32
    assert list(res) == [8, 12, 16, 23, 32, 44, 62, 87]
    # grid_scale = base * s^level - 1
    np.testing.assert_allclose(scale, 8.0 * 1.405 ** np.arange(8) - 1.0, rtol=1e-5)


def test_forward_is_bilinear_in_grid_and_relu_mlp(orc):
    cfg = default_net_config()
    p = _rand_params(orc, cfg)
    rng = np.random.default_rng(1)
    xy = rng.uniform(0, 1, (64, 2)).astype(np.float32)
    out, acts = orc.net_forward(cfg, p, xy, want_acts=True)
    assert out.shape == (64, 48) and acts.shape == (64, 32 + 3 * 64)
    assert np.all(acts[:, 32:] >= 0)
    # numpy restatement of the MLP on the oracle's own encoding
    W1 = p[:2048].reshape(64, 32); W2 = p[2048:6144].reshape(64, 64); W3 = p[6144:10240].reshape(64, 64)
    Wo = p[10240:13312].reshape(48, 64)
    h = acts[:, :32].astype(np.float64)
    for W in (W1, W2, W3):
        h = np.maximum(h @ W.astype(np.float64).T, 0)
    np.testing.assert_allclose(out, h @ Wo.astype(np.float64).T, rtol=2e-4, atol=2e-5)
    # the encoding at a grid vertex of level 0 equals that vertex's features (pos = x*scale + 0.5)
    res, scale, _ = orc.net_levels(cfg)
    gx, gy = 3, 5
    x = np.array([[(gx - 0.5) / scale[0], (gy - 0.5) / scale[0]]], dtype=np.float32)
    _, a = orc.net_forward(cfg, p, x, want_acts=True)
    grid0 = p[13312:13312 + 64 * 4].reshape(64, 4)
    np.testing.assert_allclose(a[0, :4], grid0[gx + gy * res[0]], rtol=0, atol=2e-5)


def test_backward_matches_finite_differences(orc):
    cfg = default_net_config()
    p = _rand_params(orc, cfg)
    rng = np.random.default_rng(2)
    xy = rng.uniform(0, 1, (16, 2)).astype(np.float32)
    dl = rng.normal(size=(16, 48)).astype(np.float32)
    dl[:, 33:] = 0
    grad = orc.net_backward(cfg, p, xy, dl)

    def loss(q):
        return float(np.sum(orc.net_forward(cfg, q, xy)[0].astype(np.float64) * dl))

    idx = np.concatenate([rng.integers(0, 13312, 12), 13312 + np.flatnonzero(grad[13312:])[::97][:12]])
    bad = []
    for i in idx:
        h = 2e-3
        q = p.copy(); q[i] += h; up = loss(q)
        q[i] -= 2 * h; dn = loss(q)
        fd = (up - dn) / (2 * h)
        if abs(fd - grad[i]) > 2e-2 * max(abs(fd), abs(grad[i])) + 2e-3:
            bad.append((int(i), fd, float(grad[i])))
    # a central difference may straddle a ReLU kink of one of the 16 x 192 units: allow two
    assert len(bad) <= 2, bad


def test_optimizer_first_step_closed_form(orc):
    cfg = default_net_config()
    n = orc.net_n_params(cfg)
    n_mlp = 64 * 32 + 2 * 64 * 64 + 48 * 64
    rng = np.random.default_rng(3)
    p0 = rng.normal(size=n).astype(np.float32)
    g = rng.normal(size=n).astype(np.float32) * 128
    st = orc.net_optimizer_state(cfg)
    p = p0.copy()
    inf = orc.net_optimizer_step(cfg, p, st, g, step=1, loss_scale=128.0)
    # step 1 of Adam with bias correction moves every weight by ~lr against the gradient sign;
    # the L2 term belongs to the matrix weights only (tiny-cuda-nn adam_step) ...
    gg = g / 128.0
    gg[:n_mlp] += cfg.l2_reg * p0[:n_mlp]
    np.testing.assert_allclose(p - p0, -cfg.learning_rate * np.sign(gg), rtol=1e-3, atol=1e-7)
    # ... and the debiased EMA of one sample is that sample
    np.testing.assert_allclose(inf, p, rtol=1e-5, atol=1e-7)
    p2 = p.copy()
    inf2 = orc.net_optimizer_step(cfg, p2, st, g, step=2, loss_scale=128.0)
    d = cfg.ema_decay
    np.testing.assert_allclose(inf2, (d * (1 - d) * p + (1 - d) * p2) / (1 - d * d), rtol=1e-4, atol=1e-6)


def test_optimizer_leaves_untouched_grid_entries_alone(orc):
    """tiny-cuda-nn's adam_step returns early for an encoding parameter whose gradient is exactly
    zero (no moment decay, no step, no L2 pull) and debiases every parameter with its own step
    counter: a grid entry first touched at global step 3 takes a FIRST Adam step there."""
    cfg = default_net_config()
    n = orc.net_n_params(cfg)
    n_mlp = 64 * 32 + 2 * 64 * 64 + 48 * 64
    rng = np.random.default_rng(5)
    p0 = rng.normal(size=n).astype(np.float32)
    g = rng.normal(size=n).astype(np.float32) * 128
    late = np.zeros(n, bool)
    late[n_mlp + 5::7] = True                 # grid entries that see no gradient in steps 1 and 2
    g_early = g.copy()
    g_early[late] = 0.0
    st = orc.net_optimizer_state(cfg)
    p = p0.copy()
    orc.net_optimizer_step(cfg, p, st, g_early, step=1, loss_scale=128.0)
    orc.net_optimizer_step(cfg, p, st, g_early, step=2, loss_scale=128.0)
    assert np.array_equal(p[late], p0[late])                      # not even the L2 term moved them
    assert not st["m1"][late].any() and not st["m2"][late].any() and not st["steps"][late].any()
    assert (st["steps"][~late] == 2).all()
    before = p.copy()
    inf = orc.net_optimizer_step(cfg, p, st, g, step=3, loss_scale=128.0)
    # their own first step: |dw| = lr exactly as at global step 1, no L2 on the encoding
    np.testing.assert_allclose((p - before)[late], -cfg.learning_rate * np.sign(g[late]), rtol=1e-3, atol=1e-7)
    assert (st["steps"][late] == 1).all() and (st["steps"][~late] == 3).all()
    # a matrix weight with zero gradient still decays its moments and feels the L2 term
    g0 = g.copy()
    g0[:n_mlp] = 0.0
    m1_before = st["m1"][:n_mlp].copy()
    orc.net_optimizer_step(cfg, p, st, g0, step=4, loss_scale=128.0)
    assert (st["steps"][:n_mlp] == 4).all() and not np.array_equal(st["m1"][:n_mlp], m1_before)
    assert np.isfinite(inf).all()


# ---- HIP network against the oracle ----------------------------------------------------------
@pytest.fixture(scope="module")
def net():
    from elaina_amd.guided import GuidingNetwork
    n = GuidingNetwork(seed=7)
    yield n
    n.close()


@pytest.mark.gpu
def test_gpu_initialisation_and_shapes(net, orc):
    cfg = default_net_config()
    assert net.n_params == orc.net_n_params(cfg) and net.n_mlp_params == 13312
    p = net.params()
    np.testing.assert_array_equal(p, net.inference_params())
    lim = np.sqrt(6.0 / (32 + 64))
    assert np.abs(p[:2048]).max() <= lim * 1.0001 and np.abs(p[:2048]).max() > 0.9 * lim
    assert np.abs(p[13312:]).max() <= 1.0001e-4 and abs(float(p.mean())) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 64, 1000, 65536 + 17])
def test_gpu_inference_matches_oracle(net, orc, n):
    cfg = default_net_config()
    p = _rand_params(orc, cfg, seed=11)
    net.set_params(p)
    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    xy[0] = (0.0, 1.0)
    want = orc.net_forward(cfg, p, xy)[0][:, :33]
    got = net.inference(xy)
    # same fma chains in the same order on both sides: expected bit-exact, gate at 1e-6
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7)
    assert np.array_equal(got, want), "fp32 forward drifted from the oracle's operation order"


@pytest.mark.gpu
@pytest.mark.parametrize("n", [5, 64, 4096 + 3])
def test_gpu_gradients_match_oracle(net, orc, n):
    cfg = default_net_config()
    p = _rand_params(orc, cfg, seed=13)
    net.set_params(p)
    rng = np.random.default_rng(100 + n)
    xy = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    dl = rng.normal(size=(n, 33)).astype(np.float32)
    net.train_step(xy, dl, apply_update=False)
    got = net.gradients()
    dl48 = np.zeros((n, 48), dtype=np.float32)
    dl48[:, :33] = dl
    want = orc.net_backward(cfg, p, xy, dl48)
    # sums in 64-bit fixed point with integer atomics: order-independent, hence bit-exact
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    np.testing.assert_array_equal(net.params(), p)      # apply_update=False leaves the weights alone


@pytest.mark.gpu
def test_gpu_training_steps_match_oracle(net, orc):
    """forward, backward, fixed-point gradient sums, Adam and the debiased EMA: bit-exact over
    several steps, so a whole training run is reproducible"""
    cfg = default_net_config()
    p = _rand_params(orc, cfg, seed=17, gscale=0.1)
    net.set_params(p)
    st = orc.net_optimizer_state(cfg)
    rng = np.random.default_rng(4)
    po = p.copy()
    for step in range(1, 5):
        n = 3000 + 37 * step                       # not a multiple of the 1024-point chunks
        xy = rng.uniform(0, 1, (n, 2)).astype(np.float32)
        dl = (rng.normal(size=(n, 33)) * 128 / n).astype(np.float32)
        net.train_step(xy, dl, loss_scale=128.0)
        dl48 = np.zeros((n, 48), dtype=np.float32)
        dl48[:, :33] = dl
        g = orc.net_backward(cfg, po, xy, dl48)
        assert np.array_equal(net.gradients(), g)
        inf = orc.net_optimizer_step(cfg, po, st, g, step=step, loss_scale=128.0)
        assert np.array_equal(net.params(), po), step
        assert np.array_equal(net.inference_params(), inf), step


@pytest.mark.gpu
def test_gpu_unfused_backward_kernels_give_the_same_gradient(net, orc, monkeypatch):
    """WOST_NET_FUSED=0: backward pass and weight gradients as separate kernels (deltas through
    global memory); the default is the fused kernel.  Same sums in the same order."""
    from elaina_amd.guided import GuidingNetwork
    cfg = default_net_config()
    p = _rand_params(orc, cfg, seed=23, gscale=0.1)
    rng = np.random.default_rng(12)
    n = 5000 + 13
    xy = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    dl = (rng.normal(size=(n, 33)) * 128 / n).astype(np.float32)
    net.set_params(p)
    net.train_step(xy, dl, apply_update=False)
    fused = net.gradients().copy()
    monkeypatch.setenv("WOST_NET_FUSED", "0")
    other = GuidingNetwork(seed=7)
    try:
        other.set_params(p)
        other.train_step(xy, dl, apply_update=False)
        assert np.array_equal(other.gradients(), fused)
    finally:
        other.close()
    dl48 = np.zeros((n, 48), dtype=np.float32)
    dl48[:, :33] = dl
    assert np.array_equal(fused, orc.net_backward(cfg, p, xy, dl48))


@pytest.mark.gpu
def test_gpu_scalar_kernels_match_oracle(orc, monkeypatch):
    """The one-thread-per-point kernels (other network shapes, WOST_NET_SCALAR=1) follow the same
    operation order as the MFMA kernels: inference and a training step, bit for bit."""
    from elaina_amd.guided import GuidingNetwork
    monkeypatch.setenv("WOST_NET_SCALAR", "1")
    net = GuidingNetwork(seed=7)
    try:
        cfg = default_net_config()
        p = _rand_params(orc, cfg, seed=19, gscale=0.1)
        net.set_params(p)
        rng = np.random.default_rng(9)
        n = 2048 + 77
        xy = rng.uniform(0, 1, (n, 2)).astype(np.float32)
        assert np.array_equal(net.inference(xy), orc.net_forward(cfg, p, xy)[0][:, :33])
        dl = (rng.normal(size=(n, 33)) * 128 / n).astype(np.float32)
        net.train_step(xy, dl, loss_scale=128.0)
        dl48 = np.zeros((n, 48), dtype=np.float32)
        dl48[:, :33] = dl
        g = orc.net_backward(cfg, p, xy, dl48)
        assert np.array_equal(net.gradients(), g)
        st = orc.net_optimizer_state(cfg)
        po = p.copy()
        inf = orc.net_optimizer_step(cfg, po, st, g, step=1, loss_scale=128.0)
        assert np.array_equal(net.params(), po) and np.array_equal(net.inference_params(), inf)
    finally:
        net.close()


@pytest.mark.gpu
def test_gpu_network_learns_a_field(net):
    """End-to-end sanity: L2 regression of a smooth 33-channel field drives the loss down and the
    EMA weights follow (what the guided integrator relies on between training iterations)."""
    from elaina_amd.guided import GuidingNetwork
    n = GuidingNetwork(seed=3)
    rng = np.random.default_rng(8)
    freq = rng.uniform(1, 3, (33, 2))

    def target(xy):
        return np.sin(xy @ freq.T * 2 * np.pi).astype(np.float32)

    test_xy = rng.uniform(0, 1, (2048, 2)).astype(np.float32)
    first = float(np.mean((n.inference(test_xy) - target(test_xy)) ** 2))
    for _ in range(300):
        xy = rng.uniform(0, 1, (4096, 2)).astype(np.float32)
        pred = n.inference(xy, use_inference_params=False)
        n.train_step(xy, 2.0 * (pred - target(xy)) / pred.size * 128.0, loss_scale=128.0)
    last = float(np.mean((n.inference(test_xy) - target(test_xy)) ** 2))
    n.close()
    assert first > 0.3 and last < 0.1 * first, (first, last)


# ---- half-precision inference (the reference's network precision) -----------------------------
def _half_network_numpy(orc, cfg, p, xy):
    """The half-precision network written out in numpy: grid values rounded to f16, the oracle's fp32
    bilinear interpolation, then per layer f16 inputs and weights, fp32 accumulation, ReLU, f16."""
    n_mlp = 64 * 32 + 2 * 64 * 64 + 48 * 64
    ph = p.copy()
    ph[n_mlp:] = p[n_mlp:].astype(np.float16).astype(np.float32)
    enc = orc.net_forward(cfg, ph, xy, want_acts=True)[1][:, :32]
    a = enc.astype(np.float16).astype(np.float32)
    off = 0
    for no, ni, relu in [(64, 32, True), (64, 64, True), (64, 64, True), (48, 64, False)]:
        w = p[off:off + no * ni].reshape(no, ni).astype(np.float16).astype(np.float32)
        off += no * ni
        z = (a.astype(np.float64) @ w.T.astype(np.float64)).astype(np.float32)       # products of f16 numbers are exact in fp32
        a = (np.maximum(z, 0.0) if relu else z).astype(np.float16).astype(np.float32)
    return a[:, :33]


def _half_training_numpy(orc, cfg, p, xy, dl):
    """Forward and backward pass of one half-precision training step in numpy (float64 sums of f16
    numbers): -> (raw outputs [n, 33], gradient of the MLP matrices, dL/d(encoding) [n, 32]).  The deltas
    carry the kernel's extra power-of-two scale (~ n / 512, at most 1024) while they are f16."""
    k = 0
    while k < 10 and (len(xy) >> (k + 10)) > 0:
        k += 1
    dscale = float(1 << k)
    n_mlp = 64 * 32 + 2 * 64 * 64 + 48 * 64
    f16 = lambda v: np.asarray(v).astype(np.float16).astype(np.float64)
    ph = p.copy()
    ph[n_mlp:] = p[n_mlp:].astype(np.float16).astype(np.float32)
    acts = [f16(orc.net_forward(cfg, ph, xy, want_acts=True)[1][:, :32])]
    shapes = [(64, 32), (64, 64), (64, 64), (48, 64)]
    ws, off = [], 0
    for no, ni in shapes:
        ws.append(f16(p[off:off + no * ni].reshape(no, ni)))
        off += no * ni
    for l in range(3):
        acts.append(f16(np.maximum((acts[l] @ ws[l].T).astype(np.float32), 0.0)))
    out = f16((acts[3] @ ws[3].T).astype(np.float32))[:, :33]
    d = np.zeros((len(xy), 48))
    d[:, :33] = f16(dl * np.float32(dscale))
    grads = [None] * 4
    for l in (3, 2, 1, 0):
        grads[l] = (d.T @ acts[l]) / dscale
        back = (d @ ws[l]).astype(np.float32)
        if l == 0:
            denc = back / np.float32(dscale)
        else:
            d = f16(np.where(acts[l] > 0, back, 0.0))
    return out.astype(np.float32), np.concatenate([g.ravel() for g in grads]), denc


@pytest.mark.gpu
@pytest.mark.parametrize("n", [7, 64, 8192 + 5])
def test_gpu_half_precision_inference(orc, n):
    """precision 16 = the arithmetic the reference's tiny-cuda-nn network runs in (half weights, activations
    and grid, fp32 accumulation).  Gate 1: it IS that network -- equal to the numpy emulation up to the
    summation order inside the matrix instruction (a few f16 ulps).  Gate 2: it stays close to the fp32
    network (relative error of the raw outputs around 1e-3, the half-precision mantissa)."""
    from elaina_amd.guided import GuidingNetwork
    cfg = default_net_config()
    net = GuidingNetwork(seed=7)
    try:
        p = _rand_params(orc, cfg, seed=31, wscale=0.2, gscale=0.4)
        net.set_params(p)
        rng = np.random.default_rng(n)
        xy = rng.uniform(0, 1, (n, 2)).astype(np.float32)
        fp32 = net.inference(xy)
        assert np.array_equal(fp32, orc.net_forward(cfg, p, xy)[0][:, :33])
        net.set_option("precision", 16)
        half = net.inference(xy)
        emu = _half_network_numpy(orc, cfg, p, xy)
        scale = float(np.sqrt(np.mean(fp32 ** 2)))
        assert np.abs(half - emu).max() <= 4e-3 * scale, (float(np.abs(half - emu).max()), scale)
        assert np.mean(half == emu) > 0.9                                  # most outputs agree to the last f16 bit
        rel = float(np.sqrt(np.mean((half - fp32) ** 2))) / scale
        assert rel < 5e-3, rel
        assert np.array_equal(half, half.astype(np.float16).astype(np.float32))     # outputs are half-precision numbers
        # training parameters are never evaluated in half precision; fp32 mode comes back bit for bit
        assert np.array_equal(net.inference(xy, use_inference_params=False), fp32)
        net.set_option("precision", 32)
        assert np.array_equal(net.inference(xy), fp32)
    finally:
        net.close()


@pytest.mark.gpu
def test_gpu_half_precision_follows_the_optimizer(orc):
    """the f16 fragments are refreshed after every Adam / EMA step: inference in half precision tracks the
    fp32 inference weights through training"""
    from elaina_amd.guided import GuidingNetwork
    cfg = default_net_config()
    net = GuidingNetwork(seed=5)
    try:
        net.set_option("precision", 16)
        rng = np.random.default_rng(2)
        xy = rng.uniform(0, 1, (4096, 2)).astype(np.float32)
        test = rng.uniform(0, 1, (512, 2)).astype(np.float32)
        for _ in range(20):
            pred = net.inference(xy, use_inference_params=False)
            net.train_step(xy, (2.0 * (pred - 0.5) / pred.size * 128.0).astype(np.float32), loss_scale=128.0)
        half = net.inference(test)
        emu = _half_network_numpy(orc, cfg, net.inference_params(), test)
        scale = float(np.sqrt(np.mean(emu ** 2))) + 1e-6
        assert np.abs(half - emu).max() <= 4e-3 * scale
    finally:
        net.close()


# ---- half-precision training passes ("train_precision" 16) ---------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n", [5, 64, 4096 + 3, 70000])
def test_gpu_half_precision_training_passes(orc, n):
    """train_precision 16: forward, backward and weight gradients on f16 matrix instructions with fp32
    accumulation.  Gate 1: equal to the numpy restatement of exactly that arithmetic up to summation
    order.  Gate 2: close to the fp32 gradient (the bit-exact mode).  Gate 3: reproducible bit for bit."""
    from elaina_amd.guided import GuidingNetwork
    cfg = default_net_config()
    n_mlp = 64 * 32 + 2 * 64 * 64 + 48 * 64
    net = GuidingNetwork(seed=7)
    try:
        p = _rand_params(orc, cfg, seed=41, wscale=0.2, gscale=0.4)
        net.set_params(p)
        rng = np.random.default_rng(500 + n)
        xy = rng.uniform(0, 1, (n, 2)).astype(np.float32)
        dl = (rng.normal(size=(n, 33)) * 128 / n).astype(np.float32)
        net.train_step(xy, dl, apply_update=False)
        g32 = net.gradients()
        net.set_option("train_precision", 16)
        emu_out, emu_g, _ = _half_training_numpy(orc, cfg, p, xy, dl)
        out = net.inference(xy, use_inference_params=False)
        scale = float(np.sqrt(np.mean(emu_out ** 2)))
        assert np.abs(out - emu_out).max() <= 4e-3 * scale
        assert np.mean(out == emu_out) > 0.9
        net.train_step(xy, dl, apply_update=False)
        g16 = net.gradients()
        np.testing.assert_array_equal(net.params(), p)
        # matrices: against the restatement (f16 rounding of a delta can flip with the summation order -> a few 1e-3)
        gs = float(np.sqrt(np.mean(emu_g ** 2)))
        assert np.sqrt(np.mean((g16[:n_mlp] - emu_g) ** 2)) < 3e-3 * gs, (float(np.sqrt(np.mean((g16[:n_mlp] - emu_g) ** 2))), gs)
        # everything: against the fp32 gradient.  dL/dout is noise here, so the gradient is a random-walk sum and
        # the ~1e-3 of the hidden units whose ReLU flips between the two precisions show up as sqrt(1e-3) ~ 3 %
        # (the restatement above differs from fp32 by the same amount); a structured loss is far closer (the
        # learning test below)
        for sl in (slice(0, n_mlp), slice(n_mlp, None)):
            ref = float(np.sqrt(np.mean(g32[sl] ** 2)))
            err = float(np.sqrt(np.mean((g16[sl] - g32[sl]) ** 2)))
            assert err < 6e-2 * ref, (sl, err, ref)
        # grid entries no training point touched keep a zero gradient (the optimizer skips them)
        assert np.array_equal(g16[n_mlp:] == 0, g32[n_mlp:] == 0) or n < 100
        net.train_step(xy, dl, apply_update=False)
        assert np.array_equal(net.gradients(), g16)
        net.set_option("train_precision", 32)
        net.train_step(xy, dl, apply_update=False)
        assert np.array_equal(net.gradients(), g32)
    finally:
        net.close()


@pytest.mark.gpu
def test_gpu_half_precision_training_learns_a_field():
    """the regression of test_gpu_network_learns_a_field with both passes in half precision: the loss
    falls the same way (fp32 master weights, Adam and EMA as before)"""
    from elaina_amd.guided import GuidingNetwork
    rng0 = np.random.default_rng(8)
    freq = rng0.uniform(1, 3, (33, 2))

    def target(xy):
        return np.sin(xy @ freq.T * 2 * np.pi).astype(np.float32)

    test_xy = rng0.uniform(0, 1, (2048, 2)).astype(np.float32)
    last = {}
    for prec in (32, 16):
        n = GuidingNetwork(seed=3)
        try:
            n.set_option("train_precision", prec)
            n.set_option("precision", prec)
            rng = np.random.default_rng(9)
            first = float(np.mean((n.inference(test_xy) - target(test_xy)) ** 2))
            for _ in range(300):
                xy = rng.uniform(0, 1, (4096, 2)).astype(np.float32)
                pred = n.inference(xy, use_inference_params=False)
                n.train_step(xy, 2.0 * (pred - target(xy)) / pred.size * 128.0, loss_scale=128.0)
            last[prec] = float(np.mean((n.inference(test_xy) - target(test_xy)) ** 2))
            assert first > 0.3 and last[prec] < 0.1 * first, (prec, first, last[prec])
        finally:
            n.close()
    assert abs(last[16] - last[32]) < 0.25 * last[32], last
