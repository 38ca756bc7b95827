"""GuidedIntegrator<3> (SURVEY.md 8a rows a21-a27 with DIM == 3; reference exec.cu:102-122, guided/parameters.h:26-33).

CPU part: the three-input network of the oracle (oracle/wost_net.c wo_net3_*) against finite differences and a numpy
restatement of the trilinear encoding; the guided 3-D solve of the oracle (wost_oracle3d.c wo3_solve_guided) -- deterministic,
unbiased against an analytic harmonic solution.  GPU part: the HIP integrator (wost3_guided_*, wost3_net_create) through the
C-ABI against the oracle, bit for bit: network, frozen-network walks, first-pass records, a whole trained solve.
PARITY UNPINNED w.r.t. tiny-cuda-nn and snch-lbvh (submodules absent), like the 2-D path."""
import numpy as np
import pytest

from conftest import cube_scene3
from oracle.oracle import Oracle, default_net_config3, guided_settings3

AABB3 = ((-0.1, -0.1, -0.1), (1.1, 1.1, 1.1))
EPS = 1e-3


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def _cfg():
    # four levels keep the dense 3-D grid small (8^3 + 12^3 + 16^3 + 23^3 entries); the code path is the same for eight
    return default_net_config3(n_levels=4)


def _rand_params3(orc, cfg, seed=5, wscale=0.25, gscale=0.5):
    n = orc.net3_n_params(cfg)
    rng = np.random.default_rng(seed)
    p = rng.uniform(-wscale, wscale, n).astype(np.float32)
    n_mlp = 64 * (4 * cfg.n_levels) + 2 * 64 * 64 + 48 * 64
    p[n_mlp:] = rng.uniform(-gscale, gscale, n - n_mlp).astype(np.float32)
    return p


def mixed_cube():
    """Dirichlet u = z on the faces z = 0 and z = 1, zero flux on the four others: u = z inside"""
    return cube_scene3(n=3, d_faces=(4, 5), n_faces=(0, 1, 2, 3), value=lambda x, y, z: z, flux=lambda x, y, z, f: 0.0)


# ---- CPU: the three-input network ------------------------------------------------------------------------------------
def test_net3_layout_and_trilinear_encoding(orc):
    cfg = default_net_config3()
    # 8 levels: res = ceil(8 * 1.405^l - 1) + 1, res^3 entries rounded up to 8, 4 features each
    res = [8, 12, 16, 23, 32, 44, 62, 87]
    n_grid = sum((r ** 3 + 7) // 8 * 8 for r in res) * 4
    n_mlp = 64 * 32 + 2 * 64 * 64 + 48 * 64
    assert orc.net3_n_params(cfg) == n_mlp + n_grid
    cfg = _cfg()
    p = _rand_params3(orc, cfg)
    n_mlp = 64 * (4 * cfg.n_levels) + 2 * 64 * 64 + 48 * 64
    # level 0 of the encoding against a numpy restatement: scale 7, res 8, index x + 8 y + 64 z
    rng = np.random.default_rng(2)
    x = rng.uniform(0.02, 0.98, (32, 3)).astype(np.float32)
    W1 = p[:64 * 16].reshape(64, 16).astype(np.float64)
    grid0 = p[n_mlp:n_mlp + 512 * 4].reshape(512, 4).astype(np.float64)
    pos = x.astype(np.float64) * 7.0 + 0.5
    i0 = np.floor(pos).astype(int)
    f = pos - i0
    enc0 = np.zeros((32, 4))
    for k in range(8):
        c = i0 + np.array([k & 1, (k >> 1) & 1, (k >> 2) & 1])
        w = np.prod(np.where(np.array([k & 1, (k >> 1) & 1, (k >> 2) & 1]) == 1, f, 1 - f), axis=1)
        enc0 += w[:, None] * grid0[(c[:, 0] + 8 * c[:, 1] + 64 * c[:, 2]) % 512]
    # isolate level 0: zero the other levels' grids, compare the first hidden pre-activation's contribution through W1[:, :4]
    q = p.copy()
    q[n_mlp + 512 * 4:] = 0.0
    out = orc.net3_forward(cfg, q, x)
    h = np.maximum(enc0 @ W1[:, :4].T, 0)
    W2 = p[1024:1024 + 4096].reshape(64, 64).astype(np.float64)
    W3 = p[1024 + 4096:1024 + 8192].reshape(64, 64).astype(np.float64)
    Wo = p[1024 + 8192:n_mlp].reshape(48, 64).astype(np.float64)
    h = np.maximum(np.maximum(h @ W2.T, 0) @ W3.T, 0) @ Wo.T
    np.testing.assert_allclose(out, h, rtol=3e-4, atol=3e-5)


def test_net3_backward_matches_finite_differences(orc):
    cfg = _cfg()
    p = _rand_params3(orc, cfg, seed=7)
    rng = np.random.default_rng(3)
    x = rng.uniform(0.05, 0.95, (24, 3)).astype(np.float32)
    dl = np.zeros((24, 48), np.float32)
    dl[:, :41] = rng.normal(size=(24, 41)).astype(np.float32)
    g = orc.net3_backward(cfg, p, x, dl)

    def loss(pp):
        return float((orc.net3_forward(cfg, pp, x).astype(np.float64) * dl).sum())
    n_mlp = 64 * (4 * cfg.n_levels) + 2 * 64 * 64 + 48 * 64
    idx = list(rng.integers(0, n_mlp, 6)) + [int(i) for i in np.flatnonzero(g[n_mlp:])[:6] + n_mlp]
    for i in idx:
        h = 2e-3
        a, b = p.copy(), p.copy()
        a[i] += h
        b[i] -= h
        fd = (loss(a) - loss(b)) / (float(a[i]) - float(b[i]))
        assert abs(fd - g[i]) <= 2e-2 * max(1.0, abs(fd)), (i, fd, g[i])


def test_oracle_guided3_is_deterministic_and_unbiased(orc):
    sd = mixed_cube()
    cfg = _cfg()
    p1 = _rand_params3(orc, cfg, seed=1, wscale=0.3, gscale=0.3)
    p2 = p1.copy()
    gs = guided_settings3(24, 16, 24, 64, EPS, AABB3[0], AABB3[1], train_spp_count=12, batch_size=1024, min_batch_size=256)
    r1 = orc.solve_guided3(sd, gs, cfg, p1, threads=8)
    r2 = orc.solve_guided3(sd, gs, cfg, p2, threads=3)
    assert np.array_equal(r1["field"], r2["field"]) and np.array_equal(p1, p2)
    assert r1["optimizer_steps"] > 0 and r1["guided_steps"] > 0 and r1["neumann_hits"] > 0
    assert r1["walks_started"] == 24 * 16 * 24 == r1["walks_absorbed"] + r1["walks_truncated"]
    # the slice z = 1/2 of u = z, up to the few walks cut at depth 64
    trunc = r1["walks_truncated"] / r1["walks_started"]
    assert abs(float(r1["field"][:, 0].mean()) - 0.5) < 0.02 + 0.5 * trunc
    # phases: no guided depth at all = plain uniform steps; uniform fraction 1 = walks routed to the mixture end
    g0 = guided_settings3(16, 12, 3, 32, EPS, AABB3[0], AABB3[1], train_spp_count=0, max_guided_depth=(0, 0))
    assert orc.solve_guided3(sd, g0, cfg, p1.copy(), threads=4)["guided_steps"] == 0
    g1 = guided_settings3(16, 12, 3, 32, EPS, AABB3[0], AABB3[1], train_spp_count=0, uniform_fraction=(1.0, 1.0))
    assert orc.solve_guided3(sd, g1, cfg, p1.copy(), threads=4)["guided_steps"] == 0


# ---- GPU -------------------------------------------------------------------------------------------------------------
def _hip_cfg(cfg):
    from elaina_amd import capi
    return capi.NetConfig(cfg.n_levels, cfg.n_features, cfg.base_resolution, cfg.per_level_scale, cfg.n_neurons, cfg.n_hidden_layers,
                          cfg.n_output, cfg.learning_rate, cfg.beta1, cfg.beta2, cfg.epsilon, cfg.l2_reg, cfg.ema_decay)


@pytest.mark.gpu
@pytest.mark.parametrize("n_levels,scale", [(4, None), (8, None), (8, 1.45), (8, 0)])
def test_gpu_net3_inference_and_training_match_oracle(orc, n_levels, scale, monkeypatch):
    """four levels: the scalar kernels; eight (the reference's network shape): the matrix-core kernels with the trilinear encoding
    (f32_encode_level3) -- both the oracle's numbers bit for bit.  The grid gradient goes through spatial boxes (grid_bin3_*): 8^3 of
    them with the reference's per-level scale, 16^3 with 1.45 (the finest level's sub-grid of an eighth of the cube no longer fits
    LDS), and (scale 0 here: WOST_GRID_GRAD_BINS=0) through the launches per level group that remain for grids no box size fits"""
    from elaina_amd.guided import GuidingNetwork
    if scale == 0:
        monkeypatch.setenv("WOST_GRID_GRAD_BINS", "0")
        scale = None
    cfg = default_net_config3(n_levels=n_levels) if scale is None else default_net_config3(n_levels=n_levels, per_level_scale=scale)
    net = GuidingNetwork(_hip_cfg(cfg), seed=3, dims=3)
    assert net.n_params == orc.net3_n_params(cfg)
    p = _rand_params3(orc, cfg, seed=11)
    net.set_params(p)
    rng = np.random.default_rng(4)
    x = rng.uniform(-0.05, 1.05, (3000, 3)).astype(np.float32)        # a little outside the unit cube too: the index wraps
    assert np.array_equal(net.inference(x), orc.net3_forward(cfg, p, x)[:, :41])
    # two Adam steps on random loss gradients: gradients, weights and EMA weights bit for bit
    state = orc.net_optimizer_state(cfg)
    for k in state:
        state[k] = np.zeros(len(p), state[k].dtype)
    po = p.copy()
    for step in (1, 2):
        dl = rng.normal(size=(len(x), 41)).astype(np.float32)
        dl48 = np.zeros((len(x), 48), np.float32)
        dl48[:, :41] = dl
        g = orc.net3_backward(cfg, po, x, dl48)
        net.train_step(x, dl, 128.0, apply_update=True)
        assert np.array_equal(net.gradients(), g)
        inf = orc.net3_optimizer_step(cfg, po, state, g, step, 128.0)
        assert np.array_equal(net.params(), po) and np.array_equal(net.inference_params(), inf)
    net.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_features,n_levels", [(2, 8), (8, 3)])
def test_gpu_net3_grid_gradient_with_other_feature_counts(orc, n_features, n_levels):
    """the three-input grid gradient through spatial boxes is written for any number of features per level (the reference uses four):
    two and eight features on the scalar kernels, 20 000 points so that a box's points fill several blocks -- gradients and two Adam
    steps against the oracle bit for bit"""
    from elaina_amd.guided import GuidingNetwork
    from oracle.oracle import NetConfig
    cfg = NetConfig(n_levels, n_features, 8, 1.4049999713897705, 64, 3, 41, 48, 0.00800000037997961, 0.8999999761581421, 0.9900000095367432,
                    1.0000000036274937e-15, 9.999999974752427e-07, 0.949999988079071)
    net = GuidingNetwork(_hip_cfg(cfg), seed=3, dims=3)
    n = orc.net3_n_params(cfg)
    assert net.n_params == n
    rng = np.random.default_rng(21)
    n_mlp = 64 * (n_features * n_levels) + 2 * 64 * 64 + 48 * 64
    p = rng.uniform(-0.25, 0.25, n).astype(np.float32)
    p[n_mlp:] = rng.uniform(-0.5, 0.5, n - n_mlp).astype(np.float32)
    net.set_params(p)
    # most points in one corner of the cube (one box holds thousands), the rest everywhere, a few outside
    x = np.concatenate([rng.uniform(0.0, 0.12, (12000, 3)), rng.uniform(-0.02, 1.02, (8000, 3))]).astype(np.float32)
    assert np.array_equal(net.inference(x), orc.net3_forward(cfg, p, x)[:, :41])
    state = orc.net_optimizer_state(cfg)
    for k in state:
        state[k] = np.zeros(len(p), state[k].dtype)
    po = p.copy()
    for step in (1, 2):
        dl = rng.normal(size=(len(x), 41)).astype(np.float32)
        dl48 = np.zeros((len(x), 48), np.float32)
        dl48[:, :41] = dl
        g = orc.net3_backward(cfg, po, x, dl48)
        net.train_step(x, dl, 128.0, apply_update=True)
        assert np.array_equal(net.gradients(), g)
        inf = orc.net3_optimizer_step(cfg, po, state, g, step, 128.0)
        assert np.array_equal(net.params(), po) and np.array_equal(net.inference_params(), inf)
    net.close()


def _gpu_and_oracle3(orc, sd, w, h, spp, depth, train_spp, uf=(0.5, 0.5), mgd=(10, 10), batch=1024, min_batch=256, params=None,
                     stride=1, offset=0, dump=True, cfg=None, ref=None):
    from elaina_amd.guided import GuidedIntegratorSettings
    from elaina_amd.integrator3d import GuidedIntegrator3, Problem3
    cfg = cfg or _cfg()
    st = GuidedIntegratorSettings(frameSize=(w, h), samplesPerPixel=spp, trainSppCount=train_spp, maxWalkingDepth=depth, epsilonShell=EPS,
                                  uniformFractionInTrainingPhase=uf[0], uniformFractionInGuidingPhase=uf[1],
                                  maxGuidedDepthInTrainingPhase=mgd[0], maxGuidedDepthInGuidingPhase=mgd[1], batchSize=batch,
                                  minBatchSize=min_batch, trainPixelStride=stride, trainPixelOffset=offset)
    gi = GuidedIntegrator3(Problem3.from_dict(sd), st, AABB3, network_config=_hip_cfg(cfg), seed=7)
    if params is not None:
        gi.network.set_params(params)
    p0 = gi.network.params()
    gi.solve()
    if ref is not None:      # (the oracle's half of this solve has been computed already)
        return gi, ref
    gs = guided_settings3(w, h, spp, depth, EPS, AABB3[0], AABB3[1], train_spp_count=train_spp, uniform_fraction=uf, max_guided_depth=mgd,
                          batch_size=batch, min_batch_size=min_batch, train_pixel_stride=stride, train_pixel_offset=offset)
    dump_spp = min(train_spp, spp) - 1 if (dump and train_spp > 0) else -1
    trained = p0.copy()
    ref = orc.solve_guided3(sd, gs, cfg, trained, threads=16, dump_spp=dump_spp)
    ref["params"] = trained
    return gi, ref


COUNTERS = ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated", "neumann_hits", "guided_steps")


@pytest.mark.gpu
@pytest.mark.parametrize("spp,uf,mgd", [(4, (0.5, 0.5), (10, 10)), (2, (0.0, 0.0), (10, 10)), (2, (0.9, 0.25), (3, 3)), (3, (0.5, 0.5), (0, 0)),
                                        (2, (1.0, 1.0), (10, 10))])
def test_gpu_frozen_network_walks_match_oracle(orc, spp, uf, mgd):
    """training off, a random network with pronounced lobes: routing, vMF mixture sampling, MIS pdf, reflection about the
    Neumann normals of the cube's side faces, throughput -- bit-exact; also with no guided depth (plain steps, R_B without
    the 0.99 factor) and with uniform fraction 1 (walks routed to the mixture end, :1031)"""
    sd = mixed_cube()
    p = _rand_params3(orc, _cfg(), seed=3, wscale=0.3, gscale=1.0)
    gi, ref = _gpu_and_oracle3(orc, sd, 40, 32, spp, 48, 0, uf=uf, mgd=mgd, params=p)
    assert np.array_equal(gi.solution, ref["field"]), float(np.abs(gi.solution - ref["field"]).max())
    for k in COUNTERS:
        assert gi.last_stats[k] == ref[k], k
    assert np.array_equal(gi.network.params(), p)
    gi.close()


@pytest.mark.gpu
def test_gpu_first_pass_records_match_oracle_3d(orc):
    """one training pass without an optimizer step: the 3-D records and the ordered training set, bit for bit; an emissive
    Neumann face adds contributions to the records (recordSourceContribution)"""
    sd = cube_scene3(n=3, d_faces=(4, 5), n_faces=(0, 1, 2, 3), value=lambda x, y, z: z, flux=lambda x, y, z, f: 0.3 * (f - 1.5))
    gi, ref = _gpu_and_oracle3(orc, sd, 40, 40, 1, 48, 1, min_batch=10 ** 9)
    ts, to = gi.train_set(), ref["train_set"]
    assert gi.last_stats["optimizer_steps"] == 0
    assert gi.last_stats["train_samples"] == len(ts["xyz"]) == len(to["xyz"]) == ref["train_samples"] > 1000
    assert np.array_equal(gi.solution, ref["field"])
    for k in ("xyz", "dir", "solution", "dir_pdf", "normal", "on_neumann"):
        assert np.array_equal(ts[k], to[k]), k
    for k in COUNTERS:
        assert gi.last_stats[k] == ref[k], k
    gi.close()


@pytest.mark.gpu
def test_gpu_trained_solve_matches_oracle_3d(orc):
    """8 trained + 8 guided samples: every walk, record, batch, gradient and Adam step equal; training pixels every third
    pixel from offset 1; and the harmonic check on the GPU field"""
    sd = mixed_cube()
    gi, ref = _gpu_and_oracle3(orc, sd, 36, 30, 16, 64, 8, stride=3, offset=1, batch=512, min_batch=128)
    for k in COUNTERS + ("train_samples", "optimizer_steps"):
        assert gi.last_stats[k] == ref[k], k
    assert ref["optimizer_steps"] >= 8
    assert np.array_equal(gi.solution, ref["field"]), float(np.abs(gi.solution - ref["field"]).max())
    assert np.array_equal(gi.network.params(), ref["params"])
    trunc = ref["walks_truncated"] / ref["walks_started"]
    assert abs(float(gi.solution[:, 0].mean()) - 0.5) < 0.02 + 0.5 * trunc
    # queryNetwork(Vector3f) = the inference weights at a world point (exec.cu:175-186 asks for (0, -0.21, 0))
    raw = gi.queryNetwork((0.5, 0.29, 0.5))
    assert raw.shape == (41,) and np.isfinite(raw).all()
    gi.close()


def _with_env(env, fn):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


FUSED_MODES = [({}, "one launch per sample, walkers spread over the lanes (a small frame)"),
               ({"WOST3_G_SHIFT": "0", "WOST3_G_FUSED": "1"}, "one launch per sample, 64 walkers per wave: four units of the matrices at once"),
               ({"WOST3_G_FUSED": "0"}, "the launches per depth"),
               ({"WOST3_G_FUSED": "0", "WOST3_G_SHIFT": "0"}, "the launches per depth, 64 walkers per wave")]


@pytest.mark.gpu
def test_gpu_guided3_reference_network_fused_and_per_depth(orc):
    """the reference's eight-level network (the shape the MFMA kernels and g3_fused_kernel cover; the tests above use four levels
    = the scalar kernels and the launches per depth): a frozen random network and a trained solve with an emissive Neumann face,
    each through the fused kernel (spread walkers: one unit; 64 walkers per wave: four units) and through the launches per
    depth -- all equal to the oracle bit for bit, and the fused solves take a few launches per sample"""
    cfg = default_net_config3()
    sd = mixed_cube()
    p = _rand_params3(orc, cfg, seed=3, wscale=0.3, gscale=1.0)
    ref = None
    for env, what in FUSED_MODES:
        gi, ref = _with_env(env, lambda: _gpu_and_oracle3(orc, sd, 40, 32, 3, 48, 0, params=p, cfg=cfg, ref=ref))
        assert np.array_equal(gi.solution, ref["field"]), (what, float(np.abs(gi.solution - ref["field"]).max()))
        for k in COUNTERS:
            assert gi.last_stats[k] == ref[k], (what, k)
        assert gi.last_stats["guided_steps"] > 1000
        if env.get("WOST3_G_FUSED") != "0":
            assert gi.last_stats["kernel_launches"] <= 3 * 2 + 2, gi.last_stats["kernel_launches"]
        gi.close()
    sd = cube_scene3(n=3, d_faces=(4, 5), n_faces=(0, 1, 2, 3), value=lambda x, y, z: z, flux=lambda x, y, z, f: 0.3 * (f - 1.5))
    ref = None
    for env, what in FUSED_MODES:
        gi, ref = _with_env(env, lambda: _gpu_and_oracle3(orc, sd, 30, 24, 5, 48, 3, batch=512, min_batch=128, stride=2, offset=1, cfg=cfg, ref=ref))
        for k in COUNTERS + ("train_samples", "optimizer_steps"):
            assert gi.last_stats[k] == ref[k], (what, k)
        assert ref["optimizer_steps"] >= 3
        assert np.array_equal(gi.solution, ref["field"]), what
        assert np.array_equal(gi.network.params(), ref["params"]), what
        gi.close()


@pytest.mark.gpu
def test_gpu_guided3_forms_agree_on_a_frame_that_fills_the_chip():
    """tools/probes/g3_forms_equal_at_size.py: the bench's shell scene at 724^2, two trained + two guided samples -- the live lists of the
    launches per depth hold hundreds of thousands of walkers written by thousands of blocks (the other tests' frames are small); field,
    counters and trained parameters equal to those of the one-launch-per-sample form bit for bit"""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "probes", "g3_forms_equal_at_size.py")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, FRAME="724", SPP="4"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "fields equal True, parameters equal True, counters equal" in out.stdout, out.stdout[-2000:]


@pytest.mark.gpu
def test_gpu_guided3_shards_and_refusals(orc):
    """wost3_guided_solve_sharded: with a frozen network the shards' fields add up to the full frame; a 2-D shaped network
    is refused"""
    import torch
    from elaina_amd import capi
    from elaina_amd.guided import GuidedIntegratorSettings
    from elaina_amd.integrator3d import GuidedIntegrator3, Problem3
    sd = mixed_cube()
    p = _rand_params3(orc, _cfg(), seed=9, wscale=0.3, gscale=1.0)
    st = GuidedIntegratorSettings(frameSize=(32, 24), samplesPerPixel=3, trainSppCount=0, maxWalkingDepth=32, epsilonShell=EPS)
    gi = GuidedIntegrator3(Problem3.from_dict(sd), st, AABB3, network_config=_hip_cfg(_cfg()))
    gi.network.set_params(p)
    gi.solve()
    full = gi.solution.copy()
    total = torch.zeros(32 * 24 * 3, device="cuda")
    for r in range(3):
        buf = torch.zeros(32 * 24 * 3, device="cuda")
        gi.solve_sharded(r, 3, buf.data_ptr())
        torch.cuda.synchronize()
        total += buf
    assert np.array_equal(total.cpu().numpy().reshape(-1, 3), full)
    gi.close()
    with pytest.raises(capi.WostError):
        GuidedIntegrator3(Problem3.from_dict(sd), st, AABB3, network_config=capi.NetConfig(4, 4, 8, 1.405, 64, 3, 33, 8e-3, 0.9, 0.99, 1e-15, 1e-6, 0.95))


@pytest.mark.gpu
def test_gpu_guided3_source_term_matches_oracle(orc):
    """sampleSource of the guided integrator in 3-D (guided/integrator.cu:277-364 is templated on DIM): a Poisson problem on the
    mixed cube -- a dense-grid source, Dirichlet u = z on two faces, an EMISSIVE Neumann face among the four others -- with a
    frozen network and with training (records collect the source contributions, recordSourceContribution): fields, counters,
    training set and trained weights bit for bit; and the sign / scale against the analytic solution of -laplace u = f"""
    sd = cube_scene3(n=3, d_faces=(4, 5), n_faces=(0, 1, 2, 3), value=lambda x, y, z: z, flux=lambda x, y, z, f: 0.3 * (f - 1.5))
    rng = np.random.default_rng(4)
    sd["source"] = {"rgb": rng.uniform(-1, 1, (5, 4, 6, 3)).astype(np.float32), "index_scale": (5.0, 3.0, 4.0), "index_offset": (0.0, 0.0, 0.0),
                    "intensity": 0.7}
    p = _rand_params3(orc, _cfg(), seed=3, wscale=0.3, gscale=1.0)
    gi, ref = _gpu_and_oracle3(orc, sd, 36, 28, 3, 48, 0, params=p)
    assert np.array_equal(gi.solution, ref["field"]), float(np.abs(gi.solution - ref["field"]).max())
    for k in COUNTERS:
        assert gi.last_stats[k] == ref[k], k
    gi.close()
    gi, ref = _gpu_and_oracle3(orc, sd, 30, 24, 6, 48, 4, batch=512, min_batch=128)
    for k in COUNTERS + ("train_samples", "optimizer_steps"):
        assert gi.last_stats[k] == ref[k], k
    assert ref["optimizer_steps"] >= 4
    ts, to = gi.train_set(), ref["train_set"]
    for k in ("xyz", "dir", "solution", "dir_pdf", "normal", "on_neumann"):
        assert np.array_equal(ts[k], to[k]), k
    assert np.array_equal(gi.solution, ref["field"]) and np.array_equal(gi.network.params(), ref["params"])
    gi.close()
    # -laplace u = f with f = 6 constant, u = z (1 - z) * 3 + z on the cube with zero flux on the sides: u(z = 1/2) = 3/4 + 1/2
    sd = mixed_cube()
    sd["source"] = {"rgb": np.full((2, 2, 2, 3), 6.0, np.float32), "index_scale": (1.0, 1.0, 1.0), "index_offset": (0.0, 0.0, 0.0), "intensity": 1.0}
    gi, ref = _gpu_and_oracle3(orc, sd, 24, 24, 48, 64, 0, params=p)
    assert np.array_equal(gi.solution, ref["field"])
    trunc = ref["walks_truncated"] / ref["walks_started"]
    assert abs(float(gi.solution[:, 0].mean()) - 1.25) < 0.06 + 1.5 * trunc, float(gi.solution[:, 0].mean())
    gi.close()


# ---- the reference's network precision with three inputs ("precision" / "train_precision" 16, integrator/guided/integrator.h:54) ----
def _half_network3_numpy(orc, cfg, p, x):
    """the half-precision three-input network in numpy: grid values rounded to f16, the oracle's fp32 trilinear interpolation,
    then per layer f16 inputs and weights, fp32 accumulation, ReLU, f16 (tests/test_guided_network.py does the same for two inputs)"""
    n_mlp = 64 * 32 + 2 * 64 * 64 + 48 * 64
    ph = p.copy()
    ph[n_mlp:] = p[n_mlp:].astype(np.float16).astype(np.float32)
    a = orc.net3_forward(cfg, ph, x, want_acts=True)[1][:, :32].astype(np.float16).astype(np.float32)
    off = 0
    for no, ni, relu in [(64, 32, True), (64, 64, True), (64, 64, True), (48, 64, False)]:
        w = p[off:off + no * ni].reshape(no, ni).astype(np.float16).astype(np.float32)
        off += no * ni
        z = (a.astype(np.float64) @ w.T.astype(np.float64)).astype(np.float32)
        a = (np.maximum(z, 0.0) if relu else z).astype(np.float16).astype(np.float32)
    return a[:, :41]


@pytest.mark.gpu
def test_gpu_net3_half_precision_inference_and_training(orc):
    """the three-input network of the reference's shape (eight levels) in half precision: inference equal to the numpy
    restatement up to the summation order inside the matrix instruction and close to the fp32 network; training passes
    (f16 forward / backward / weight gradients, fp32 master weights) reproducible bit for bit, close to the fp32 gradient, and
    a regression learns"""
    from elaina_amd.guided import GuidingNetwork
    cfg = default_net_config3()
    p = _rand_params3(orc, cfg, seed=13, wscale=0.2, gscale=0.4)
    rng = np.random.default_rng(3)
    x = rng.uniform(0.0, 1.0, (5000, 3)).astype(np.float32)
    net = GuidingNetwork(_hip_cfg(cfg), seed=3, dims=3)
    net.set_params(p)
    fp32 = net.inference(x)
    assert np.array_equal(fp32, orc.net3_forward(cfg, p, x)[:, :41])
    net.set_option("precision", 16)
    half = net.inference(x)
    emu = _half_network3_numpy(orc, cfg, p, x)
    scale = float(np.sqrt(np.mean(fp32 ** 2)))
    assert np.abs(half - emu).max() <= 4e-3 * scale and np.mean(half == emu) > 0.9
    assert float(np.sqrt(np.mean((half - fp32) ** 2))) / scale < 5e-3
    assert np.array_equal(half, half.astype(np.float16).astype(np.float32))
    net.set_option("precision", 32)
    assert np.array_equal(net.inference(x), fp32)
    # one training step in both precisions from the same weights: the gradients agree, the half-precision one is reproducible
    dl = (rng.normal(size=(len(x), 41)) * 0.01).astype(np.float32)
    net.train_step(x, dl, 128.0, apply_update=False)
    g32 = net.gradients().copy()
    net.set_option("train_precision", 16)
    net.train_step(x, dl, 128.0, apply_update=False)
    g16 = net.gradients().copy()
    net.train_step(x, dl, 128.0, apply_update=False)
    assert np.array_equal(net.gradients(), g16)
    n_mlp = 64 * 32 + 2 * 64 * 64 + 48 * 64
    for sl in (slice(0, n_mlp), slice(n_mlp, None)):
        ref = g32[sl]
        assert float(np.linalg.norm(g16[sl] - ref)) < 0.08 * float(np.linalg.norm(ref)) + 1e-9
    # a regression through the half-precision passes: the loss falls
    target = lambda q: np.stack([np.sin(3 * q[:, 0]) * q[:, 1], q[:, 2] ** 2] + [0.1 * q[:, 0]] * 39, 1).astype(np.float32)
    net.set_option("precision", 16)
    first = last = None
    for it in range(60):
        xb = rng.uniform(0, 1, (8192, 3)).astype(np.float32)
        pred = net.inference(xb, use_inference_params=False)
        loss = float(np.mean((pred - target(xb)) ** 2))
        first = loss if first is None else first
        last = loss
        net.train_step(xb, (2.0 * (pred - target(xb)) / pred.size * 128.0).astype(np.float32), loss_scale=128.0)
    assert last < 0.3 * first, (first, last)
    net.close()


@pytest.mark.gpu
def test_gpu_guided3_half_precision_solve_is_unbiased_and_reproducible(orc):
    """GuidedIntegrator<3> with the half-precision network (inference and training passes): two solves give the same field and the
    same network, the harmonic check u = z on the mixed cube holds as in fp32, and the field differs from the fp32 mode's"""
    from elaina_amd.guided import GuidedIntegratorSettings
    from elaina_amd.integrator3d import GuidedIntegrator3, Problem3
    sd = mixed_cube()
    out = {}
    for prec in (32, 16, 16):
        st = GuidedIntegratorSettings(frameSize=(40, 40), samplesPerPixel=24, trainSppCount=12, maxWalkingDepth=64, epsilonShell=EPS,
                                      batchSize=2048, minBatchSize=512)
        gi = GuidedIntegrator3(Problem3.from_dict(sd), st, AABB3, network_config=_hip_cfg(default_net_config3()), seed=7)
        if prec == 16:
            gi.network.set_option("precision", 16)
            gi.network.set_option("train_precision", 16)
        gi.solve()
        assert gi.last_stats["optimizer_steps"] > 0 and gi.last_stats["guided_steps"] > 0
        out.setdefault(prec, []).append((gi.solution.copy(), gi.network.params(), dict(gi.last_stats)))
        gi.close()
    (f16a, p16a, s16), (f16b, p16b, _) = out[16]
    f32, _, s32 = out[32][0]
    assert np.array_equal(f16a, f16b) and np.array_equal(p16a, p16b) and not np.array_equal(f16a, f32)
    for f, s in ((f16a, s16), (f32, s32)):
        trunc = s["walks_truncated"] / s["walks_started"]
        assert abs(float(f[:, 0].mean()) - 0.5) < 0.02 + 0.5 * trunc
    r16, r32 = float(np.sqrt(np.mean((f16a[:, 0] - 0.5) ** 2))), float(np.sqrt(np.mean((f32[:, 0] - 0.5) ** 2)))
    assert r16 < 1.25 * r32
