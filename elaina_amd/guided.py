"""Python stub for the guided path's distribution layer of the C-ABI (include/wost.h):
von Mises log-Bessel / pdf / d-dkappa, rejection sampling, and the 8-lobe mixture VMM<2,8>.
Thin ctypes calls; no arithmetic here."""
import ctypes as C

import numpy as np

from . import capi
from .capi import _check, _fp


def _u64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def vonmises_eval(kappa, cos_theta, device=0):
    lib = capi.load()
    k = np.ascontiguousarray(kappa, dtype=np.float32)
    c = np.ascontiguousarray(cos_theta, dtype=np.float32)
    n = len(k)
    out = [np.zeros(n, dtype=np.float32) for _ in range(4)]
    _check(lib.wost_vonmises_eval(device, _fp(k), _fp(c), n, *[_fp(o) for o in out]), "wost_vonmises_eval")
    return dict(zip(("log_i0", "log_i1", "log_pdf", "dlog_dkappa"), out))


def vonmises_sample(kappa, seed, per_point=1, device=0):
    lib = capi.load()
    k = np.ascontiguousarray(kappa, dtype=np.float32)
    s = np.ascontiguousarray(seed, dtype=np.uint64)
    th = np.zeros(len(k) * per_point, dtype=np.float32)
    _check(lib.wost_vonmises_sample(device, _fp(k), _u64(s), len(k), per_point, _fp(th)), "wost_vonmises_sample")
    return th.reshape(len(k), per_point)


def vmm_pdf_sample(raw, wi, seed, device=0):
    lib = capi.load()
    r = np.ascontiguousarray(raw, dtype=np.float32)
    w = np.ascontiguousarray(wi, dtype=np.float32)
    s = np.ascontiguousarray(seed, dtype=np.uint64)
    n = len(w)
    pdf = np.zeros(n, dtype=np.float32)
    d = np.zeros((n, 2), dtype=np.float32)
    _check(lib.wost_vmm_pdf_sample(device, _fp(r), _fp(w), _u64(s), n, _fp(pdf), _fp(d)), "wost_vmm_pdf_sample")
    return pdf, d


def vmm_loss_gradients(raw33, dirs, li, dir_pdf, on_neumann, normal, loss_scale=128.0, device=0):
    lib = capi.load()
    r = np.ascontiguousarray(raw33, dtype=np.float32)
    d = np.ascontiguousarray(dirs, dtype=np.float32)
    l = np.ascontiguousarray(li, dtype=np.float32)
    p = np.ascontiguousarray(dir_pdf, dtype=np.float32)
    o = np.ascontiguousarray(on_neumann, dtype=np.uint8)
    nn = np.ascontiguousarray(normal, dtype=np.float32)
    n = len(l)
    g = np.zeros((n, 33), dtype=np.float32)
    lk = np.zeros(n, dtype=np.float32)
    _check(lib.wost_vmm_loss_gradients(device, _fp(r), _fp(d), _fp(l), _fp(p), o.ctypes.data_as(C.POINTER(C.c_uint8)),
                                       _fp(nn), n, C.c_float(loss_scale), _fp(g), _fp(lk)), "wost_vmm_loss_gradients")
    return g, lk


def default_net_config():
    """Network of the reference's 2-D guided integrator: data/ladybug/n.json:49-81, 33 outputs
    (8 lobes x (kappa, mu.x, mu.y, weight) + selection logit, guided/parameters.h:16-24)."""
    return capi.NetConfig(8, 4, 8, 1.4049999713897705, 64, 3, 33, 0.00800000037997961, 0.8999999761581421,
                          0.9900000095367432, 1.0000000036274937e-15, 9.999999974752427e-07, 0.949999988079071)


class GuidingNetwork:
    """Device-resident guiding network (wost_net_* of include/wost.h): DenseGrid encoding ->
    ReLU MLP, Adam nested in EMA.  Mirrors how integrator/guided/integrator.cu drives
    tiny-cuda-nn: inference() (:560,:597) and one training step (:655-662)."""

    def __init__(self, config=None, seed=1337, device=0, dims=2):
        """dims = 3: the three-input network of GuidedIntegrator<3> (wost3_net_create)"""
        self._lib = capi.load()
        self.config = config or default_net_config()
        self.dims = dims
        self._h = C.c_void_p()
        create = self._lib.wost3_net_create if dims == 3 else self._lib.wost_net_create
        _check(create(device, C.byref(self.config), seed, C.byref(self._h)), "wost_net_create")
        total, mlp = C.c_uint64(), C.c_uint64()
        _check(self._lib.wost_net_n_params(self._h, C.byref(total), C.byref(mlp)), "wost_net_n_params")
        self.n_params, self.n_mlp_params = total.value, mlp.value

    def close(self):
        if self._h:
            self._lib.wost_net_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _get(self, which):
        out = np.zeros(self.n_params, dtype=np.float32)
        _check(self._lib.wost_net_get_params(self._h, which, _fp(out)), "wost_net_get_params")
        return out

    def params(self):
        return self._get(0)

    def inference_params(self):
        return self._get(1)

    def gradients(self):
        return self._get(2)

    def set_params(self, params):
        p = np.ascontiguousarray(params, dtype=np.float32)
        if p.size != self.n_params:
            raise ValueError("expected %d parameters" % self.n_params)
        _check(self._lib.wost_net_set_params(self._h, _fp(p)), "wost_net_set_params")

    def set_option(self, key, value):
        """"precision": 32 (fp32, bit-exact mode, default) or 16 (the reference's half-precision inference);
        "train_precision": 32 (default) or 16 (forward / backward / weight gradients of the training steps on
        f16 matrix instructions, fp32 master weights)"""
        _check(self._lib.wost_net_set_option(self._h, key.encode(), float(value)), "wost_net_set_option")

    def inference(self, xy, use_inference_params=True):
        x = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, getattr(self, "dims", 2))
        out = np.zeros((len(x), self.config.n_output), dtype=np.float32)
        _check(self._lib.wost_net_inference(self._h, _fp(x), len(x), _fp(out), int(use_inference_params)),
               "wost_net_inference")
        return out

    def train_step(self, xy, dl_dout, loss_scale=128.0, apply_update=True):
        x = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, getattr(self, "dims", 2))
        g = np.ascontiguousarray(dl_dout, dtype=np.float32).reshape(len(x), self.config.n_output)
        _check(self._lib.wost_net_train_step(self._h, _fp(x), _fp(g), len(x), loss_scale, int(apply_update)),
               "wost_net_train_step")


class GuidedIntegratorSettings:
    """reference integrator/guided/integrator.h:54-75; the trailing group exposes the reference's
    compile-time training constants (parameters.h:7-14, integrator.h:237-239)"""

    def __init__(self, frameSize=(800, 800), samplesPerPixel=512, trainSppCount=150,
                 uniformFractionInTrainingPhase=0.5, uniformFractionInGuidingPhase=0.5,
                 maxGuidedDepthInTrainingPhase=10, maxGuidedDepthInGuidingPhase=10, maxWalkingDepth=32,
                 epsilonShell=1e-5, debugPixel=0, saveSppMetricsDuration=-1, saveSppMetricsUntil=1024,
                 saveTimeMetricsDuration=-1,
                 maxTrainDepth=3, batchSize=524288, minBatchSize=65536, batchPerFrame=5, trainPixelStride=1,
                 trainPixelOffset=-1, lossScale=128.0):
        self.frameSize = (int(frameSize[0]), int(frameSize[1]))
        self.samplesPerPixel = int(samplesPerPixel)
        self.trainSppCount = int(trainSppCount)
        self.uniformFractionInTrainingPhase = float(uniformFractionInTrainingPhase)
        self.uniformFractionInGuidingPhase = float(uniformFractionInGuidingPhase)
        self.maxGuidedDepthInTrainingPhase = int(maxGuidedDepthInTrainingPhase)
        self.maxGuidedDepthInGuidingPhase = int(maxGuidedDepthInGuidingPhase)
        self.maxWalkingDepth = int(maxWalkingDepth)
        self.epsilonShell = float(epsilonShell)
        self.debugPixel = debugPixel
        self.saveSppMetricsDuration = saveSppMetricsDuration
        self.saveSppMetricsUntil = saveSppMetricsUntil
        self.saveTimeMetricsDuration = saveTimeMetricsDuration
        self.maxTrainDepth, self.batchSize, self.minBatchSize = int(maxTrainDepth), int(batchSize), int(minBatchSize)
        self.batchPerFrame, self.trainPixelStride, self.trainPixelOffset = int(batchPerFrame), int(trainPixelStride), int(trainPixelOffset)
        self.lossScale = float(lossScale)


class _BorrowedNetwork(GuidingNetwork):
    """the integrator's own network: same methods, lifetime owned by the integrator"""

    def __init__(self, lib, handle, config, dims=2):
        self._lib, self._h, self.config, self.dims = lib, handle, config, dims
        total, mlp = C.c_uint64(), C.c_uint64()
        _check(lib.wost_net_n_params(handle, C.byref(total), C.byref(mlp)), "wost_net_n_params")
        self.n_params, self.n_mlp_params = total.value, mlp.value

    def close(self):
        self._h = C.c_void_p()


class GuidedIntegrator:
    """Mirror of GuidedIntegrator<2> (reference integrator/guided/integrator.h:77-256): ctor +
    resetNetwork, solve(), queryNetwork(); `aabb` is scene.aabb of the JSON configuration."""

    def __init__(self, problem, settings, aabb, network_config=None, seed=42, device=0):
        from .integrator import scene_desc
        self.lib = capi.load()
        self.problem, self.settings = problem, settings
        keep = []
        w, h = settings.frameSize
        sc = scene_desc(keep, problem, w, h)
        gs = capi.GuidedSettings(w, h, settings.samplesPerPixel, settings.maxWalkingDepth, settings.epsilonShell,
                                 settings.trainSppCount, settings.uniformFractionInTrainingPhase,
                                 settings.uniformFractionInGuidingPhase, settings.maxGuidedDepthInTrainingPhase,
                                 settings.maxGuidedDepthInGuidingPhase)
        (gs.aabb_min[0], gs.aabb_min[1]), (gs.aabb_max[0], gs.aabb_max[1]) = aabb
        gs.max_train_depth, gs.batch_size, gs.min_batch_size = settings.maxTrainDepth, settings.batchSize, settings.minBatchSize
        gs.batches_per_spp, gs.train_pixel_stride = settings.batchPerFrame, settings.trainPixelStride
        gs.train_pixel_offset, gs.loss_scale = settings.trainPixelOffset, settings.lossScale
        self.network_config = network_config or default_net_config()
        self._handle = C.c_void_p()
        _check(self.lib.wost_guided_create(C.byref(sc), C.byref(gs), C.byref(self.network_config), seed, device,
                                           C.byref(self._handle)), "wost_guided_create")
        nh = C.c_void_p()
        _check(self.lib.wost_guided_network(self._handle, C.byref(nh)), "wost_guided_network")
        self.network = _BorrowedNetwork(self.lib, nh, self.network_config)
        self.n_pixels = w * h
        self.aabb = aabb
        self.solution = None
        self.last_stats = None

    def close(self):
        if self._handle:
            self.network.close()
            self.lib.wost_guided_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        """wost_guided_set_option: "pipeline" 1 = the pipelined training order (sample k + 1 walks with the weights of training
        pass k - 1 while pass k trains on a second stream; statistically equivalent, never the parity mode)"""
        _check(self.lib.wost_guided_set_option(self._handle, key.encode(), float(value)), "wost_guided_set_option")

    def solve(self):
        """returns wall milliseconds like the reference; the field is in self.solution"""
        field = np.zeros((self.n_pixels, 3), dtype=np.float32)
        st = capi.GuidedStats()
        _check(self.lib.wost_guided_solve(self._handle, _fp(field), C.byref(st)), "wost_guided_solve")
        self.solution = field
        self.last_stats = st.as_dict()
        return int(st.solve_ms)

    def solve_sharded(self, shard_index, shard_count, field_dev_ptr):
        """field_dev_ptr: device pointer (int) to a width*height*3 float buffer; returns the stats"""
        st = capi.GuidedStats()
        _check(self.lib.wost_guided_solve_sharded(self._handle, shard_index, shard_count, C.c_void_p(field_dev_ptr),
                                                  C.byref(st)), "wost_guided_solve_sharded")
        self.last_stats = st.as_dict()
        return self.last_stats

    def share_network(self):
        """Train ONE network across all ranks of the initialised torch.distributed group
        (wost_guided_set_sync): the fixed-point gradient buffer lives in a torch tensor, which is
        all-reduced (RCCL over xGMI with the nccl backend) before every Adam step; integer sums
        keep the ranks' networks bit-identical.  Without this call every shard trains its own."""
        import torch
        import torch.distributed as dist
        self._grad = torch.zeros(self.network.n_params, dtype=torch.int64, device="cuda")
        _check(self.lib.wost_net_set_gradient_buffer(self.network._h, C.c_void_p(self._grad.data_ptr())),
               "wost_net_set_gradient_buffer")

        from . import distributed as D
        body = D.make_network_sync(self._grad, torch.cuda.synchronize)

        def sync(user, op, data, count):
            return body(op, data, count)

        self._sync = capi.SYNC_FN(sync)       # keep the trampoline alive
        _check(self.lib.wost_guided_set_sync(self._handle, self._sync, None), "wost_guided_set_sync")

    def train_set(self):
        """training set of the most recent training pass, (pixel, record) order"""
        n = C.c_int32()
        _check(self.lib.wost_guided_train_set(self._handle, 0, C.byref(n), None, None, None, None, None, None),
               "wost_guided_train_set")
        m = n.value
        out = {"xy": np.zeros((m, 2), np.float32), "dir": np.zeros((m, 2), np.float32),
               "solution": np.zeros((m, 3), np.float32), "dir_pdf": np.zeros(m, np.float32),
               "normal": np.zeros((m, 2), np.float32), "on_neumann": np.zeros(m, np.uint8)}
        if m:
            _check(self.lib.wost_guided_train_set(self._handle, m, C.byref(n), _fp(out["xy"]), _fp(out["dir"]),
                                                  _fp(out["solution"]), _fp(out["dir_pdf"]), _fp(out["normal"]),
                                                  out["on_neumann"].ctypes.data_as(C.POINTER(C.c_uint8))),
                   "wost_guided_train_set")
        return out

    def queryNetwork(self, p):
        """raw mixture parameters of the guiding network at world position(s) p (reference
        integrator.cu:566-615 prints the VMM built from them)"""
        pts = np.ascontiguousarray(p, dtype=np.float32).reshape(-1, 2)
        raw = np.zeros((len(pts), 33), dtype=np.float32)
        _check(self.lib.wost_guided_query_network(self._handle, _fp(pts), len(pts), _fp(raw)), "wost_guided_query_network")
        return raw if np.ndim(p) > 1 else raw[0]
