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

    def __init__(self, config=None, seed=1337, device=0):
        self._lib = capi.load()
        self.config = config or default_net_config()
        self._h = C.c_void_p()
        _check(self._lib.wost_net_create(device, C.byref(self.config), seed, C.byref(self._h)), "wost_net_create")
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

    def inference(self, xy, use_inference_params=True):
        x = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((len(x), self.config.n_output), dtype=np.float32)
        _check(self._lib.wost_net_inference(self._h, _fp(x), len(x), _fp(out), int(use_inference_params)),
               "wost_net_inference")
        return out

    def train_step(self, xy, dl_dout, loss_scale=128.0, apply_update=True):
        x = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
        g = np.ascontiguousarray(dl_dout, dtype=np.float32).reshape(len(x), self.config.n_output)
        _check(self._lib.wost_net_train_step(self._h, _fp(x), _fp(g), len(x), loss_scale, int(apply_update)),
               "wost_net_train_step")
