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
