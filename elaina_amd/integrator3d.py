"""Python mirror of Problem<3> / UniformIntegrator<3> (reference core/problem.h:197-260,
integrator/uniform/integrator.h:55-131 with DIM = 3) on top of the wost3_* entry points of the C-ABI.
Thin ctypes calls into libwost_hip.so -- no arithmetic happens here."""
import ctypes as C

import numpy as np

from . import capi
from .capi import Mesh3Desc, Scene3Desc, Settings, Stats, _check, _fp, _ip


class Problem3:
    """Triangle meshes + EvaluationGrid<3> probe.  probe = (scale, pos[3], up[3], right[3])
    (core/evaluation_grid.h:48-55); colors [n_verts, 6] = rgb on the normal side, rgb on the other."""

    def __init__(self, d_verts=None, d_tris=None, d_colors=None, n_verts=None, n_tris=None, n_colors=None,
                 probe=(1.0, (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0)), dirichlet_intensity=1.0, neumann_intensity=1.0,
                 mask=None, source=None):
        def arr(a, dt, cols):
            return None if a is None else np.ascontiguousarray(a, dtype=dt).reshape(-1, cols)
        self.d_verts, self.d_tris, self.d_colors = arr(d_verts, np.float32, 3), arr(d_tris, np.int32, 3), arr(d_colors, np.float32, 6)
        self.n_verts, self.n_tris, self.n_colors = arr(n_verts, np.float32, 3), arr(n_tris, np.int32, 3), arr(n_colors, np.float32, 6)
        self.probe = probe
        self.dirichlet_intensity, self.neumann_intensity = float(dirichlet_intensity), float(neumann_intensity)
        self.mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
        # source term: {"rgb": [nz, ny, nx, 3], "index_scale": (3,), "index_offset": (3,), "intensity": f} or None
        self.source = None if source is None else dict(source, rgb=np.ascontiguousarray(source["rgb"], dtype=np.float32))

    @classmethod
    def from_dict(cls, sd):
        return cls(**{k: sd.get(k) for k in ("d_verts", "d_tris", "d_colors", "n_verts", "n_tris", "n_colors", "mask", "source")},
                   probe=sd["probe"], dirichlet_intensity=sd.get("dirichlet_intensity", 1.0),
                   neumann_intensity=sd.get("neumann_intensity", 1.0))

    def as_dict(self):
        return {k: getattr(self, k) for k in ("d_verts", "d_tris", "d_colors", "n_verts", "n_tris", "n_colors", "probe",
                                              "dirichlet_intensity", "neumann_intensity", "mask", "source")}


def _mesh3(keep, verts, tris, colors):
    m = Mesh3Desc()
    if verts is None or tris is None or len(tris) == 0:
        return m
    keep += [verts, tris]
    m.n_verts, m.n_tris, m.verts, m.tris = len(verts), len(tris), _fp(verts), _ip(tris)
    if colors is not None:
        if colors.shape != (len(verts), 6):
            raise ValueError("colors must be [n_verts, 6]")
        keep.append(colors)
        m.colors = _fp(colors)
    return m


def scene3_desc(keep, problem, w, h):
    """wost3_scene_desc of a Problem3 (arrays referenced from `keep`)"""
    sc = Scene3Desc()
    sc.dirichlet = _mesh3(keep, problem.d_verts, problem.d_tris, problem.d_colors)
    sc.neumann = _mesh3(keep, problem.n_verts, problem.n_tris, problem.n_colors)
    sc.dirichlet_intensity, sc.neumann_intensity = problem.dirichlet_intensity, problem.neumann_intensity
    scale, pos, up, right = problem.probe
    sc.probe_scale = float(scale)
    for k in range(3):
        sc.probe_pos[k], sc.probe_up[k], sc.probe_right[k] = float(pos[k]), float(up[k]), float(right[k])
    if problem.mask is not None:
        if problem.mask.size != w * h:
            raise ValueError("mask must have width*height entries")
        keep.append(problem.mask)
        sc.mask = problem.mask.ctypes.data_as(C.POINTER(C.c_uint8))
    if problem.source is not None:
        rgb = problem.source["rgb"]
        if rgb.ndim != 4 or rgb.shape[3] != 3:
            raise ValueError("source rgb must be [nz, ny, nx, 3]")
        keep.append(rgb)
        sc.source.nz, sc.source.ny, sc.source.nx = rgb.shape[:3]
        sc.source.rgb = _fp(rgb)
        for k in range(3):
            sc.source.index_scale[k] = float(problem.source["index_scale"][k])
            sc.source.index_offset[k] = float(problem.source["index_offset"][k])
        sc.source.intensity = float(problem.source.get("intensity", 1.0))
    return sc


class UniformIntegrator3:
    VectorType = tuple

    def __init__(self, problem, settings, device=0):
        self.lib = capi.load()
        self.problem, self.settings = problem, settings
        keep = []
        w, h = settings.frameSize
        sc = scene3_desc(keep, problem, w, h)
        st = Settings(w, h, settings.samplesPerPixel, settings.maxWalkingDepth, settings.epsilonShell)
        self._handle = C.c_void_p()
        _check(self.lib.wost3_create(C.byref(sc), C.byref(st), device, C.byref(self._handle)), "wost3_create")
        self.n_pixels = w * h
        self.solution, self.last_stats = None, None

    def solve(self, pixel_begin=0, pixel_end=None):
        """returns wall milliseconds like the reference; the field is in self.solution"""
        if pixel_end is None:
            pixel_end = self.n_pixels
        field = np.zeros((pixel_end - pixel_begin, 3), dtype=np.float32)
        st = Stats()
        _check(self.lib.wost3_solve(self._handle, pixel_begin, pixel_end, _fp(field), C.byref(st)), "wost3_solve")
        self.solution, self.last_stats = field, st.as_dict()
        return int(st.solve_ms)

    def render_sdf(self, which=0):
        """renderDirichletSDF (which = 0) / renderSilhouetteSDF (which = 1): one distance per pixel of the frame"""
        out = np.zeros(self.n_pixels, dtype=np.float32)
        _check(self.lib.wost3_render_sdf(self._handle, int(which), _fp(out)), "wost3_render_sdf")
        return out

    def render_source(self):
        """renderSource: intensity * f at every pixel's evaluation point"""
        out = np.zeros((self.n_pixels, 3), dtype=np.float32)
        _check(self.lib.wost3_render_source(self._handle, _fp(out)), "wost3_render_source")
        return out

    def solve_sharded(self, shard_index, shard_count, field_dev_ptr, stream_ptr=None):
        st = Stats()
        _check(self.lib.wost3_solve_sharded(self._handle, shard_index, shard_count, C.c_void_p(field_dev_ptr),
                                            C.c_void_p(stream_ptr or 0), C.byref(st)), "wost3_solve_sharded")
        self.last_stats = st.as_dict()
        return self.last_stats

    def closest_point(self, pts, which=capi.MESH_DIRICHLET):
        p = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
        n = len(p)
        idx, dist, uv, side = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 2), np.float32), np.zeros(n, np.int32)
        _check(self.lib.wost3_closest_point(self._handle, which, _fp(p), n, _ip(idx), _fp(dist), _fp(uv), _ip(side)), "wost3_closest_point")
        return idx, dist, uv, side

    def closest_silhouette(self, pts, rmax=None, which=capi.MESH_NEUMANN):
        p = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(len(p), np.float32)
        r = None if rmax is None else np.ascontiguousarray(rmax, dtype=np.float32)
        _check(self.lib.wost3_closest_silhouette(self._handle, which, _fp(p), _fp(r) if r is not None else None, len(p), _fp(out)),
               "wost3_closest_silhouette")
        return out

    def ray_intersect(self, origins, dirs, tmax, which=capi.MESH_NEUMANN):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tmax, dtype=np.float32)
        n = len(o)
        hit, tt, idx = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        _check(self.lib.wost3_ray_intersect(self._handle, which, _fp(o), _fp(d), _fp(t), n, _ip(hit), _fp(tt), _ip(idx)),
               "wost3_ray_intersect")
        return hit, tt, idx

    def queryNetwork(self, p):
        raise NotImplementedError("uniform integrator has no network (reference integrator.cu:661-664)")

    def close(self):
        if self._handle:
            self.lib.wost3_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


MESH_BUILD_ARRAYS = ("nodes", "tri", "triOrig", "slotOfOrig", "triVerts", "colors", "flat", "flatVerts", "edges", "slotEdges", "cones", "obox",
                     "areas", "sampTri", "scalars")


def mesh_build_check(verts, tris, colors=None, repeat=1, device=0):
    """wost3_mesh_build_check: the device build of a triangle mesh against the host builder kept as its checker ->
    ({array: differing bytes}, bytes compared, host ms, device ms)"""
    keep = []
    verts = np.ascontiguousarray(verts, np.float32)
    tris = np.ascontiguousarray(tris, np.int32)
    colors = None if colors is None else np.ascontiguousarray(colors, np.float32)
    m = _mesh3(keep, verts, tris, colors)
    host_ms, dev_ms = C.c_double(0.0), C.c_double(0.0)
    mism = (C.c_int64 * 16)()
    lib = capi.load()
    _check(lib.wost3_mesh_build_check(C.byref(m), device, repeat, C.byref(host_ms), C.byref(dev_ms), mism), "wost3_mesh_build_check")
    return {k: int(mism[i]) for i, k in enumerate(MESH_BUILD_ARRAYS)}, int(mism[15]), host_ms.value, dev_ms.value


def vmf_eval(kappa, cos_theta, device=0):
    """VMF::eval(cosTheta) (reference util/vmf.h:27-32) for pairs (kappa, cos_theta)"""
    lib = capi.load()
    k = np.ascontiguousarray(kappa, dtype=np.float32)
    c = np.ascontiguousarray(cos_theta, dtype=np.float32)
    out = np.zeros(len(k), np.float32)
    _check(lib.wost3_vmf_eval(device, _fp(k), _fp(c), len(k), _fp(out)), "wost3_vmf_eval")
    return out


def vmf_sample(kappa, mu, seed, per_point=1, device=0):
    """VMF::sample(sampler, mu) (reference util/vmf.h:45-70): per point a PCG32 stream setSeed(seed, 1) and per_point
    consecutive directions about mu"""
    lib = capi.load()
    k = np.ascontiguousarray(kappa, dtype=np.float32)
    m = np.ascontiguousarray(mu, dtype=np.float32).reshape(-1, 3)
    s = np.ascontiguousarray(seed, dtype=np.uint64)
    out = np.zeros((len(k), per_point, 3), np.float32)
    _check(lib.wost3_vmf_sample(device, _fp(k), _fp(m), s.ctypes.data_as(C.POINTER(C.c_uint64)), len(k), per_point, _fp(out)), "wost3_vmf_sample")
    return out


def vmm3_pdf_sample(raw, wi, seed, device=0):
    """VMM<3,8>::pdf(wi) and ::sample (reference distribution.h:279-345) from 40 raw outputs per point"""
    lib = capi.load()
    r = np.ascontiguousarray(raw, dtype=np.float32)
    w = np.ascontiguousarray(wi, dtype=np.float32).reshape(-1, 3)
    s = np.ascontiguousarray(seed, dtype=np.uint64)
    n = len(w)
    pdf, d = np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    _check(lib.wost3_vmm_pdf_sample(device, _fp(r), _fp(w), s.ctypes.data_as(C.POINTER(C.c_uint64)), n, _fp(pdf), _fp(d)), "wost3_vmm_pdf_sample")
    return pdf, d


def vmm3_loss_gradients(raw41, dirs, li, dir_pdf, on_neumann, normal, loss_scale=128.0, device=0):
    """the loss gradient of the 3-D mixture by its 41 raw outputs (reference train.h:492-553) and the likelihood term"""
    lib = capi.load()
    r = np.ascontiguousarray(raw41, dtype=np.float32)
    d = np.ascontiguousarray(dirs, dtype=np.float32)
    l = np.ascontiguousarray(li, dtype=np.float32)
    p = np.ascontiguousarray(dir_pdf, dtype=np.float32)
    o = np.ascontiguousarray(on_neumann, dtype=np.uint8)
    nn = np.ascontiguousarray(normal, dtype=np.float32)
    n = len(l)
    g, lk = np.zeros((n, 41), np.float32), np.zeros(n, np.float32)
    _check(lib.wost3_vmm_loss_gradients(device, _fp(r), _fp(d), _fp(l), _fp(p), o.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(nn), n,
                                        C.c_float(loss_scale), _fp(g), _fp(lk)), "wost3_vmm_loss_gradients")
    return g, lk


def default_net_config3(n_levels=8, base_resolution=8, per_level_scale=1.4049999713897705):
    """the network of GuidedIntegrator<3>: data/ladybug/n.json:49-81 with the 41 outputs of guided/parameters.h:26-33"""
    return capi.NetConfig(n_levels, 4, base_resolution, per_level_scale, 64, 3, 41, 0.00800000037997961, 0.8999999761581421,
                          0.9900000095367432, 1.0000000036274937e-15, 9.999999974752427e-07, 0.949999988079071)


class GuidedIntegrator3:
    """Mirror of GuidedIntegrator<3> (reference integrator/guided/integrator.h:77-256 with DIM = 3; exec.cu:102-122): ctor +
    resetNetwork, solve(), queryNetwork(); `aabb` = scene.aabb ((min xyz), (max xyz)); settings: GuidedIntegratorSettings"""

    def __init__(self, problem, settings, aabb, network_config=None, seed=42, device=0):
        from .guided import _BorrowedNetwork
        self.lib = capi.load()
        self.problem, self.settings = problem, settings
        keep = []
        w, h = settings.frameSize
        sc = scene3_desc(keep, problem, w, h)
        gs = capi.Guided3Settings(w, h, settings.samplesPerPixel, settings.maxWalkingDepth, settings.epsilonShell, settings.trainSppCount,
                                  settings.uniformFractionInTrainingPhase, settings.uniformFractionInGuidingPhase,
                                  settings.maxGuidedDepthInTrainingPhase, settings.maxGuidedDepthInGuidingPhase)
        for k in range(3):
            gs.aabb_min[k], gs.aabb_max[k] = float(aabb[0][k]), float(aabb[1][k])
        gs.max_train_depth, gs.batch_size, gs.min_batch_size = settings.maxTrainDepth, settings.batchSize, settings.minBatchSize
        gs.batches_per_spp, gs.train_pixel_stride = settings.batchPerFrame, settings.trainPixelStride
        gs.train_pixel_offset, gs.loss_scale = settings.trainPixelOffset, settings.lossScale
        self.network_config = network_config or default_net_config3()
        self._handle = C.c_void_p()
        _check(self.lib.wost3_guided_create(C.byref(sc), C.byref(gs), C.byref(self.network_config), seed, device, C.byref(self._handle)),
               "wost3_guided_create")
        nh = C.c_void_p()
        _check(self.lib.wost3_guided_network(self._handle, C.byref(nh)), "wost3_guided_network")
        self.network = _BorrowedNetwork(self.lib, nh, self.network_config, dims=3)
        self.n_pixels = w * h
        self.solution, self.last_stats = None, None

    def close(self):
        if self._handle:
            self.network.close()
            self.lib.wost3_guided_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self):
        field = np.zeros((self.n_pixels, 3), dtype=np.float32)
        st = capi.GuidedStats()
        _check(self.lib.wost3_guided_solve(self._handle, _fp(field), C.byref(st)), "wost3_guided_solve")
        self.solution, self.last_stats = field, st.as_dict()
        return int(st.solve_ms)

    def solve_sharded(self, shard_index, shard_count, field_dev_ptr):
        st = capi.GuidedStats()
        _check(self.lib.wost3_guided_solve_sharded(self._handle, shard_index, shard_count, C.c_void_p(field_dev_ptr), C.byref(st)),
               "wost3_guided_solve_sharded")
        self.last_stats = st.as_dict()
        return self.last_stats

    def queryNetwork(self, p):
        pts = np.ascontiguousarray(p, dtype=np.float32).reshape(-1, 3)
        raw = np.zeros((len(pts), 41), dtype=np.float32)
        _check(self.lib.wost3_guided_query_network(self._handle, _fp(pts), len(pts), _fp(raw)), "wost3_guided_query_network")
        return raw if np.ndim(p) > 1 else raw[0]

    def train_set(self):
        n = C.c_int32()
        _check(self.lib.wost3_guided_train_set(self._handle, 0, C.byref(n), None, None, None, None, None, None), "wost3_guided_train_set")
        m = n.value
        out = {"xyz": np.zeros((m, 3), np.float32), "dir": np.zeros((m, 3), np.float32), "solution": np.zeros((m, 3), np.float32),
               "dir_pdf": np.zeros(m, np.float32), "normal": np.zeros((m, 3), np.float32), "on_neumann": np.zeros(m, np.uint8)}
        if m:
            _check(self.lib.wost3_guided_train_set(self._handle, m, C.byref(n), _fp(out["xyz"]), _fp(out["dir"]), _fp(out["solution"]),
                                                   _fp(out["dir_pdf"]), _fp(out["normal"]),
                                                   out["on_neumann"].ctypes.data_as(C.POINTER(C.c_uint8))), "wost3_guided_train_set")
        return out
