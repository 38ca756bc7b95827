"""Python mirror of UniformIntegrator<2> (reference integrator/uniform/integrator.h:55-131)
on top of the C-ABI.  Same method names and argument meaning as the reference class; every
method is a thin call into libwost_hip.so -- no arithmetic happens here.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import MeshDesc, SceneDesc, Settings, Stats, _check, _fp, _ip


class UniformIntegratorSettings:
    """reference integrator/uniform/integrator.h:27-48 (metric-dump keys are host-only)"""

    def __init__(self, frameSize=(800, 800), samplesPerPixel=512, maxWalkingDepth=32, epsilonShell=1e-5,
                 debugPixel=0, saveSppMetricsDuration=-1, saveSppMetricsUntil=1024, saveTimeMetricsDuration=-1):
        self.frameSize = (int(frameSize[0]), int(frameSize[1]))
        self.samplesPerPixel = int(samplesPerPixel)
        self.maxWalkingDepth = int(maxWalkingDepth)
        self.epsilonShell = float(epsilonShell)
        self.debugPixel = debugPixel
        self.saveSppMetricsDuration = saveSppMetricsDuration
        self.saveSppMetricsUntil = saveSppMetricsUntil
        self.saveTimeMetricsDuration = saveTimeMetricsDuration


def _mesh_desc(keep, verts, segs, colors):
    m = MeshDesc()
    if verts is None or segs is None or len(segs) == 0:
        return m
    keep += [verts, segs]
    m.n_verts, m.n_segs = len(verts), len(segs)
    m.verts, m.segs = _fp(verts), _ip(segs)
    if colors is not None:
        if colors.shape != (len(verts), 6):
            raise ValueError("colors must be [n_verts, 6] (left rgb, right rgb)")
        keep.append(colors)
        m.colors = _fp(colors)
    return m


def scene_desc(keep, problem, w, h):
    """wost_scene_desc of a Problem; `keep` collects the arrays the descriptor points into"""
    sc = SceneDesc()
    sc.dirichlet = _mesh_desc(keep, problem.d_verts, problem.d_segs, problem.d_colors)
    sc.neumann = _mesh_desc(keep, problem.n_verts, problem.n_segs, problem.n_colors)
    sc.dirichlet_intensity = problem.dirichlet_intensity
    sc.neumann_intensity = problem.neumann_intensity
    sc.probe_scale = float(problem.probe[0])
    sc.probe_pos[0], sc.probe_pos[1] = float(problem.probe[1]), float(problem.probe[2])
    sc.probe_up[0], sc.probe_up[1] = float(problem.probe[3]), float(problem.probe[4])
    if problem.mask is not None:
        if problem.mask.size != w * h:
            raise ValueError("mask must have width*height entries")
        keep.append(problem.mask)
        sc.mask = problem.mask.ctypes.data_as(C.POINTER(C.c_uint8))
    if problem.source is not None:
        rgb = problem.source["rgb"]
        keep.append(rgb)
        sc.source.ny, sc.source.nx = rgb.shape[0], rgb.shape[1]
        sc.source.rgb = _fp(rgb)
        sc.source.index_scale[0], sc.source.index_scale[1] = problem.source["index_scale"]
        sc.source.index_offset[0], sc.source.index_offset[1] = problem.source["index_offset"]
        sc.source.intensity = problem.source["intensity"]
    return sc


class UniformIntegrator:
    VectorType = tuple

    def __init__(self, problem, settings, device=0):
        self.lib = capi.load()
        self.problem = problem
        self.settings = settings
        keep = []
        w, h = settings.frameSize
        sc = scene_desc(keep, problem, w, h)
        st = Settings(w, h, settings.samplesPerPixel, settings.maxWalkingDepth, settings.epsilonShell)
        self._handle = C.c_void_p()
        _check(self.lib.wost_create(C.byref(sc), C.byref(st), device, C.byref(self._handle)), "wost_create")
        self.n_pixels = w * h
        self.last_stats = None
        self.solution = None

    # -- reference surface ------------------------------------------------------------------
    def solve(self, pixel_begin=0, pixel_end=None):
        """returns wall milliseconds like the reference; the field is in self.solution"""
        if pixel_end is None:
            pixel_end = self.n_pixels
        field = np.zeros((pixel_end - pixel_begin, 3), dtype=np.float32)
        st = Stats()
        _check(self.lib.wost_solve(self._handle, pixel_begin, pixel_end, _fp(field), C.byref(st)), "wost_solve")
        self.solution = field
        self.last_stats = st.as_dict()
        return int(st.solve_ms)

    def solve_sharded(self, shard_index, shard_count, field_dev_ptr, stream_ptr=None):
        """field_dev_ptr: device pointer (int) to a zero-filled width*height*3 float buffer"""
        st = Stats()
        _check(self.lib.wost_solve_sharded(self._handle, shard_index, shard_count, C.c_void_p(field_dev_ptr),
                                           C.c_void_p(stream_ptr or 0), C.byref(st)), "wost_solve_sharded")
        self.last_stats = st.as_dict()
        return self.last_stats

    def renderDirichletSDF(self):
        out = np.zeros(self.n_pixels, dtype=np.float32)
        _check(self.lib.wost_render_sdf(self._handle, capi.MESH_DIRICHLET, _fp(out)), "wost_render_sdf")
        return out

    def renderSilhouetteSDF(self):
        out = np.zeros(self.n_pixels, dtype=np.float32)
        _check(self.lib.wost_render_sdf(self._handle, capi.MESH_NEUMANN, _fp(out)), "wost_render_sdf")
        return out

    def renderSource(self):
        out = np.zeros((self.n_pixels, 3), dtype=np.float32)
        _check(self.lib.wost_render_source(self._handle, _fp(out)), "wost_render_source")
        return out

    def queryNetwork(self, p):
        raise NotImplementedError("uniform integrator has no network (reference integrator.cu:661-664)")

    # -- lbvh query call sites, batched ---------------------------------------------------------
    def closest_point(self, pts, which=capi.MESH_DIRICHLET):
        p = np.ascontiguousarray(pts, dtype=np.float32)
        n = len(p)
        idx = np.zeros(n, dtype=np.int32)
        dist = np.zeros(n, dtype=np.float32)
        uv = np.zeros(n, dtype=np.float32)
        side = np.zeros(n, dtype=np.int32)
        _check(self.lib.wost_closest_point(self._handle, which, _fp(p), n, _ip(idx), _fp(dist), _fp(uv), _ip(side)),
               "wost_closest_point")
        return idx, dist, uv, side

    def closest_silhouette(self, pts, rmax=None, which=capi.MESH_NEUMANN):
        p = np.ascontiguousarray(pts, dtype=np.float32)
        n = len(p)
        out = np.zeros(n, dtype=np.float32)
        r = None if rmax is None else np.ascontiguousarray(rmax, dtype=np.float32)
        _check(self.lib.wost_closest_silhouette(self._handle, which, _fp(p), _fp(r) if r is not None else None, n,
                                                _fp(out)), "wost_closest_silhouette")
        return out

    def ray_intersect(self, origins, dirs, tmax, which=capi.MESH_NEUMANN):
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        t = np.ascontiguousarray(tmax, dtype=np.float32)
        n = len(o)
        hit = np.zeros(n, dtype=np.int32)
        tt = np.zeros(n, dtype=np.float32)
        idx = np.zeros(n, dtype=np.int32)
        _check(self.lib.wost_ray_intersect(self._handle, which, _fp(o), _fp(d), _fp(t), n, _ip(hit), _fp(tt),
                                           _ip(idx)), "wost_ray_intersect")
        return hit, tt, idx

    def set_option(self, key, value):
        _check(self.lib.wost_set_option(self._handle, key.encode(), float(value)), "wost_set_option")

    def last_launches(self):
        """the launches of the last solve, in order (wost_last_launches): dicts of kind / walkers / walkers_beside / grid / ms /
        walk_steps_done plus "steps" = the walk steps counted between the end of the previous launch and the end of this one"""
        n = C.c_int32(0)
        _check(self.lib.wost_last_launches(self._handle, None, 0, C.byref(n)), "wost_last_launches")
        buf = (capi.LaunchInfo * max(n.value, 1))()
        _check(self.lib.wost_last_launches(self._handle, buf, n.value, C.byref(n)), "wost_last_launches")
        out, prev = [], 0
        for i in range(n.value):
            d = buf[i].as_dict()
            d["kind_name"] = capi.LAUNCH_NAMES.get(d["kind"], str(d["kind"]))
            if d["kind"] != capi.LAUNCH_WAIT:
                d["steps"] = d["walk_steps_done"] - prev
                prev = d["walk_steps_done"]
            out.append(d)
        return out

    def close(self):
        if self._handle:
            self.lib.wost_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


MESH_BUILD_ARRAYS2 = ("nodes", "cones", "segA", "segInv", "segOrig", "segCol", "segVerts", "flat", "flatCol", "sil", "silN", "scanBox", "scanHl", "scanId",
                      "scalars")


def mesh_build_check(verts, segs, colors=None, repeat=1, device=0):
    """wost_mesh_build_check: the device build of a segment mesh (csrc/wost_build2.hip) against the host builder kept as its
    checker -> ({array: differing bytes}, bytes compared, host ms, device ms)"""
    import ctypes as C
    from . import capi
    lib = capi.load()
    v = np.ascontiguousarray(verts, np.float32)
    s_ = np.ascontiguousarray(segs, np.int32)
    c = None if colors is None else np.ascontiguousarray(colors, np.float32)
    m = capi.MeshDesc(len(v), len(s_), v.ctypes.data_as(C.POINTER(C.c_float)), s_.ctypes.data_as(C.POINTER(C.c_int32)),
                      None if c is None else c.ctypes.data_as(C.POINTER(C.c_float)))
    host_ms, dev_ms = C.c_double(0.0), C.c_double(0.0)
    mism = (C.c_int64 * 16)()
    capi._check(lib.wost_mesh_build_check(C.byref(m), device, repeat, C.byref(host_ms), C.byref(dev_ms), mism), "wost_mesh_build_check")
    return {k: int(mism[i]) for i, k in enumerate(MESH_BUILD_ARRAYS2)}, int(mism[15]), host_ms.value, dev_ms.value
