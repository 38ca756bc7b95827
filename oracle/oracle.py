"""ctypes binding of the CPU oracle (oracle/wost_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (elaina_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force=False):
    """Compile the oracle with gcc (plain C, seconds).  The Makefile lists every source and header
    of both libraries, so `make` itself decides what is stale."""
    targets = [os.path.join(_BUILD, "libwost_oracle.so"), os.path.join(_BUILD, "libwost_oracle_libm.so")]
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return targets[0]


class Mesh(C.Structure):
    _fields_ = [
        ("n_verts", C.c_int),
        ("n_segs", C.c_int),
        ("verts", C.POINTER(C.c_float)),
        ("segs", C.POINTER(C.c_int)),
        ("colors", C.POINTER(C.c_float)),
    ]


class Source(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("rgb", C.POINTER(C.c_float)),
        ("index_scale", C.c_float * 2), ("index_offset", C.c_float * 2), ("intensity", C.c_float),
    ]


class Scene(C.Structure):
    _fields_ = [
        ("dirichlet", Mesh),
        ("neumann", Mesh),
        ("dirichlet_intensity", C.c_float),
        ("neumann_intensity", C.c_float),
        ("probe_scale", C.c_float),
        ("probe_pos", C.c_float * 2),
        ("probe_up", C.c_float * 2),
        ("mask", C.POINTER(C.c_ubyte)),
        ("source", Source),
    ]


class Settings(C.Structure):
    _fields_ = [
        ("width", C.c_int),
        ("height", C.c_int),
        ("spp", C.c_int),
        ("max_depth", C.c_int),
        ("eps_shell", C.c_float),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("walk_steps", C.c_uint64),
        ("walks_started", C.c_uint64),
        ("walks_absorbed", C.c_uint64),
        ("walks_truncated", C.c_uint64),
        ("neumann_hits", C.c_uint64),
        ("seconds", C.c_double),
    ]


class Mesh3(C.Structure):
    _fields_ = [("n_verts", C.c_int), ("n_tris", C.c_int), ("verts", C.POINTER(C.c_float)), ("tris", C.POINTER(C.c_int)),
                ("colors", C.POINTER(C.c_float))]


class Source3(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("rgb", C.POINTER(C.c_float)), ("index_scale", C.c_float * 3),
                ("index_offset", C.c_float * 3), ("intensity", C.c_float)]


class Scene3(C.Structure):
    _fields_ = [("dirichlet", Mesh3), ("neumann", Mesh3), ("dirichlet_intensity", C.c_float), ("neumann_intensity", C.c_float),
                ("probe_scale", C.c_float), ("probe_pos", C.c_float * 3), ("probe_up", C.c_float * 3),
                ("probe_right", C.c_float * 3), ("mask", C.POINTER(C.c_ubyte)), ("source", Source3)]


class GuidedSettings(C.Structure):
    """wo_guided_settings; defaults = the reference's constants (see wost_oracle.h)"""
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("spp", C.c_int), ("max_depth", C.c_int), ("eps_shell", C.c_float),
        ("train_spp_count", C.c_int),
        ("uniform_fraction_training", C.c_float), ("uniform_fraction_guiding", C.c_float),
        ("max_guided_depth_training", C.c_int), ("max_guided_depth_guiding", C.c_int),
        ("aabb_min", C.c_float * 2), ("aabb_max", C.c_float * 2),
        ("max_train_depth", C.c_int), ("batch_size", C.c_int), ("min_batch_size", C.c_int), ("batches_per_spp", C.c_int),
        ("train_pixel_stride", C.c_int), ("train_pixel_offset", C.c_int), ("loss_scale", C.c_float),
    ]


def guided_settings(width, height, spp, max_depth, eps, aabb_min, aabb_max, train_spp_count=150,
                    uniform_fraction=(0.5, 0.5), max_guided_depth=(10, 10), max_train_depth=3, batch_size=524288,
                    min_batch_size=65536, batches_per_spp=5, train_pixel_stride=1, train_pixel_offset=0,
                    loss_scale=128.0):
    g = GuidedSettings(width, height, spp, max_depth, eps, train_spp_count, uniform_fraction[0], uniform_fraction[1],
                       max_guided_depth[0], max_guided_depth[1])
    g.aabb_min[0], g.aabb_min[1] = aabb_min
    g.aabb_max[0], g.aabb_max[1] = aabb_max
    g.max_train_depth, g.batch_size, g.min_batch_size, g.batches_per_spp = max_train_depth, batch_size, min_batch_size, batches_per_spp
    g.train_pixel_stride, g.train_pixel_offset, g.loss_scale = train_pixel_stride, train_pixel_offset, loss_scale
    return g


class GuidedSettings3(C.Structure):
    """wo3_guided_settings: the same with a 3-D box"""
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("spp", C.c_int), ("max_depth", C.c_int), ("eps_shell", C.c_float),
        ("train_spp_count", C.c_int),
        ("uniform_fraction_training", C.c_float), ("uniform_fraction_guiding", C.c_float),
        ("max_guided_depth_training", C.c_int), ("max_guided_depth_guiding", C.c_int),
        ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3),
        ("max_train_depth", C.c_int), ("batch_size", C.c_int), ("min_batch_size", C.c_int), ("batches_per_spp", C.c_int),
        ("train_pixel_stride", C.c_int), ("train_pixel_offset", C.c_int), ("loss_scale", C.c_float),
    ]


def guided_settings3(width, height, spp, max_depth, eps, aabb_min, aabb_max, train_spp_count=150,
                     uniform_fraction=(0.5, 0.5), max_guided_depth=(10, 10), max_train_depth=3, batch_size=524288,
                     min_batch_size=65536, batches_per_spp=5, train_pixel_stride=1, train_pixel_offset=0, loss_scale=128.0):
    g = GuidedSettings3(width, height, spp, max_depth, eps, train_spp_count, uniform_fraction[0], uniform_fraction[1],
                        max_guided_depth[0], max_guided_depth[1])
    for k in range(3):
        g.aabb_min[k], g.aabb_max[k] = float(aabb_min[k]), float(aabb_max[k])
    g.max_train_depth, g.batch_size, g.min_batch_size, g.batches_per_spp = max_train_depth, batch_size, min_batch_size, batches_per_spp
    g.train_pixel_stride, g.train_pixel_offset, g.loss_scale = train_pixel_stride, train_pixel_offset, loss_scale
    return g


class GuidedStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated",
                                          "neumann_hits", "guided_steps", "train_samples", "optimizer_steps")]


class TrainDump(C.Structure):
    _fields_ = [("capacity", C.c_int), ("n", C.c_int), ("xy", C.POINTER(C.c_float)), ("dir", C.POINTER(C.c_float)),
                ("solution", C.POINTER(C.c_float)), ("dir_pdf", C.POINTER(C.c_float)), ("normal", C.POINTER(C.c_float)),
                ("on_neumann", C.POINTER(C.c_ubyte))]


class NetConfig(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int), ("n_features", C.c_int), ("base_resolution", C.c_int), ("per_level_scale", C.c_float),
        ("n_neurons", C.c_int), ("n_hidden_layers", C.c_int), ("n_output", C.c_int), ("n_output_padded", C.c_int),
        ("learning_rate", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float),
        ("l2_reg", C.c_float), ("ema_decay", C.c_float),
    ]


def default_net_config():
    """data/ladybug/n.json:49-81 of the reference + guided/parameters.h:16-24"""
    return NetConfig(8, 4, 8, 1.4049999713897705, 64, 3, 33, 48, 0.00800000037997961, 0.8999999761581421,
                     0.9900000095367432, 1.0000000036274937e-15, 9.999999974752427e-07, 0.949999988079071)


def default_net_config3(n_levels=8, base_resolution=8, per_level_scale=1.4049999713897705):
    """the same network with the 41 outputs of GuidedIntegrator<3> (guided/parameters.h:26-33: 8 lobes x (lambda, kappa, mean
    vector) + the selection logit); its input has three components (wo_net3_*)"""
    return NetConfig(n_levels, 4, base_resolution, per_level_scale, 64, 3, 41, 48, 0.00800000037997961, 0.8999999761581421,
                     0.9900000095367432, 1.0000000036274937e-15, 9.999999974752427e-07, 0.949999988079071)


class Pcg(C.Structure):
    _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64)]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class Oracle:
    def __init__(self, libm=False):
        build()
        name = "libwost_oracle_libm.so" if libm else "libwost_oracle.so"
        self.lib = C.CDLL(os.path.join(_BUILD, name))
        L = self.lib
        L.wo_version.restype = C.c_char_p
        L.wo_pcg_next_uint.restype = C.c_uint32
        L.wo_pcg_next_float.restype = C.c_float
        L.wo_pcg_next_double.restype = C.c_double
        L.wo_interleave_32bit.restype = C.c_uint32
        L.wo_interleave_32bit.argtypes = [C.c_uint32, C.c_uint32]
        L.wo_pcg_set_seed.argtypes = [C.POINTER(Pcg), C.c_uint64, C.c_uint64]
        L.wo_pcg_advance.argtypes = [C.POINTER(Pcg), C.c_int64]
        L.wo_pcg_seed_pixel.argtypes = [C.POINTER(Pcg), C.c_int, C.c_int]
        L.wo_logf.restype = C.c_float
        L.wo_logf.argtypes = [C.c_float]
        L.wo_sincos_2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        self._keep = []

    def version(self):
        return self.lib.wo_version().decode()

    # ---- helpers to build the C structs from numpy arrays -------------------
    def _mesh(self, verts, segs, colors):
        m = Mesh()
        if verts is None or segs is None or len(segs) == 0:
            m.n_verts = 0
            m.n_segs = 0
            return m
        v = np.ascontiguousarray(verts, dtype=np.float32)
        s = np.ascontiguousarray(segs, dtype=np.int32)
        self._keep += [v, s]
        m.n_verts, m.n_segs = len(v), len(s)
        m.verts, m.segs = _fp(v), _ip(s)
        if colors is not None:
            c = np.ascontiguousarray(colors, dtype=np.float32)
            assert c.shape == (len(v), 6)
            self._keep.append(c)
            m.colors = _fp(c)
        return m

    def make_scene(self, sd):
        """sd: dict with d_verts,d_segs,d_colors,n_verts,n_segs,n_colors(optional),probe,
        dirichlet_intensity, neumann_intensity, mask (optional)"""
        self._keep = []
        sc = Scene()
        sc.dirichlet = self._mesh(sd.get("d_verts"), sd.get("d_segs"), sd.get("d_colors"))
        sc.neumann = self._mesh(sd.get("n_verts"), sd.get("n_segs"), sd.get("n_colors"))
        sc.dirichlet_intensity = float(sd.get("dirichlet_intensity", 1.0))
        sc.neumann_intensity = float(sd.get("neumann_intensity", 1.0))
        p = np.asarray(sd["probe"], dtype=np.float32)
        sc.probe_scale = float(p[0])
        sc.probe_pos[0], sc.probe_pos[1] = float(p[1]), float(p[2])
        sc.probe_up[0], sc.probe_up[1] = float(p[3]), float(p[4])
        mask = sd.get("mask")
        if mask is not None:
            mk = np.ascontiguousarray(mask, dtype=np.uint8)
            self._keep.append(mk)
            sc.mask = mk.ctypes.data_as(C.POINTER(C.c_ubyte))
        src = sd.get("source")
        if src is not None:
            rgb = np.ascontiguousarray(src["rgb"], dtype=np.float32)
            assert rgb.ndim == 3 and rgb.shape[2] == 3
            self._keep.append(rgb)
            sc.source.ny, sc.source.nx = rgb.shape[0], rgb.shape[1]
            sc.source.rgb = _fp(rgb)
            sc.source.index_scale[0], sc.source.index_scale[1] = [float(v) for v in src["index_scale"]]
            sc.source.index_offset[0], sc.source.index_offset[1] = [float(v) for v in src["index_offset"]]
            sc.source.intensity = float(src.get("intensity", 1.0))
        return sc

    # ---- solver ----------------------------------------------------------------
    def solve(self, sd, width, height, spp, max_depth, eps, pixel_begin=0, pixel_end=None, threads=8,
              want_steps=False, want_hist=False):
        sc = self.make_scene(sd)
        st = Settings(width, height, spp, max_depth, eps)
        if pixel_end is None:
            pixel_end = width * height
        n = pixel_end - pixel_begin
        field = np.zeros((n, 3), dtype=np.float32)
        steps = np.zeros(n, dtype=np.uint32) if want_steps else None
        hist = np.zeros(max_depth, dtype=np.uint64) if want_hist else None
        stats = Stats()
        rc = self.lib.wo_solve(
            C.byref(sc), C.byref(st), pixel_begin, pixel_end, threads, _fp(field),
            steps.ctypes.data_as(C.POINTER(C.c_uint32)) if want_steps else None,
            hist.ctypes.data_as(C.POINTER(C.c_uint64)) if want_hist else None,
            C.byref(stats),
        )
        if rc != 0:
            raise RuntimeError("wo_solve failed: %d" % rc)
        out = {
            "field": field,
            "walk_steps": int(stats.walk_steps),
            "walks_started": int(stats.walks_started),
            "walks_absorbed": int(stats.walks_absorbed),
            "walks_truncated": int(stats.walks_truncated),
            "neumann_hits": int(stats.neumann_hits),
            "seconds": float(stats.seconds),
        }
        if want_steps:
            out["steps"] = steps
        if want_hist:
            out["depth_hist"] = hist
        return out

    def solve_guided(self, sd, gs, net_cfg, params, threads=8, dump_spp=-1):
        """Guided integrator; params (float32, n_params) are trained IN PLACE.  Returns field, stats and,
        when dump_spp >= 0, the training set built after that sample pass."""
        sc = self.make_scene(sd)
        n = gs.width * gs.height
        field = np.zeros((n, 3), dtype=np.float32)
        stats = GuidedStats()
        assert params.dtype == np.float32 and params.flags.c_contiguous and params.size == self.net_n_params(net_cfg)
        dump = None
        arrays = {}
        if dump_spp >= 0:
            cap = n * 4
            arrays = {"xy": np.zeros((cap, 2), np.float32), "dir": np.zeros((cap, 2), np.float32),
                      "solution": np.zeros((cap, 3), np.float32), "dir_pdf": np.zeros(cap, np.float32),
                      "normal": np.zeros((cap, 2), np.float32), "on_neumann": np.zeros(cap, np.uint8)}
            dump = TrainDump(cap, 0, _fp(arrays["xy"]), _fp(arrays["dir"]), _fp(arrays["solution"]), _fp(arrays["dir_pdf"]),
                             _fp(arrays["normal"]), arrays["on_neumann"].ctypes.data_as(C.POINTER(C.c_ubyte)))
        rc = self.lib.wo_solve_guided(C.byref(sc), C.byref(gs), C.byref(net_cfg), _fp(params), threads, _fp(field),
                                      C.byref(stats), dump_spp, C.byref(dump) if dump is not None else None)
        if rc != 0:
            raise RuntimeError("wo_solve_guided failed: %d" % rc)
        out = {"field": field}
        out.update({k: int(getattr(stats, k)) for k, _ in GuidedStats._fields_})
        if dump is not None:
            out["train_set"] = {k: v[:dump.n] for k, v in arrays.items()}
        return out

    def solve_guided3(self, sd, gs, net_cfg, params, threads=8, dump_spp=-1):
        """GuidedIntegrator<3> (wo3_solve_guided); params (float32, net3_n_params) are trained IN PLACE"""
        sc = self.make_scene3(sd)
        n = gs.width * gs.height
        field = np.zeros((n, 3), dtype=np.float32)
        stats = GuidedStats()
        assert params.dtype == np.float32 and params.flags.c_contiguous and params.size == self.net3_n_params(net_cfg)
        dump = None
        arrays = {}
        if dump_spp >= 0:
            cap = n * 4
            arrays = {"xyz": np.zeros((cap, 3), np.float32), "dir": np.zeros((cap, 3), np.float32),
                      "solution": np.zeros((cap, 3), np.float32), "dir_pdf": np.zeros(cap, np.float32),
                      "normal": np.zeros((cap, 3), np.float32), "on_neumann": np.zeros(cap, np.uint8)}
            dump = TrainDump(cap, 0, _fp(arrays["xyz"]), _fp(arrays["dir"]), _fp(arrays["solution"]), _fp(arrays["dir_pdf"]),
                             _fp(arrays["normal"]), arrays["on_neumann"].ctypes.data_as(C.POINTER(C.c_ubyte)))
        rc = self.lib.wo3_solve_guided(C.byref(sc), C.byref(gs), C.byref(net_cfg), _fp(params), threads, _fp(field),
                                       C.byref(stats), dump_spp, C.byref(dump) if dump is not None else None)
        if rc != 0:
            raise RuntimeError("wo3_solve_guided failed: %d" % rc)
        out = {"field": field}
        out.update({k: int(getattr(stats, k)) for k, _ in GuidedStats._fields_})
        if dump is not None:
            out["train_set"] = {k: v[:dump.n] for k, v in arrays.items()}
        return out

    # ---- 3-D uniform path (oracle/wost_oracle3d.c) -------------------------------------------
    def _mesh3(self, verts, tris, colors):
        m = Mesh3()
        if verts is None or tris is None or len(tris) == 0:
            return m
        v = np.ascontiguousarray(verts, dtype=np.float32)
        t = np.ascontiguousarray(tris, dtype=np.int32)
        self._keep += [v, t]
        m.n_verts, m.n_tris, m.verts, m.tris = len(v), len(t), _fp(v), _ip(t)
        if colors is not None:
            c = np.ascontiguousarray(colors, dtype=np.float32)
            assert c.shape == (len(v), 6)
            self._keep.append(c)
            m.colors = _fp(c)
        return m

    def make_scene3(self, sd):
        """sd: d_verts [n,3], d_tris [m,3], d_colors [n,6], n_verts, n_tris, n_colors, probe = (scale, pos[3],
        up[3], right[3]), intensities, mask"""
        self._keep = []
        sc = Scene3()
        sc.dirichlet = self._mesh3(sd.get("d_verts"), sd.get("d_tris"), sd.get("d_colors"))
        sc.neumann = self._mesh3(sd.get("n_verts"), sd.get("n_tris"), sd.get("n_colors"))
        sc.dirichlet_intensity = float(sd.get("dirichlet_intensity", 1.0))
        sc.neumann_intensity = float(sd.get("neumann_intensity", 1.0))
        scale, pos, up, right = sd["probe"]
        sc.probe_scale = float(scale)
        for k in range(3):
            sc.probe_pos[k], sc.probe_up[k], sc.probe_right[k] = float(pos[k]), float(up[k]), float(right[k])
        mask = sd.get("mask")
        if mask is not None:
            mk = np.ascontiguousarray(mask, dtype=np.uint8)
            self._keep.append(mk)
            sc.mask = mk.ctypes.data_as(C.POINTER(C.c_ubyte))
        src = sd.get("source")
        if src is not None:
            # {"rgb": [nz, ny, nx, 3], "index_scale": (sx, sy, sz), "index_offset": (ox, oy, oz), "intensity": f}
            rgb = np.ascontiguousarray(src["rgb"], dtype=np.float32)
            self._keep.append(rgb)
            sc.source.nz, sc.source.ny, sc.source.nx = rgb.shape[:3]
            sc.source.rgb = _fp(rgb)
            for k in range(3):
                sc.source.index_scale[k] = float(src["index_scale"][k])
                sc.source.index_offset[k] = float(src["index_offset"][k])
            sc.source.intensity = float(src.get("intensity", 1.0))
        return sc

    def source_eval3(self, sd, pts):
        sc = self.make_scene3(sd)
        p = np.ascontiguousarray(pts, dtype=np.float32)
        out = np.zeros((len(p), 3), np.float32)
        tmp = (C.c_float * 3)()
        for i in range(len(p)):
            self.lib.wo3_source_eval(C.byref(sc.source), C.c_float(p[i, 0]), C.c_float(p[i, 1]), C.c_float(p[i, 2]), tmp)
            out[i] = tmp[:]
        return out

    def solve3(self, sd, width, height, spp, max_depth, eps, pixel_begin=0, pixel_end=None, threads=8):
        sc = self.make_scene3(sd)
        st = Settings(width, height, spp, max_depth, eps)
        if pixel_end is None:
            pixel_end = width * height
        field = np.zeros((pixel_end - pixel_begin, 3), dtype=np.float32)
        stats = Stats()
        rc = self.lib.wo3_solve(C.byref(sc), C.byref(st), pixel_begin, pixel_end, threads, _fp(field), C.byref(stats))
        if rc != 0:
            raise RuntimeError("wo3_solve failed: %d" % rc)
        out = {"field": field}
        out.update({k: int(getattr(stats, k)) for k in ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated",
                                                         "neumann_hits")})
        return out

    def render_sdf3(self, sd, width, height, which):
        sc = self.make_scene3(sd)
        st = Settings(width, height, 1, 1, 1e-3)
        out = np.zeros(width * height, np.float32)
        if self.lib.wo3_render_sdf(C.byref(sc), C.byref(st), int(which), _fp(out)) != 0:
            raise RuntimeError("wo3_render_sdf failed")
        return out

    def render_source3(self, sd, width, height):
        sc = self.make_scene3(sd)
        st = Settings(width, height, 1, 1, 1e-3)
        out = np.zeros((width * height, 3), np.float32)
        if self.lib.wo3_render_source(C.byref(sc), C.byref(st), _fp(out)) != 0:
            raise RuntimeError("wo3_render_source failed")
        return out

    def closest_point3(self, verts, tris, pts):
        self._keep = []
        m = self._mesh3(verts, tris, None)
        p = np.ascontiguousarray(pts, dtype=np.float32)
        n = len(p)
        idx, dist, uv, side = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 2), np.float32), np.zeros(n, np.int32)
        rc = self.lib.wo3_closest_point_batch(C.byref(m), _fp(p), n, _ip(idx), _fp(dist), _fp(uv), _ip(side))
        if rc != 0:
            raise RuntimeError("wo3_closest_point_batch failed")
        return idx, dist, uv, side

    def closest_silhouette3(self, verts, tris, pts, rmax=None):
        self._keep = []
        m = self._mesh3(verts, tris, None)
        p = np.ascontiguousarray(pts, dtype=np.float32)
        out = np.zeros(len(p), np.float32)
        r = None if rmax is None else np.ascontiguousarray(rmax, dtype=np.float32)
        rc = self.lib.wo3_closest_silhouette_batch(C.byref(m), _fp(p), _fp(r) if r is not None else None, len(p), _fp(out))
        if rc != 0:
            raise RuntimeError("wo3_closest_silhouette_batch failed")
        return out

    def ray_intersect3(self, verts, tris, origins, dirs, tmax):
        self._keep = []
        m = self._mesh3(verts, tris, None)
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        t = np.ascontiguousarray(tmax, dtype=np.float32)
        n = len(o)
        hit, tt, idx = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        rc = self.lib.wo3_ray_intersect_batch(C.byref(m), _fp(o), _fp(d), _fp(t), n, _ip(hit), _fp(tt), _ip(idx))
        if rc != 0:
            raise RuntimeError("wo3_ray_intersect_batch failed")
        return hit, tt, idx

    def vmf_eval(self, kappa, cos_theta):
        k = np.ascontiguousarray(kappa, dtype=np.float32)
        c = np.ascontiguousarray(cos_theta, dtype=np.float32)
        out = np.zeros(len(k), dtype=np.float32)
        self.lib.wo3_vmf_eval_batch(_fp(k), _fp(c), len(k), _fp(out))
        return out

    def vmf_sample(self, kappa, mu, seed, per_point=1):
        k = np.ascontiguousarray(kappa, dtype=np.float32)
        m = np.ascontiguousarray(mu, dtype=np.float32)
        s = np.ascontiguousarray(seed, dtype=np.uint64)
        out = np.zeros((len(k), per_point, 3), dtype=np.float32)
        self.lib.wo3_vmf_sample_batch(_fp(k), _fp(m), s.ctypes.data_as(C.POINTER(C.c_uint64)), len(k), per_point, _fp(out))
        return out

    def vmm3_pdf_sample(self, raw, wi, seed):
        r = np.ascontiguousarray(raw, dtype=np.float32)
        w = np.ascontiguousarray(wi, dtype=np.float32)
        s = np.ascontiguousarray(seed, dtype=np.uint64)
        n = len(w)
        pdf = np.zeros(n, dtype=np.float32)
        d = np.zeros((n, 3), dtype=np.float32)
        self.lib.wo3_vmm_pdf_sample(_fp(r), _fp(w), s.ctypes.data_as(C.POINTER(C.c_uint64)), n, _fp(pdf), _fp(d))
        return pdf, d

    def vmm3_loss_gradients(self, raw41, dirs, li, dir_pdf, on_neumann, normal, loss_scale=128.0):
        r = np.ascontiguousarray(raw41, dtype=np.float32)
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        l = np.ascontiguousarray(li, dtype=np.float32)
        p = np.ascontiguousarray(dir_pdf, dtype=np.float32)
        o = np.ascontiguousarray(on_neumann, dtype=np.uint8)
        nn = np.ascontiguousarray(normal, dtype=np.float32)
        n = len(l)
        g = np.zeros((n, 41), dtype=np.float32)
        lk = np.zeros(n, dtype=np.float32)
        self.lib.wo3_vmm_loss_gradients(_fp(r), _fp(d), _fp(l), _fp(p), o.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(nn), n,
                                        C.c_float(loss_scale), _fp(g), _fp(lk))
        return g, lk

    def green_ball3(self, R, r):
        e, nrm, pdf = C.c_float(), C.c_float(), C.c_float()
        self.lib.wo3_green_ball.restype = None
        self.lib.wo3_green_ball(C.c_float(R), C.c_float(r), C.byref(e), C.byref(nrm), C.byref(pdf))
        return e.value, nrm.value, pdf.value

    def render_source(self, sd, width, height):
        sc = self.make_scene(sd)
        st = Settings(width, height, 1, 1, 1.0)
        out = np.zeros((width * height, 3), dtype=np.float32)
        rc = self.lib.wo_render_source(C.byref(sc), C.byref(st), _fp(out))
        if rc != 0:
            raise RuntimeError("wo_render_source failed: %d" % rc)
        return out

    def render_dirichlet_sdf(self, sd, width, height, threads=8):
        sc = self.make_scene(sd)
        st = Settings(width, height, 1, 1, 1.0)
        out = np.zeros(width * height, dtype=np.float32)
        rc = self.lib.wo_render_dirichlet_sdf(C.byref(sc), C.byref(st), threads, _fp(out))
        if rc != 0:
            raise RuntimeError("wo_render_dirichlet_sdf failed: %d" % rc)
        return out

    # ---- batch queries -----------------------------------------------------------
    def closest_point(self, verts, segs, pts, mode=1):
        self._keep = []
        m = self._mesh(verts, segs, None)
        p = np.ascontiguousarray(pts, dtype=np.float32)
        n = len(p)
        idx = np.zeros(n, dtype=np.int32)
        dist = np.zeros(n, dtype=np.float32)
        uv = np.zeros(n, dtype=np.float32)
        side = np.zeros(n, dtype=np.int32)
        rc = self.lib.wo_closest_point_batch(C.byref(m), _fp(p), n, mode, _ip(idx), _fp(dist), _fp(uv), _ip(side))
        if rc != 0:
            raise RuntimeError("wo_closest_point_batch failed: %d" % rc)
        return idx, dist, uv, side

    def closest_silhouette(self, verts, segs, pts, rmax=None):
        self._keep = []
        m = self._mesh(verts, segs, None)
        p = np.ascontiguousarray(pts, dtype=np.float32)
        n = len(p)
        out = np.zeros(n, dtype=np.float32)
        r = None
        if rmax is not None:
            r = np.ascontiguousarray(rmax, dtype=np.float32)
        rc = self.lib.wo_closest_silhouette_batch(C.byref(m), _fp(p), _fp(r) if r is not None else None, n, _fp(out))
        if rc != 0:
            raise RuntimeError("wo_closest_silhouette_batch failed: %d" % rc)
        return out

    def ray_intersect(self, verts, segs, origins, dirs, tmax):
        self._keep = []
        m = self._mesh(verts, segs, None)
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        t = np.ascontiguousarray(tmax, dtype=np.float32)
        n = len(o)
        hit = np.zeros(n, dtype=np.int32)
        tt = np.zeros(n, dtype=np.float32)
        idx = np.zeros(n, dtype=np.int32)
        rc = self.lib.wo_ray_intersect_batch(C.byref(m), _fp(o), _fp(d), _fp(t), n, _ip(hit), _fp(tt), _ip(idx))
        if rc != 0:
            raise RuntimeError("wo_ray_intersect_batch failed: %d" % rc)
        return hit, tt, idx

    # ---- scalar helpers for KATs ---------------------------------------------------
    def pcg_seed(self, initstate, initseq):
        r = Pcg()
        self.lib.wo_pcg_set_seed(C.byref(r), initstate, initseq)
        return r

    def pcg_seed_pixel(self, pixel_id, width):
        r = Pcg()
        self.lib.wo_pcg_seed_pixel(C.byref(r), pixel_id, width)
        return r

    def pcg_uint(self, r):
        return int(self.lib.wo_pcg_next_uint(C.byref(r)))

    def pcg_float(self, r):
        return float(self.lib.wo_pcg_next_float(C.byref(r)))

    def pcg_double(self, r):
        return float(self.lib.wo_pcg_next_double(C.byref(r)))

    def pcg_advance(self, r, delta):
        self.lib.wo_pcg_advance(C.byref(r), delta)

    def sincos_2pi(self, u):
        c, s = C.c_float(), C.c_float()
        self.lib.wo_sincos_2pi(C.c_float(u), C.byref(c), C.byref(s))
        return c.value, s.value

    def logf(self, x):
        return float(self.lib.wo_logf(C.c_float(x)))

    # ---- guided path: network -----------------------------------------------------------------
    def net_n_params(self, cfg):
        self.lib.wo_net_n_params.restype = C.c_uint64
        return int(self.lib.wo_net_n_params(C.byref(cfg)))

    def net3_n_params(self, cfg):
        self.lib.wo_net3_n_params.restype = C.c_uint64
        return int(self.lib.wo_net3_n_params(C.byref(cfg)))

    def net3_forward(self, cfg, params, xyz, want_acts=False):
        p = np.ascontiguousarray(params, dtype=np.float32)
        x = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((len(x), cfg.n_output_padded), dtype=np.float32)
        acts = np.zeros((len(x), cfg.n_levels * cfg.n_features + cfg.n_hidden_layers * cfg.n_neurons), dtype=np.float32) if want_acts else None
        self.lib.wo_net3_forward(C.byref(cfg), _fp(p), _fp(x), len(x), _fp(out), _fp(acts) if want_acts else None)
        return (out, acts) if want_acts else out

    def net3_backward(self, cfg, params, xyz, dl_dout):
        p = np.ascontiguousarray(params, dtype=np.float32)
        x = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dl_dout, dtype=np.float32)
        assert d.shape == (len(x), cfg.n_output_padded)
        g = np.zeros(len(p), dtype=np.float32)
        self.lib.wo_net3_backward(C.byref(cfg), _fp(p), _fp(x), _fp(d), len(x), _fp(g))
        return g

    def net3_optimizer_step(self, cfg, params, state, grad, step, loss_scale):
        assert params.dtype == np.float32 and params.flags.c_contiguous
        g = np.ascontiguousarray(grad, dtype=np.float32)
        inf = np.zeros_like(params)
        self.lib.wo_net3_optimizer_step(C.byref(cfg), _fp(params), _fp(state["m1"]), _fp(state["m2"]), _fp(state["ema_raw"]),
                                        _fp(inf), _fp(g), step, C.c_float(loss_scale),
                                        state["steps"].ctypes.data_as(C.POINTER(C.c_uint32)))
        return inf

    def net_levels(self, cfg):
        res = np.zeros(cfg.n_levels, dtype=np.int32)
        scale = np.zeros(cfg.n_levels, dtype=np.float32)
        enc = self.lib.wo_net_levels(C.byref(cfg), _ip(res), _fp(scale))
        return res, scale, enc

    def net_forward(self, cfg, params, xy, want_acts=False):
        """-> (out[n, n_output_padded], acts[n, enc + hidden*neurons] or None)"""
        p = np.ascontiguousarray(params, dtype=np.float32)
        x = np.ascontiguousarray(xy, dtype=np.float32)
        out = np.zeros((len(x), cfg.n_output_padded), dtype=np.float32)
        acts = None
        if want_acts:
            acts = np.zeros((len(x), cfg.n_levels * cfg.n_features + cfg.n_hidden_layers * cfg.n_neurons), dtype=np.float32)
        self.lib.wo_net_forward(C.byref(cfg), _fp(p), _fp(x), len(x), _fp(out), _fp(acts) if want_acts else None)
        return out, acts

    def net_backward(self, cfg, params, xy, dl_dout):
        p = np.ascontiguousarray(params, dtype=np.float32)
        x = np.ascontiguousarray(xy, dtype=np.float32)
        d = np.ascontiguousarray(dl_dout, dtype=np.float32)
        assert d.shape == (len(x), cfg.n_output_padded)
        g = np.zeros(len(p), dtype=np.float32)
        self.lib.wo_net_backward(C.byref(cfg), _fp(p), _fp(x), _fp(d), len(x), _fp(g))
        return g

    def net_optimizer_state(self, cfg):
        n = self.net_n_params(cfg)
        st = {k: np.zeros(n, dtype=np.float32) for k in ("m1", "m2", "ema_raw")}
        st["steps"] = np.zeros(n, dtype=np.uint32)      # per-parameter Adam step counters (tiny-cuda-nn adam_step)
        return st

    def net_optimizer_step(self, cfg, params, state, grad, step, loss_scale):
        """updates params and state in place, returns the inference (EMA) parameters"""
        assert params.dtype == np.float32 and params.flags.c_contiguous
        g = np.ascontiguousarray(grad, dtype=np.float32)
        inf = np.zeros_like(params)
        self.lib.wo_net_optimizer_step(C.byref(cfg), _fp(params), _fp(state["m1"]), _fp(state["m2"]), _fp(state["ema_raw"]),
                                       _fp(inf), _fp(g), step, C.c_float(loss_scale),
                                       state["steps"].ctypes.data_as(C.POINTER(C.c_uint32)))
        return inf

    # ---- guided path: von Mises / mixture --------------------------------------------------
    def eval_poly_large0(self, y):
        self.lib.wo_eval_poly_large0.restype = C.c_float
        return float(self.lib.wo_eval_poly_large0(C.c_float(y)))

    def vonmises_eval(self, kappa, cos_theta):
        k = np.ascontiguousarray(kappa, dtype=np.float32)
        c = np.ascontiguousarray(cos_theta, dtype=np.float32)
        n = len(k)
        out = [np.zeros(n, dtype=np.float32) for _ in range(4)]
        self.lib.wo_vonmises_eval(_fp(k), _fp(c), n, *[_fp(o) for o in out])
        return dict(zip(("log_i0", "log_i1", "log_pdf", "dlog_dkappa"), out))

    def vonmises_sample(self, kappa, seed, per_point=1):
        k = np.ascontiguousarray(kappa, dtype=np.float32)
        s = np.ascontiguousarray(seed, dtype=np.uint64)
        th = np.zeros(len(k) * per_point, dtype=np.float32)
        self.lib.wo_vonmises_sample(_fp(k), s.ctypes.data_as(C.POINTER(C.c_uint64)), len(k), per_point, _fp(th))
        return th.reshape(len(k), per_point)

    def vmm_loss_gradients(self, raw33, dirs, li, dir_pdf, on_neumann, normal, loss_scale=128.0):
        r = np.ascontiguousarray(raw33, dtype=np.float32)
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        l = np.ascontiguousarray(li, dtype=np.float32)
        p = np.ascontiguousarray(dir_pdf, dtype=np.float32)
        o = np.ascontiguousarray(on_neumann, dtype=np.uint8)
        nn = np.ascontiguousarray(normal, dtype=np.float32)
        n = len(l)
        g = np.zeros((n, 33), dtype=np.float32)
        lk = np.zeros(n, dtype=np.float32)
        self.lib.wo_vmm_loss_gradients(_fp(r), _fp(d), _fp(l), _fp(p), o.ctypes.data_as(C.POINTER(C.c_ubyte)), _fp(nn), n,
                                       C.c_float(loss_scale), _fp(g), _fp(lk))
        return g, lk

    def vmm_pdf_sample(self, raw, wi, seed):
        r = np.ascontiguousarray(raw, dtype=np.float32)
        w = np.ascontiguousarray(wi, dtype=np.float32)
        s = np.ascontiguousarray(seed, dtype=np.uint64)
        n = len(w)
        pdf = np.zeros(n, dtype=np.float32)
        d = np.zeros((n, 2), dtype=np.float32)
        self.lib.wo_vmm_pdf_sample(_fp(r), _fp(w), s.ctypes.data_as(C.POINTER(C.c_uint64)), n, _fp(pdf), _fp(d))
        return pdf, d
