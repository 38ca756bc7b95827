"""ctypes binding of the C-ABI in include/wost.h (libwost_hip.so).

This is the Python-side stub a maintainer would write against the boundary; it adds
nothing to the data path.  If the library is missing it raises -- there is no CPU
fallback in the product.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

WOST_OK = 0
MESH_DIRICHLET = 0
MESH_NEUMANN = 1


class MeshDesc(C.Structure):
    _fields_ = [
        ("n_verts", C.c_int32),
        ("n_segs", C.c_int32),
        ("verts", C.POINTER(C.c_float)),
        ("segs", C.POINTER(C.c_int32)),
        ("colors", C.POINTER(C.c_float)),
    ]


class SourceDesc(C.Structure):
    _fields_ = [
        ("nx", C.c_int32), ("ny", C.c_int32), ("rgb", C.POINTER(C.c_float)),
        ("index_scale", C.c_float * 2), ("index_offset", C.c_float * 2), ("intensity", C.c_float),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("dirichlet", MeshDesc),
        ("neumann", MeshDesc),
        ("dirichlet_intensity", C.c_float),
        ("neumann_intensity", C.c_float),
        ("probe_scale", C.c_float),
        ("probe_pos", C.c_float * 2),
        ("probe_up", C.c_float * 2),
        ("mask", C.POINTER(C.c_uint8)),
        ("source", SourceDesc),
    ]


class Settings(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("spp", C.c_int32),
        ("max_depth", C.c_int32),
        ("eps_shell", C.c_float),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("walk_steps", C.c_uint64),
        ("walks_started", C.c_uint64),
        ("walks_absorbed", C.c_uint64),
        ("walks_truncated", C.c_uint64),
        ("neumann_hits", C.c_uint64),
        ("inner_visits", C.c_uint64),
        ("leaf_visits", C.c_uint64),
        ("trav_trips", C.c_uint64),
        ("step_trips", C.c_uint64),
        ("solve_ms", C.c_double),
        ("kernel_ms", C.c_double),
        ("kernel_launches", C.c_uint32),
        ("reserved", C.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Mesh3Desc(C.Structure):
    """wost3_mesh_desc (include/wost.h)"""
    _fields_ = [("n_verts", C.c_int32), ("n_tris", C.c_int32), ("verts", C.POINTER(C.c_float)), ("tris", C.POINTER(C.c_int32)),
                ("colors", C.POINTER(C.c_float))]


class Source3Desc(C.Structure):
    """wost3_source_desc (include/wost.h)"""
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("rgb", C.POINTER(C.c_float)),
                ("index_scale", C.c_float * 3), ("index_offset", C.c_float * 3), ("intensity", C.c_float)]


class Scene3Desc(C.Structure):
    """wost3_scene_desc (include/wost.h)"""
    _fields_ = [("dirichlet", Mesh3Desc), ("neumann", Mesh3Desc), ("dirichlet_intensity", C.c_float),
                ("neumann_intensity", C.c_float), ("probe_scale", C.c_float), ("probe_pos", C.c_float * 3),
                ("probe_up", C.c_float * 3), ("probe_right", C.c_float * 3), ("mask", C.POINTER(C.c_uint8)),
                ("source", Source3Desc)]


class NetConfig(C.Structure):
    """wost_net_config (include/wost.h); defaults = reference data/ladybug/n.json:49-81."""
    _fields_ = [
        ("n_levels", C.c_int32), ("n_features_per_level", C.c_int32), ("base_resolution", C.c_int32),
        ("per_level_scale", C.c_float),
        ("n_neurons", C.c_int32), ("n_hidden_layers", C.c_int32), ("n_output", C.c_int32),
        ("learning_rate", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float),
        ("l2_reg", C.c_float), ("ema_decay", C.c_float),
    ]


class GuidedSettings(C.Structure):
    """wost_guided_settings (include/wost.h)"""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
        ("eps_shell", C.c_float), ("train_spp_count", C.c_int32),
        ("uniform_fraction_training", C.c_float), ("uniform_fraction_guiding", C.c_float),
        ("max_guided_depth_training", C.c_int32), ("max_guided_depth_guiding", C.c_int32),
        ("aabb_min", C.c_float * 2), ("aabb_max", C.c_float * 2),
        ("max_train_depth", C.c_int32), ("batch_size", C.c_int32), ("min_batch_size", C.c_int32),
        ("batches_per_spp", C.c_int32), ("train_pixel_stride", C.c_int32), ("train_pixel_offset", C.c_int32),
        ("loss_scale", C.c_float),
    ]


class Guided3Settings(C.Structure):
    """wost3_guided_settings (include/wost.h)"""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
        ("eps_shell", C.c_float), ("train_spp_count", C.c_int32),
        ("uniform_fraction_training", C.c_float), ("uniform_fraction_guiding", C.c_float),
        ("max_guided_depth_training", C.c_int32), ("max_guided_depth_guiding", C.c_int32),
        ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3),
        ("max_train_depth", C.c_int32), ("batch_size", C.c_int32), ("min_batch_size", C.c_int32),
        ("batches_per_spp", C.c_int32), ("train_pixel_stride", C.c_int32), ("train_pixel_offset", C.c_int32),
        ("loss_scale", C.c_float),
    ]


class GuidedStats(C.Structure):
    _fields_ = [
        ("walk_steps", C.c_uint64), ("walks_started", C.c_uint64), ("walks_absorbed", C.c_uint64),
        ("walks_truncated", C.c_uint64), ("neumann_hits", C.c_uint64), ("guided_steps", C.c_uint64),
        ("train_samples", C.c_uint64), ("optimizer_steps", C.c_uint64),
        ("solve_ms", C.c_double), ("train_ms", C.c_double), ("kernel_launches", C.c_uint32), ("reserved", C.c_uint32),
        ("net_points", C.c_uint64), ("net_infer_ms", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class LaunchInfo(C.Structure):
    """wost_launch_info (include/wost.h)"""
    _fields_ = [("kind", C.c_int32), ("walkers", C.c_uint32), ("walkers_beside", C.c_uint32), ("grid", C.c_uint32),
                ("ms", C.c_double), ("walk_steps_done", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


LAUNCH_ROUND, LAUNCH_QUAD, LAUNCH_ONE, LAUNCH_PERSISTENT, LAUNCH_WAIT = 0, 1, 2, 3, 4
LAUNCH_NAMES = {0: "round", 1: "quad round", 2: "one launch", 3: "persistent", 4: "wait"}


# wost_sync_fn (include/wost.h): int (*)(void *user, int op, void *data, uint64_t count)
SYNC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64)
SYNC_SUM_I64_DEVICE, SYNC_MIN_I64_HOST, SYNC_RANKS_I64_HOST = 0, 1, 2
SYNC_UNSUPPORTED = 2       # callback return value for an unknown op
# wost_frame_fn: int (*)(void *user, int reason, int32_t sample_id, double elapsed_ms, const float *field_rgb)
FRAME_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int32, C.c_double, C.POINTER(C.c_float))

EXPORTS = [
    "wost_create", "wost_solve", "wost_solve_sharded", "wost_render_sdf", "wost_render_source", "wost_closest_point",
    "wost_closest_silhouette", "wost_ray_intersect", "wost_set_option", "wost_last_launches", "wost_destroy",
    "wost_vonmises_eval", "wost_vonmises_sample", "wost_vmm_pdf_sample", "wost_vmm_loss_gradients",
    "wost_net_create", "wost_net_destroy", "wost_net_n_params", "wost_net_get_params",
    "wost_net_set_params", "wost_net_set_gradient_buffer", "wost_net_inference", "wost_net_train_step", "wost_net_set_option",
    "wost_guided_create", "wost_guided_set_sync", "wost_guided_set_frame_callback", "wost_guided_network", "wost_guided_scene", "wost_guided_query_network", "wost_guided_solve", "wost_guided_solve_sharded", "wost_guided_train_set", "wost_guided_set_option", "wost_guided_destroy",
    "wost3_create", "wost3_solve", "wost3_solve_sharded", "wost3_closest_point", "wost3_closest_silhouette", "wost3_ray_intersect", "wost3_mesh_build_check",
    "wost3_render_sdf", "wost3_render_source", "wost3_destroy", "wost3_vmf_eval", "wost3_vmf_sample", "wost3_vmm_pdf_sample", "wost3_vmm_loss_gradients",
    "wost3_net_create", "wost3_guided_create", "wost3_guided_destroy", "wost3_guided_network", "wost3_guided_solve", "wost3_guided_solve_sharded",
    "wost3_guided_query_network", "wost3_guided_train_set", "wost3_guided_scene",
    "wost_last_error", "wost_version", "wost_mesh_build_check",
]

_lib = None


def library_path():
    # WOST_LIB: developer override to A/B alternative builds of the same C-ABI
    return os.environ.get("WOST_LIB", _build.LIB_PATH)


def load():
    """Load libwost_hip.so (built in-tree by elaina_amd.build). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7; loading it FIRST makes this library bind to the same
    # HIP runtime (same SONAME), so one process never holds two runtimes (DESIGN.md "runtime").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "libwost_hip.so is missing (%s): run `python -m elaina_amd.build`; "
            "this package has no CPU fallback" % path)
    L = C.CDLL(path)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.wost_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(Settings), C.c_int, C.POINTER(C.c_void_p)]
    L.wost_solve.argtypes = [C.c_void_p, C.c_int32, C.c_int32, fp, C.POINTER(Stats)]
    L.wost_solve_sharded.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.wost_render_sdf.argtypes = [C.c_void_p, C.c_int, fp]
    L.wost_render_source.argtypes = [C.c_void_p, fp]
    L.wost_closest_point.argtypes = [C.c_void_p, C.c_int, fp, C.c_int32, ip, fp, fp, ip]
    L.wost_closest_silhouette.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_int32, fp]
    L.wost_ray_intersect.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, C.c_int32, ip, fp, ip]
    L.wost_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.wost_last_launches.argtypes = [C.c_void_p, C.POINTER(LaunchInfo), C.c_int32, C.POINTER(C.c_int32)]
    u64p = C.POINTER(C.c_uint64)
    L.wost_vonmises_eval.argtypes = [C.c_int, fp, fp, C.c_int32, fp, fp, fp, fp]
    L.wost_vonmises_sample.argtypes = [C.c_int, fp, u64p, C.c_int32, C.c_int32, fp]
    L.wost_vmm_pdf_sample.argtypes = [C.c_int, fp, fp, u64p, C.c_int32, fp, fp]
    L.wost_vmm_loss_gradients.argtypes = [C.c_int, fp, fp, fp, fp, C.POINTER(C.c_uint8), fp, C.c_int32, C.c_float, fp, fp]
    L.wost_net_create.argtypes = [C.c_int, C.POINTER(NetConfig), C.c_uint64, C.POINTER(C.c_void_p)]
    L.wost_net_destroy.argtypes = [C.c_void_p]
    L.wost_net_n_params.argtypes = [C.c_void_p, u64p, u64p]
    L.wost_net_get_params.argtypes = [C.c_void_p, C.c_int, fp]
    L.wost_net_set_params.argtypes = [C.c_void_p, fp]
    L.wost_net_inference.argtypes = [C.c_void_p, fp, C.c_int32, fp, C.c_int]
    L.wost_net_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.wost_net_train_step.argtypes = [C.c_void_p, fp, fp, C.c_int32, C.c_float, C.c_int]
    L.wost_guided_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(GuidedSettings), C.POINTER(NetConfig), C.c_uint64,
                                     C.c_int, C.POINTER(C.c_void_p)]
    L.wost_guided_network.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.wost_guided_set_sync.argtypes = [C.c_void_p, SYNC_FN, C.c_void_p]
    L.wost_guided_set_frame_callback.argtypes = [C.c_void_p, FRAME_FN, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.wost_net_set_gradient_buffer.argtypes = [C.c_void_p, C.c_void_p]
    L.wost_guided_scene.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.wost_guided_query_network.argtypes = [C.c_void_p, fp, C.c_int32, fp]
    L.wost_guided_solve.argtypes = [C.c_void_p, fp, C.POINTER(GuidedStats)]
    L.wost_guided_solve_sharded.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(GuidedStats)]
    L.wost_guided_train_set.argtypes = [C.c_void_p, C.c_int32, ip, fp, fp, fp, fp, fp, C.POINTER(C.c_uint8)]
    L.wost_guided_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.wost_guided_destroy.argtypes = [C.c_void_p]
    L.wost_destroy.argtypes = [C.c_void_p]
    L.wost3_create.argtypes = [C.POINTER(Scene3Desc), C.POINTER(Settings), C.c_int, C.POINTER(C.c_void_p)]
    L.wost3_solve.argtypes = [C.c_void_p, C.c_int32, C.c_int32, fp, C.POINTER(Stats)]
    L.wost3_solve_sharded.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.wost3_closest_point.argtypes = [C.c_void_p, C.c_int, fp, C.c_int32, ip, fp, fp, ip]
    L.wost3_closest_silhouette.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_int32, fp]
    L.wost3_ray_intersect.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, C.c_int32, ip, fp, ip]
    L.wost3_mesh_build_check.argtypes = [C.POINTER(Mesh3Desc), C.c_int, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.wost_mesh_build_check.argtypes = [C.POINTER(MeshDesc), C.c_int, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.wost3_render_sdf.argtypes = [C.c_void_p, C.c_int, fp]
    L.wost3_render_source.argtypes = [C.c_void_p, fp]
    L.wost3_destroy.argtypes = [C.c_void_p]
    L.wost3_vmf_eval.argtypes = [C.c_int, fp, fp, C.c_int32, fp]
    L.wost3_vmf_sample.argtypes = [C.c_int, fp, fp, u64p, C.c_int32, C.c_int32, fp]
    L.wost3_vmm_pdf_sample.argtypes = [C.c_int, fp, fp, u64p, C.c_int32, fp, fp]
    L.wost3_vmm_loss_gradients.argtypes = [C.c_int, fp, fp, fp, fp, C.POINTER(C.c_ubyte), fp, C.c_int32, C.c_float, fp, fp]
    L.wost3_net_create.argtypes = [C.c_int, C.POINTER(NetConfig), C.c_uint64, C.POINTER(C.c_void_p)]
    L.wost3_guided_create.argtypes = [C.POINTER(Scene3Desc), C.POINTER(Guided3Settings), C.POINTER(NetConfig), C.c_uint64, C.c_int,
                                      C.POINTER(C.c_void_p)]
    L.wost3_guided_destroy.argtypes = [C.c_void_p]
    L.wost3_guided_network.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.wost3_guided_scene.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.wost3_guided_solve.argtypes = [C.c_void_p, fp, C.POINTER(GuidedStats)]
    L.wost3_guided_solve_sharded.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(GuidedStats)]
    L.wost3_guided_query_network.argtypes = [C.c_void_p, fp, C.c_int32, fp]
    L.wost3_guided_train_set.argtypes = [C.c_void_p, C.c_int32, ip, fp, fp, fp, fp, fp, C.POINTER(C.c_uint8)]
    L.wost_last_error.restype = C.c_char_p
    L.wost_version.restype = C.c_char_p
    for name in EXPORTS:
        getattr(L, name)  # fail loudly on a missing symbol
    _lib = L
    return L


class WostError(RuntimeError):
    pass


def _check(rc, what):
    if rc != WOST_OK:
        raise WostError("%s failed (%d): %s" % (what, rc, load().wost_last_error().decode()))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))
