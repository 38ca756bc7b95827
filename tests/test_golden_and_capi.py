"""CPU tests: the oracle reproduces the committed golden vectors; the C-ABI library builds,
loads and exports every symbol include/wost.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLD = os.path.join(ROOT, "tests", "golden", "oracle_golden.npz")


@pytest.mark.parametrize("scene", ["ladybug", "fille"])
def test_oracle_reproduces_golden_cfg1(oracle, scene):
    from elaina_amd import Problem
    g = np.load(GOLD)
    p = Problem.load_scene(scene)
    # a strip of the 128^2 / 16 spp / depth 32 frame (BASELINE.json configs[0]) keeps this fast
    b, e = 40 * 128, 56 * 128
    r = oracle.solve(p.as_dict(), 128, 128, 16, 32, 1.0, pixel_begin=b, pixel_end=e, want_steps=True)
    assert np.array_equal(r["field"], g[scene + "_cfg1_field"][b:e])
    assert np.array_equal(r["steps"], g[scene + "_cfg1_steps"][b:e])


@pytest.mark.parametrize("scene", ["ladybug", "fille"])
def test_oracle_bvh_reproduces_golden_closest_points(oracle, scene):
    from elaina_amd import Problem
    g = np.load(GOLD)
    p = Problem.load_scene(scene)
    idx, dist, uv, side = oracle.closest_point(p.d_verts, p.d_segs, g[scene + "_cp_pts"], mode=1)
    assert np.array_equal(idx, g[scene + "_cp_idx"])
    assert np.array_equal(dist, g[scene + "_cp_dist"])
    assert np.array_equal(uv, g[scene + "_cp_uv"])
    assert np.array_equal(side, g[scene + "_cp_side"].astype(np.int32))


def test_scene_fixture_hashes():
    import hashlib
    from elaina_amd import Problem
    sums = dict(line.split()[:2][::-1] for line in open(os.path.join(ROOT, "data", "scenes", "SHA256SUMS")))
    for scene in ("ladybug", "fille"):
        d = np.load(Problem.scene_path(scene))
        h = hashlib.sha256()
        for k in ("d_verts", "d_segs", "d_colors", "n_verts", "n_segs"):
            h.update(np.ascontiguousarray(d[k]).tobytes())
        assert h.hexdigest() == sums[scene]
    lb = np.load(Problem.scene_path("ladybug"))
    assert lb["d_verts"].shape == (61626, 2) and lb["d_segs"].shape == (61476, 2)
    assert lb["n_segs"].tolist() == [[2, 0], [0, 1], [1, 3], [3, 2]]


def test_library_builds_and_exports_every_declared_symbol():
    from elaina_amd import build, capi
    path = build.build_library()
    assert os.path.exists(path)
    header = open(os.path.join(ROOT, "include", "wost.h")).read()
    declared = set(re.findall(r"\b(wost3?_[a-z_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    lib = C.CDLL(path)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in capi.load().wost_version()


def test_create_fails_loudly_without_a_gpu_or_with_bad_arguments(ladybug):
    import torch
    from elaina_amd import UniformIntegrator, UniformIntegratorSettings, capi
    lib = capi.load()
    assert lib.wost_create(None, None, 0, None) == -1
    assert b"null" in lib.wost_last_error()
    if not torch.cuda.is_available():
        with pytest.raises(capi.WostError, match="no HIP device"):
            UniformIntegrator(ladybug, UniformIntegratorSettings((16, 16), 1, 4, 1.0))


def test_product_never_imports_the_oracle():
    # the oracle is test infrastructure: nothing under elaina_amd/ may reference it
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "elaina_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"oracle[/.]|wost_oracle|from oracle|import oracle", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_guided_and_network_entry_points_reject_bad_arguments(ladybug):
    """every new entry point of the boundary validates its arguments before touching a device,
    and without a GPU the constructors fail loudly (no CPU path)"""
    import torch
    from elaina_amd import capi
    from elaina_amd.guided import (GuidedIntegrator, GuidedIntegratorSettings, GuidingNetwork, default_net_config)
    lib = capi.load()
    assert lib.wost_net_create(0, None, 1, None) == -1
    assert lib.wost_guided_create(None, None, None, 1, 0, None) == -1
    assert lib.wost_net_inference(None, None, 0, None, 1) == -1
    assert lib.wost_net_train_step(None, None, None, 0, 1.0, 1) == -1
    assert lib.wost_guided_solve(None, None, None) == -1
    assert lib.wost_guided_solve_sharded(None, 0, 1, None, None) == -1
    assert lib.wost_guided_train_set(None, 0, None, None, None, None, None, None, None) == -1
    assert lib.wost_guided_query_network(None, None, 0, None) == -1
    # shape restrictions are reported before any device work
    cfg = default_net_config()
    cfg.n_neurons = 100
    h = C.c_void_p()
    assert lib.wost_net_create(0, C.byref(cfg), 1, C.byref(h)) == -3 and b"network shape" in lib.wost_last_error()
    if not torch.cuda.is_available():
        with pytest.raises(capi.WostError, match="no HIP device"):
            GuidingNetwork()
        with pytest.raises(capi.WostError, match="no HIP device"):
            GuidedIntegrator(ladybug, GuidedIntegratorSettings(frameSize=(8, 8), samplesPerPixel=1), ((0, 0), (1, 1)))
        # the triangle trees are built on the device: no host fallback behind wost3_create, and the build check says so too
        import numpy as np
        from elaina_amd.integrator3d import mesh_build_check
        with pytest.raises(capi.WostError, match="no HIP device"):
            mesh_build_check(np.asarray([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.asarray([[0, 1, 2]], np.int32))
    assert lib.wost3_mesh_build_check(None, 0, 1, None, None, None) == -1
    # settings the device code does not cover are refused up front
    bad = GuidedIntegratorSettings(frameSize=(8, 8), samplesPerPixel=1, maxTrainDepth=7)
    with pytest.raises(capi.WostError, match="max_train_depth|no HIP device"):
        GuidedIntegrator(ladybug, bad, ((0, 0), (1, 1)))
    with pytest.raises(capi.WostError, match="bad guided settings|no HIP device"):
        GuidedIntegrator(ladybug, GuidedIntegratorSettings(frameSize=(8, 8), samplesPerPixel=1), ((1, 1), (0, 0)))
