"""The triangle LBVH built on the device (csrc/wost_build3.hip; the reference builds its trees on the GPU, core/problem.cu:31-37,
48-54) against the host builder kept as its checker: every array of the uploaded mesh -- triangle and edge records, slots in
Morton order, child boxes, normal cones, the silhouette test's operands per slot, the emissive sampler's run boxes -- byte for
byte, on closed and open meshes, emissive and not, degenerate triangles and repeated vertices included."""
import numpy as np
import pytest

import bench
from elaina_amd.integrator3d import mesh_build_check


def _bumpy(level, seed, open_=0, offset=(0.0, 0.0, 0.0)):
    rng = np.random.default_rng(seed)
    V, T = bench.icosphere(level, 1.0)
    V = V.astype(np.float64)
    V *= 1.0 + 0.15 * np.sin(3 * V[:, :1] + rng.uniform(0, 6)) * np.cos(4 * V[:, 1:2] + rng.uniform(0, 6))
    V = (V + np.asarray(offset)).astype(np.float32)
    T = rng.permutation(T)                                   # the index order is not the spatial order
    if open_:
        T = np.ascontiguousarray(T[:-open_])
    return V, np.ascontiguousarray(T, np.int32)


CASES = {
    "one_triangle": lambda: (np.asarray([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.asarray([[0, 1, 2]], np.int32), None),
    "tetrahedron": lambda: (np.asarray([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32),
                            np.asarray([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int32), None),
    "sphere_320": lambda: _bumpy(2, 1) + (None,),
    "sphere_5120_open": lambda: _bumpy(4, 2, open_=37) + (None,),
    "sphere_20480_far": lambda: _bumpy(5, 3, offset=(300.0, -200.0, 250.0)) + (None,),
    "sphere_81920": lambda: _bumpy(6, 4) + (None,),
}


def _emissive(level, seed, open_):
    V, T = _bumpy(level, seed, open_)
    rng = np.random.default_rng(seed + 100)
    return V, T, rng.uniform(0.0, 1.0, (len(V), 6)).astype(np.float32)


CASES["emissive_80"] = lambda: _emissive(1, 5, 0)           # below the flat limit: no run boxes
CASES["emissive_5120_open"] = lambda: _emissive(4, 6, 11)
CASES["emissive_20480"] = lambda: _emissive(5, 7, 0)
CASES["zero_colors_1280"] = lambda: _bumpy(3, 8) + (np.zeros((642, 6), np.float32),)


def _degenerate():
    """zero-area triangles, a triangle with a repeated vertex (a side between a vertex and itself), two triangles on the same
    three vertices, an edge shared by three triangles, -0.0 next to 0.0"""
    V, T = _bumpy(3, 9)
    V[5] = V[6]                                              # two vertices at one point
    V[10, 0], V[11, 0] = 0.0, -0.0
    extra = np.asarray([[0, 1, 1], [2, 2, 2], [3, 4, 5], [5, 4, 3], [3, 4, 7], [3, 4, 9], [12, 13, 12]], np.int32)
    return V, np.ascontiguousarray(np.concatenate([T[:600], extra, T[600:]]), np.int32), None


CASES["degenerate"] = _degenerate


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_gpu_device_mesh_build_equals_the_host_builder(case):
    V, T, colors = CASES[case]()
    diff, compared, host_ms, dev_ms = mesh_build_check(V, T, colors)
    assert all(v == 0 for v in diff.values()), (case, diff)
    assert compared > 0
    print("%s: %d triangles, %d bytes compared, host %.2f ms, device %.2f ms" % (case, len(T), compared, host_ms, dev_ms))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_device_mesh_build_on_random_soups(seed):
    """triangle soups: random vertices, random index triples (edges shared by many triangles, most edges open, repeated vertices)"""
    rng = np.random.default_rng(seed)
    nv, nt = int(rng.integers(3, 400)), int(rng.integers(1, 3000))
    V = (rng.normal(size=(nv, 3)) * rng.choice([1e-3, 1.0, 1e3]) + rng.choice([0.0, 50.0])).astype(np.float32)
    T = rng.integers(0, nv, (nt, 3)).astype(np.int32)
    colors = rng.uniform(0, 1, (nv, 6)).astype(np.float32) if seed % 3 == 0 else None
    diff, compared, _, _ = mesh_build_check(V, T, colors)
    assert all(v == 0 for v in diff.values()), (seed, diff)


@pytest.mark.gpu
def test_gpu_device_mesh_build_rejects_an_index_out_of_range():
    from elaina_amd.capi import WostError
    V, T = _bumpy(2, 1)
    T = T.copy()
    T[17, 1] = len(V)
    with pytest.raises(WostError, match="out of range"):
        mesh_build_check(V, T)
    T[17, 1] = -1
    with pytest.raises(WostError, match="out of range"):
        mesh_build_check(V, T)


@pytest.mark.gpu
def test_gpu_device_mesh_build_of_81920_triangles_takes_milliseconds():
    """an 82 k-triangle mesh in under 10 ms (the host builder: two orders of magnitude more)"""
    V, T = _bumpy(6, 4)
    diff, _, host_ms, dev_ms = mesh_build_check(V, T, None, repeat=5)
    assert all(v == 0 for v in diff.values()), diff
    print("81 920 triangles: host %.1f ms, device %.2f ms" % (host_ms, dev_ms))
    assert dev_ms < 10.0, (host_ms, dev_ms)
