"""The segment LBVH built on the device (csrc/wost_build2.hip; the reference builds its trees on the GPU, core/problem.cu:31-37,
48-54) against the host builder kept as its checker (csrc/lbvh_build.cpp): every array of the uploaded tree -- segment records,
the refined (perimeter-weighted, top-down) assignment of the segments to the leaves, oriented child boxes, normal cones, vertex
adjacency, the compact scan copies -- byte for byte: the shipped scenes, closed and open curves, soups, emissive and not,
zero-length segments, repeated vertices, collinear and axis-parallel runs, meshes far from the origin."""
import numpy as np
import pytest

from elaina_amd.integrator import mesh_build_check


def _curve(n, seed, closed=True, scale=1.0, offset=(0.0, 0.0), wiggle=0.3):
    rng = np.random.default_rng(seed)
    t = np.linspace(0.0, 2.0 * np.pi, n, endpoint=False)
    r = scale * (1.0 + wiggle * np.sin(5 * t + rng.uniform(0, 6)) + 0.05 * np.sin(31 * t))
    V = (np.stack([r * np.cos(t), r * np.sin(t)], 1) + np.asarray(offset)).astype(np.float32)
    S = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1).astype(np.int32)
    if not closed:
        S = S[:-max(1, n // 50)]
    return V, np.ascontiguousarray(rng.permutation(S))          # the index order is not the spatial order


def _soup(n, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-scale, scale, (n, 2))
    V = np.concatenate([a, a + rng.normal(0, 0.02 * scale, (n, 2))]).astype(np.float32)
    return V, np.stack([np.arange(n), np.arange(n) + n], 1).astype(np.int32)


def _degenerate():
    """zero-length segments, a segment between a vertex and itself, two segments on the same vertices, three segments at one
    vertex, -0.0 next to 0.0, a long collinear run and an axis-parallel staircase: equal centroids, equal sort keys, exact ties
    between split costs"""
    V, S = _curve(2000, 3)
    V[5] = V[6]
    V[10, 0], V[11, 0] = 0.0, -0.0
    line = np.stack([np.linspace(2.0, 3.0, 301), np.full(301, 0.5)], 1).astype(np.float32)
    stair = np.cumsum(np.tile(np.asarray([[0.01, 0.0], [0.0, 0.01]]), (150, 1)), 0).astype(np.float32) + np.asarray([3.0, 3.0], np.float32)
    base = len(V)
    V = np.concatenate([V, line, stair, line[:50]])              # the last fifty: duplicates of the line's first vertices
    extra = [[0, 0], [7, 7], [20, 21], [20, 21], [21, 20], [30, 31], [30, 32], [30, 33]]
    extra += [[base + k, base + k + 1] for k in range(300)]
    extra += [[base + 301 + k, base + 302 + k] for k in range(299)]
    extra += [[base + 601 + k, base + 602 + k] for k in range(49)]
    return V, np.ascontiguousarray(np.concatenate([S[:900], np.asarray(extra, np.int32), S[900:]]), np.int32), None


def _shipped(name):
    from elaina_amd import Problem
    p = Problem.load_scene(name)
    return p.d_verts, p.d_segs, p.d_colors


def _emissive(n, seed, closed):
    V, S = _curve(n, seed, closed)
    rng = np.random.default_rng(seed + 100)
    return V, S, rng.uniform(0.0, 1.0, (len(V), 6)).astype(np.float32)


CASES = {
    "one_segment": lambda: (np.asarray([[0, 0], [1, 0.5]], np.float32), np.asarray([[0, 1]], np.int32), None),
    "box_4": lambda: (np.asarray([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32), np.asarray([[0, 1], [1, 2], [2, 3], [3, 0]], np.int32), None),
    "curve_5": lambda: _curve(5, 1) + (None,),
    "curve_17_open": lambda: _curve(17, 2, closed=False) + (None,),
    "curve_64": lambda: _curve(64, 3) + (None,),
    "curve_65": lambda: _curve(65, 4) + (None,),
    "curve_300_open": lambda: _curve(300, 5, closed=False) + (None,),
    "curve_3000": lambda: _curve(3000, 6) + (None,),
    "curve_30000_open": lambda: _curve(30000, 7, closed=False, scale=100.0) + (None,),
    "curve_20000_far": lambda: _curve(20000, 8, scale=3.0, offset=(900.0, -600.0)) + (None,),
    "curve_8000_tiny": lambda: _curve(8000, 9, scale=1e-3) + (None,),
    "soup_5000": lambda: _soup(5000, 10) + (None,),
    "soup_70000": lambda: _soup(70000, 11, scale=50.0) + (None,),
    "emissive_600_open": lambda: _emissive(600, 12, False),
    "emissive_3000": lambda: _emissive(3000, 13, True),
    "zero_colors_1000": lambda: _curve(1000, 14) + (np.zeros((1000, 6), np.float32),),
    "degenerate": _degenerate,
    "ladybug": lambda: _shipped("ladybug"),
    "fille": lambda: _shipped("fille"),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_gpu_device_tree_build_equals_the_host_builder(case):
    V, S, colors = CASES[case]()
    diff, compared, host_ms, dev_ms = mesh_build_check(V, S, colors)
    assert all(v == 0 for v in diff.values()), (case, diff)
    assert compared > 0
    print("%s: %d segments, %d bytes compared, host %.2f ms, device %.2f ms" % (case, len(S), compared, host_ms, dev_ms))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_gpu_device_tree_build_on_random_meshes(seed):
    rng = np.random.default_rng(900 + seed)
    n = int(rng.choice([40, 700, 2500, 9000, 40000]))
    scale = 10.0 ** rng.uniform(-2, 3)
    if seed % 2:
        V, S = _soup(n, seed, scale)
    else:
        V, S = _curve(n, seed, closed=rng.uniform() < 0.5, scale=scale, offset=tuple(scale * rng.choice([0.0, 10.0, 300.0]) * rng.uniform(-1, 1, 2)))
    colors = rng.uniform(0, 1, (len(V), 6)).astype(np.float32) if seed % 3 == 0 else None
    diff, compared, _, _ = mesh_build_check(V, S, colors)
    assert all(v == 0 for v in diff.values()), (seed, n, scale, diff)


@pytest.mark.gpu
def test_gpu_device_tree_build_rejects_an_index_out_of_range():
    from elaina_amd.capi import WostError
    V, S = _curve(2000, 1)
    S = S.copy()
    S[1234, 1] = len(V)
    with pytest.raises(WostError, match="out of range"):
        mesh_build_check(V, S)


@pytest.mark.gpu
def test_gpu_create_refuses_an_index_out_of_range_before_any_launch():
    """wost_create builds its trees on the device (no host builder in front of it, unlike wost_mesh_build_check): a 2 000-segment
    mesh with one bad index must come back as an error, twice in one process (the second build finds the first one's buffers in
    the allocator), and a good mesh must still build and solve afterwards"""
    from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
    from elaina_amd.capi import WostError
    V, S = _curve(2000, 1)
    cols = np.random.default_rng(3).uniform(0, 1, (len(V), 6)).astype(np.float32)
    for bad in (len(V), -1):
        for _ in range(2):
            Sb = S.copy()
            Sb[1234, 1] = bad
            with pytest.raises(WostError, match="out of range"):
                UniformIntegrator(Problem(d_verts=V, d_segs=Sb, d_colors=cols, probe=(1.0, 0.0, 0.0, 0.0, 1.0)), UniformIntegratorSettings((16, 16), 2, 8, 1e-2))
    it = UniformIntegrator(Problem(d_verts=V, d_segs=S, d_colors=cols, probe=(1.0, 0.0, 0.0, 0.0, 1.0)), UniformIntegratorSettings((16, 16), 2, 8, 1e-2))
    it.solve()
    assert np.isfinite(it.solution).all()
    it.close()


@pytest.mark.gpu
def test_gpu_device_tree_build_of_the_shipped_scenes_takes_milliseconds():
    """wost_create's tree build of the BASELINE scenes (61 476 / 153 000 segments) -- fastest of five builds each way"""
    for name in ("ladybug", "fille"):
        V, S, colors = _shipped(name)
        diff, compared, host_ms, dev_ms = mesh_build_check(V, S, colors, repeat=5)
        assert all(v == 0 for v in diff.values()), (name, diff)
        print("%s: %d segments, host build %.1f ms, device build %.1f ms" % (name, len(S), host_ms, dev_ms))
        assert dev_ms < 40.0, (name, dev_ms)
