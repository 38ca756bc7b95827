"""Tree-sized boundary meshes FAR from the origin (the class of scene the ray / box rounding bug of round 3, c5eaca1, came from):

* tools/fuzz/fuzz_far_trees.py -- Neumann meshes of 5 000 .. 30 000 primitives, closed and open, emissive and not, the whole
  scene 10 .. 300 scene sizes away -- through the guided 2-D, the uniform 3-D and the guided 3-D kernels, 40 seeds each with a
  frozen network and 10 seeds each with trainSppCount >= 2 (records, Adam / EMA steps, the trained network's walks), bit for bit
  against the oracle; the guided 3-D kernels once more with the reference's EIGHT-level network (guided3d_l8: 24 seeds frozen,
  guided3d_l8_train: 8 seeds training) -- the matrix-core network kernels, the grid gradient through spatial boxes and
  g3_fused_kernel (a sample in one launch), which the four-level seeds do not reach.  The oracle half of every seed (fields, counters, final parameters) is the committed fixture
  tests/golden/far_trees_<mode>.npz, written in the build container by `fuzz_far_trees.py golden <mode>`; the GPU box regenerates
  the scene from the seed, runs HIP and compares -- none of the oracle's CPU time (10 - 17 s a guided seed) inside the GPU suite;
* the batch ray queries (ray_kernel / ray3_kernel behind wost_ray_intersect / wost3_ray_intersect) with origins ON the mesh
  at 1 .. 10^4 mesh extents from the origin against BRUTE FORCE: the oracle's ray queries are plain loops over every
  primitive (oracle/wost_oracle.c ray_closest, wost_oracle3d.c ray_closest3), no tree of its own that could share a flaw.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools", "fuzz"))

MODES = ["guided2d", "uniform3d", "guided3d", "guided2d_train", "guided3d_train", "guided3d_l8", "guided3d_l8_train"]


@pytest.mark.parametrize("mode", MODES)
def test_far_tree_fixtures_hold_every_seed(mode):
    """(CPU) the fixture is there, holds the seeds the mode asks for, and -- spot check, seed 0 of the cheap mode -- is what the
    oracle computes today"""
    import fuzz_far_trees as F
    g = np.load(F.golden_path(mode))
    n = int(g["count"])
    assert n == F.DEFAULT_COUNT[mode] and str(g["mode"]) == mode
    for seed in range(n):
        c = F.CASES[mode](seed) if seed < 2 else None
        assert g["field_%d" % seed].dtype == np.float32 and g["counters_%d" % seed].dtype == np.uint64
        if c is not None:
            assert g["field_%d" % seed].shape == (c["w"] * c["h"], 3) and len(g["counters_%d" % seed]) == len(c["keys"])
    if mode == "uniform3d":
        from oracle.oracle import Oracle
        c = F.CASES[mode](0)
        r = F.oracle_run(Oracle(), c)
        assert np.array_equal(r["field"], g["field_0"], equal_nan=True) and np.array_equal(r["counters"], g["counters_0"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_gpu_tree_sized_neumann_meshes_far_from_the_origin(mode):
    import fuzz_far_trees as F
    bad, n = F.check_golden(mode)
    assert n == F.DEFAULT_COUNT[mode]
    assert not bad, bad[:5]


@pytest.mark.gpu
def test_gpu_guided3_forms_agree_on_far_tree_sized_meshes():
    """tools/fuzz/fuzz_g3_forms.py: the same scenes with the reference's eight-level network (the shape g3_fused_kernel and the MFMA
    kernels cover; the oracle-backed seeds above use four levels), frozen and training, frames of 80 .. 51 200 pixels -- one launch per
    sample with spread walkers, with 64 walkers per wave, and the launches per depth agree bit for bit (eight seeds here; 200 once
    under gpurun: profiles/r05_ag_fuzz_g3_forms.txt)"""
    import fuzz_g3_forms as G
    bad = []
    for seed in range(8):
        c = G.case(seed)
        d = G.run_forms(c)
        if d:
            bad.append((seed, d, c["what"]))
    assert not bad, bad[:3]


@pytest.mark.gpu
@pytest.mark.parametrize("extents", [1.0, 100.0, 1e3, 1e4])
def test_gpu_rays_from_points_on_a_fine_mesh_far_from_the_origin_2d(oracle, extents):
    """30 000 segments; the mesh centre `extents` mesh sizes from the origin (at 10^4 one ulp of a coordinate is five segment
    lengths: vertices collapse, segments degenerate -- the answer must still be brute force's); rays start on the segments, a
    hair off them, and at vertices"""
    from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
    rng = np.random.default_rng(int(extents) + 3)
    n_seg, size = 30000, 100.0
    t = np.linspace(0.0, 2.0 * np.pi, n_seg, endpoint=False)
    r = size * (1.0 + 0.2 * np.sin(7 * t) + 0.05 * np.sin(31 * t))
    off = np.asarray([0.8, -0.6]) * 2.0 * size * extents
    V = (np.stack([r * np.cos(t), r * np.sin(t)], 1) + off).astype(np.float32)
    S = np.stack([np.arange(n_seg), (np.arange(n_seg) + 1) % n_seg], 1).astype(np.int32)[:-5]          # open: five segments removed
    td = np.linspace(0.0, 2.0 * np.pi, 64, endpoint=False)
    dv = (np.stack([15.0 * np.cos(td), 15.0 * np.sin(td)], 1) + off).astype(np.float32)
    ds = np.stack([np.arange(64), (np.arange(64) + 1) % 64], 1).astype(np.int32)
    p = Problem(d_verts=dv, d_segs=ds, d_colors=np.ones((64, 6), np.float32), n_verts=V, n_segs=S, n_colors=None,
                probe=(110.0, float(off[0]), float(off[1]), 0.0, 1.0))
    it = UniformIntegrator(p, UniformIntegratorSettings((8, 8), 1, 4, 1.0))
    n = 50000
    si = rng.integers(0, len(S), n)
    a, b = V[S[si, 0]], V[S[si, 1]]
    on = (a + (b - a) * rng.choice([0.0, 1.0, 0.5, 0.25, 0.9], n)[:, None].astype(np.float32)).astype(np.float32)
    e = (b - a).astype(np.float64)
    nrm = np.stack([e[:, 1], -e[:, 0]], 1)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    pts = (on + (rng.choice([0.0, 0.0, 0.0, 0.05, -0.05, 1e-3, -1e-3], n)[:, None] * nrm)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, size=n)
    d = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)
    tmax = (rng.uniform(0.5, 1.0, n) * rng.choice([0.03, 0.5, 10.0, 300.0], n)).astype(np.float32)
    gh, gt, gi = it.ray_intersect(pts, d, tmax)
    rh, rt, ri = oracle.ray_intersect(V, S, pts, d, tmax)
    it.close()
    # (at 10^4 extents the collapsed mesh is a lump of degenerate segments around every origin: nearly every ray hits something)
    assert np.array_equal(gh, rh) and 0.05 < rh.mean() <= 1.0, (int((gh != rh).sum()), float(rh.mean()))
    hit = rh == 1
    assert np.array_equal(gt[hit], rt[hit]) and np.array_equal(gi[hit], ri[hit])


@pytest.mark.gpu
@pytest.mark.parametrize("extents", [1.0, 100.0, 1e3, 1e4])
def test_gpu_rays_from_points_on_a_fine_mesh_far_from_the_origin_3d(oracle, extents):
    """a bumpy sphere of 20 480 triangles, its centre `extents` diameters from the origin; rays start on triangles (random
    barycentric points), at vertices and a hair off the surface"""
    import bench
    from elaina_amd import UniformIntegratorSettings
    from elaina_amd.integrator3d import Problem3, UniformIntegrator3
    rng = np.random.default_rng(int(extents) + 11)
    V0, T = bench.icosphere(5, 1.0)
    V0 = V0.astype(np.float64)
    V0 *= 1.0 + 0.1 * np.sin(3 * V0[:, :1] + 0.7) * np.cos(4 * V0[:, 1:2])
    off = np.asarray([0.6, -0.5, 0.62]) * 2.0 * extents
    V = (V0 + off).astype(np.float32)
    T = np.ascontiguousarray(T[:-300])                                     # open: boundary edges
    dV, dT = bench.icosphere(1, 0.3)
    sd = {"d_verts": (dV.astype(np.float64) + off).astype(np.float32), "d_tris": dT, "d_colors": np.ones((len(dV), 6), np.float32),
          "n_verts": V, "n_tris": T, "n_colors": np.zeros((len(V), 6), np.float32), "dirichlet_intensity": 1.0, "neumann_intensity": 1.0,
          "probe": (1.1, tuple(float(x) for x in off), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0))}
    it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((8, 8), 1, 4, 1e-3))
    n = 20000
    ti = rng.integers(0, len(T), n)
    bary = rng.dirichlet((1.0, 1.0, 1.0), n)
    bary[rng.uniform(size=n) < 0.25] = (1.0, 0.0, 0.0)                     # a quarter of the origins are vertices
    A, B, Cc = V[T[ti, 0]].astype(np.float64), V[T[ti, 1]].astype(np.float64), V[T[ti, 2]].astype(np.float64)
    on = A * bary[:, :1] + B * bary[:, 1:2] + Cc * bary[:, 2:]
    nrm = np.cross(B - A, Cc - A)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    pts = (on + rng.choice([0.0, 0.0, 0.0, 1e-3, -1e-3, 1e-5], n)[:, None] * nrm).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    tmax = (rng.uniform(0.5, 1.0, n) * rng.choice([0.01, 0.3, 5.0], n)).astype(np.float32)
    gh, gt, gi = it.ray_intersect(pts, d, tmax)
    rh, rt, ri = oracle.ray_intersect3(V, T, pts, d, tmax)
    it.close()
    assert np.array_equal(gh, rh) and 0.05 < rh.mean() <= 1.0, (int((gh != rh).sum()), float(rh.mean()))
    hit = rh == 1
    assert np.array_equal(gt[hit], rt[hit]) and np.array_equal(gi[hit], ri[hit])
