"""3-D uniform path on the GPU (wost3_* of include/wost.h) against the CPU oracle
(oracle/wost_oracle3d.c) on the same seeded inputs: bit-exact fields, counters and query results."""
import numpy as np
import pytest

from conftest import cube_scene3, sphere_scene3

pytestmark = pytest.mark.gpu


def _it(sd, w, h, spp, depth, eps):
    from elaina_amd import UniformIntegratorSettings
    from elaina_amd.integrator3d import Problem3, UniformIntegrator3
    return UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((w, h), spp, depth, eps))


def _same_solve(oracle, sd, w, h, spp, depth, eps):
    it = _it(sd, w, h, spp, depth, eps)
    it.solve()
    ref = oracle.solve3(sd, w, h, spp, depth, eps, threads=16)
    for k in ("walk_steps", "walks_started", "walks_absorbed", "walks_truncated", "neumann_hits"):
        assert it.last_stats[k] == ref[k], k
    assert np.array_equal(it.solution, ref["field"]), float(np.abs(it.solution - ref["field"]).max())
    it.close()
    return ref


@pytest.mark.parametrize("subdiv", [0, 2, 3])
def test_closest_point_on_triangles_matches_oracle(oracle, subdiv):
    sd = sphere_scene3(subdiv=subdiv)
    it = _it(sd, 8, 8, 1, 4, 1e-3)
    rng = np.random.default_rng(subdiv)
    pts = rng.uniform(-1.6, 1.6, size=(20000, 3)).astype(np.float32)
    pts[:2000] = sd["d_verts"][rng.integers(0, len(sd["d_verts"]), 2000)]          # exactly on vertices: ties between triangles
    got = it.closest_point(pts)
    ref = oracle.closest_point3(sd["d_verts"], sd["d_tris"], pts)
    for x, y in zip(got, ref):
        assert np.array_equal(x, y)
    it.close()


def test_silhouette_and_ray_queries_match_oracle(oracle):
    sd = cube_scene3(n=2, d_faces=(0, 1), n_faces=(2, 3, 4, 5), value=lambda x, y, z: x)
    it = _it(sd, 8, 8, 1, 4, 1e-3)
    rng = np.random.default_rng(3)
    pts = rng.uniform(-0.5, 1.5, size=(20000, 3)).astype(np.float32)
    V, T = sd["n_verts"], sd["n_tris"]
    assert np.array_equal(it.closest_silhouette(pts), oracle.closest_silhouette3(V, T, pts))
    rmax = rng.uniform(0.05, 1.0, 20000).astype(np.float32)
    assert np.array_equal(it.closest_silhouette(pts, rmax), oracle.closest_silhouette3(V, T, pts, rmax))
    o = rng.uniform(0.05, 0.95, size=(20000, 3)).astype(np.float32)
    d = rng.normal(size=(20000, 3))
    d = (d / np.linalg.norm(d, axis=1)[:, None]).astype(np.float32)
    tmax = rng.uniform(0.1, 3.0, 20000).astype(np.float32)
    got, ref = it.ray_intersect(o, d, tmax), oracle.ray_intersect3(V, T, o, d, tmax)
    assert np.array_equal(got[0], ref[0]) and ref[0].mean() > 0.2
    hit = ref[0] == 1
    assert np.array_equal(got[1][hit], ref[1][hit]) and np.array_equal(got[2][hit], ref[2][hit])
    it.close()


def test_dirichlet_sphere_solve_matches_oracle_and_the_harmonic_solution(oracle):
    sd = sphere_scene3(subdiv=3, value=lambda x, y, z: x * y + z)          # 1280 triangles, harmonic data
    ref = _same_solve(oracle, sd, 24, 20, 24, 128, 2e-3)
    assert ref["walks_absorbed"] > 0.99 * ref["walks_started"]
    sd["probe"] = (0.55, (0.0, 0.0, 0.1), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0))       # the whole slice inside the ball
    it = _it(sd, 32, 32, 512, 256, 2e-3)
    it.solve()
    scale, pos, up, right = sd["probe"]
    ys, xs = np.divmod(np.arange(32 * 32), 32)
    p = scale * ((2.0 * xs / 32 - 1.0)[:, None] * np.asarray(right)[None] + (2.0 * ys / 32 - 1.0)[:, None] * np.asarray(up)[None]) + np.asarray(pos)
    exact = p[:, 0] * p[:, 1] + p[:, 2]
    err = it.solution[:, 0] - exact
    assert abs(float(err.mean())) < 5e-3 and float(np.sqrt((err ** 2).mean())) < 0.03
    it.close()


@pytest.mark.parametrize("case", ["zero_flux", "flux", "ragged_mask"])
def test_mixed_boundary_cube_solves_match_oracle(oracle, case):
    if case == "flux":
        sd = cube_scene3(n=2, d_faces=(0, 1), n_faces=(2, 3, 4, 5), value=lambda x, y, z: z,
                         flux=lambda x, y, z, f: {4: 1.0, 5: -1.0}.get(f, 0.0), weld=False)
        sd["probe"] = (0.35, (0.5, 0.5, 0.5), (0.0, 0.0, 1.0), (1.0, 0.0, 0.0))
        ref = _same_solve(oracle, sd, 16, 12, 12, 96, 2e-3)
    else:
        sd = cube_scene3(n=2, d_faces=(0, 1), n_faces=(2, 3, 4, 5), value=lambda x, y, z: x, flux=lambda x, y, z, f: 0.0)
        w, h = (16, 16) if case == "zero_flux" else (19, 13)
        if case == "ragged_mask":
            sd["mask"] = (np.arange(w * h) % 3 != 0).astype(np.uint8)
        ref = _same_solve(oracle, sd, w, h, 10, 64, 2e-3)
        if case == "ragged_mask":
            assert ref["walks_started"] == int(sd["mask"].sum()) * 10
    assert ref["neumann_hits"] > 0


def test_3d_sharded_solve_sums_to_the_full_field(oracle):
    import torch
    sd = sphere_scene3(subdiv=2, value=lambda x, y, z: x)
    it = _it(sd, 40, 24, 6, 64, 2e-3)
    it.solve()
    full = it.solution.copy()
    acc = torch.zeros(40 * 24 * 3, dtype=torch.float32, device="cuda")
    for r in range(3):
        part = torch.zeros_like(acc)
        it.solve_sharded(r, 3, part.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        acc += part
    assert np.array_equal(acc.cpu().numpy().reshape(-1, 3), full)
    it.close()


def _shell_scene(subdiv_d, subdiv_n, flux=None):
    """a Dirichlet icosphere of radius 0.45 (value x) inside a Neumann icosphere of radius 1 (inward normals for the
    domain between them do not matter: both sides are coloured alike); the probe is a slice of the shell"""
    inner = sphere_scene3(subdiv=subdiv_d, radius=0.45, value=lambda x, y, z: x)
    outer = sphere_scene3(subdiv=subdiv_n, radius=1.0, value=(lambda x, y, z: flux(x, y, z)) if flux else None)
    sd = dict(inner)
    sd["n_verts"], sd["n_tris"], sd["n_colors"] = outer["d_verts"], outer["d_tris"], outer["d_colors"]
    sd["probe"] = (0.7, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0))
    return sd


@pytest.mark.parametrize("subdiv", [2, 3])
def test_3d_neumann_tree_queries_match_oracle(oracle, subdiv):
    """Neumann meshes above 64 triangles answer their silhouette and ray queries through the tree (320 and 1280
    triangles here); the oracle walks every triangle / edge -- same distances, same hits, same triangle on ties"""
    sd = _shell_scene(1, subdiv)
    V, T = sd["n_verts"], sd["n_tris"]
    it = _it(sd, 8, 8, 1, 4, 1e-3)
    rng = np.random.default_rng(subdiv)
    pts = rng.uniform(-1.3, 1.3, size=(6000, 3)).astype(np.float32)
    pts[:500] = V[rng.integers(0, len(V), 500)] * np.float32(0.999)          # next to vertices: many edges at almost the same distance
    assert np.array_equal(it.closest_silhouette(pts), oracle.closest_silhouette3(V, T, pts))
    rmax = rng.uniform(0.02, 0.6, len(pts)).astype(np.float32)
    got, ref = it.closest_silhouette(pts, rmax), oracle.closest_silhouette3(V, T, pts, rmax)
    assert np.array_equal(got, ref) and np.isinf(ref).any() and np.isfinite(ref).any()
    o = rng.uniform(-0.9, 0.9, size=(6000, 3)).astype(np.float32)
    d = rng.normal(size=(6000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:300] = 0.0
    d[np.arange(300), rng.integers(0, 3, 300)] = 1.0                            # axis-parallel rays: zero direction components
    o[300:600] = 0.0                                                            # from the centre through ...
    d[300:600] = V[rng.integers(0, len(V), 300)]                                # ... the vertices: hits shared by up to six triangles
    tmax = rng.uniform(0.3, 3.0, len(o)).astype(np.float32)
    got, ref = it.ray_intersect(o, d, tmax), oracle.ray_intersect3(V, T, o, d, tmax)
    assert np.array_equal(got[0], ref[0]) and ref[0].any() and not ref[0].all()
    hit = ref[0] != 0
    assert np.array_equal(got[1][hit], ref[1][hit]) and np.array_equal(got[2][hit], ref[2][hit])
    it.close()


@pytest.mark.parametrize("case", ["bumpy", "bumpy_with_holes", "flat_patches", "tiny", "huge", "degenerate_and_doubled"])
def test_3d_silhouette_tree_prunes_by_normal_cones_exactly(oracle, case):
    """the tree skips subtrees whose normal cone proves that no edge below can be a silhouette from the query point;
    reflex and convex folds seen from inside and outside, boundary edges (always silhouettes), coplanar neighbours,
    query points on the surface and on edges: the distances of the oracle's loop over all edges, bit for bit"""
    if case == "flat_patches":
        sd = cube_scene3(n=8, d_faces=(0,), n_faces=(1, 2, 3, 4, 5), value=lambda x, y, z: x, flux=lambda x, y, z, f: 0.0)
        V, T = sd["n_verts"], sd["n_tris"]
    else:
        sd = _shell_scene(1, 3)
        V = sd["n_verts"].astype(np.float64)
        V *= (1.0 + 0.22 * np.sin(5.0 * V[:, :1]) * np.sin(4.0 * V[:, 1:2] + 0.3) * np.cos(3.0 * V[:, 2:3]))
        V = V.astype(np.float32)
        T = sd["n_tris"]
        if case == "bumpy_with_holes":
            keep = np.ones(len(T), bool)
            keep[np.random.default_rng(5).choice(len(T), 40, replace=False)] = False
            T = np.ascontiguousarray(T[keep])
        if case == "degenerate_and_doubled":
            # zero-area triangles (no normal), triangles listed twice (edges with four incident triangles: the first two count)
            T = np.concatenate([T, T[:60], np.stack([T[100:160, 0], T[100:160, 0], T[100:160, 1]], axis=1)]).astype(T.dtype)
        sd["n_verts"], sd["n_tris"] = V, T
    # the silhouette test carries ABSOLUTE thresholds (1e-3): a scene of that size is all "near" cases, a huge one none
    scale = np.float32({"tiny": 2e-3, "huge": 3e3}.get(case, 1.0))
    if scale != 1.0:
        V = (V * scale).astype(np.float32)
        sd["n_verts"] = V
        sd["d_verts"] = (sd["d_verts"] * scale).astype(np.float32)
    it = _it(sd, 8, 8, 1, 4, 1e-3)
    rng = np.random.default_rng(11)
    pts = (rng.uniform(-1.5, 1.5, size=(8000, 3)) * scale).astype(np.float32)
    pts[:1000] = V[rng.integers(0, len(V), 1000)] * rng.uniform(0.97, 1.03, (1000, 1)).astype(np.float32)
    pts[1000:1300] = V[rng.integers(0, len(V), 300)]                                                     # on vertices
    tri = T[rng.integers(0, len(T), 300)]
    pts[1300:1600] = (0.5 * (V[tri[:, 0]].astype(np.float64) + V[tri[:, 1]])).astype(np.float32)       # on edges
    pts[1600:1900] = (V[tri].astype(np.float64).mean(axis=1)).astype(np.float32)                         # on faces
    pts[1900:2200] *= np.float32(40.0)                                                                   # far away
    got, ref = it.closest_silhouette(pts), oracle.closest_silhouette3(V, T, pts)
    assert np.array_equal(got, ref) and np.isfinite(ref).any()
    rmax = (rng.uniform(0.02, 0.8, len(pts)) * scale).astype(np.float32)
    got, ref = it.closest_silhouette(pts, rmax), oracle.closest_silhouette3(V, T, pts, rmax)
    assert np.array_equal(got, ref) and np.isinf(ref).any() and np.isfinite(ref).any()
    it.close()


@pytest.mark.parametrize("case", ["zero_flux_shell", "emissive_shell", "cube_walls"])
def test_3d_neumann_mesh_of_hundreds_of_triangles(oracle, case):
    """whole solves with the Neumann side on the tree: bit-exact against the oracle"""
    if case == "cube_walls":
        sd = cube_scene3(n=6, d_faces=(0, 1), n_faces=(2, 3, 4, 5), value=lambda x, y, z: x, flux=lambda x, y, z, f: 0.0)
        assert len(sd["n_tris"]) == 288
        ref = _same_solve(oracle, sd, 12, 12, 8, 48, 2e-3)
    else:
        sd = _shell_scene(2, 3, flux=(lambda x, y, z: 0.3 * y) if case == "emissive_shell" else None)
        assert len(sd["n_tris"]) == 1280
        ref = _same_solve(oracle, sd, 14, 12, 6, 64, 2e-3)
    assert ref["neumann_hits"] > 0


@pytest.mark.parametrize("case", ["zero_flux", "emissive"])
def test_3d_source_term_inside_a_tree_sized_neumann_shell(oracle, case):
    """the source sample's line to the boundary and the boundary sample's shadow ray, both answered by the wave through its task
    pools on a 1280-triangle shell (walk3_kernel<EMISSIVE, SOURCE, NTREE>, part B in three stages): the oracle's field and counters"""
    sd = _shell_scene(2, 3, flux=(lambda x, y, z: 0.3 * y) if case == "emissive" else None)
    rng = np.random.default_rng(9)
    sd["source"] = {"rgb": rng.uniform(0, 2, (6, 5, 4, 3)).astype(np.float32), "index_scale": (2.0, 2.5, 3.0),
                    "index_offset": (2.0, 2.5, 3.0), "intensity": 0.8}
    ref = _same_solve(oracle, sd, 14, 12, 6, 64, 2e-3)
    assert ref["neumann_hits"] > 0 and np.any(ref["field"] != 0)


@pytest.mark.parametrize("case", ["ball", "cube_with_reflecting_walls", "emissive_walls_and_mask"])
def test_3d_source_term_matches_oracle(oracle, case):
    """sampleSource in 3-D (dense grid, trilinear; HarmonicGreenBall<3>::sample): bit-exact against the oracle, whose
    Poisson solutions are checked analytically in tests/test_oracle_3d.py"""
    from test_oracle_3d import _unit_source
    if case == "ball":
        sd = sphere_scene3(subdiv=2, value=lambda x, y, z: 0.25 * x)
        sd["probe"] = (0.5, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0))
        sd["source"] = _unit_source()
        w, h = 16, 16
    else:
        flux = (lambda x, y, z, f: 0.0) if case.startswith("cube") else (lambda x, y, z, f: {4: 0.5, 5: -0.5}.get(f, 0.0))
        sd = cube_scene3(n=2, d_faces=(0, 1), n_faces=(2, 3, 4, 5), value=lambda x, y, z: x, flux=flux, weld=case.startswith("cube"))
        rng = np.random.default_rng(4)
        sd["source"] = {"rgb": rng.uniform(0, 2, (3, 4, 5, 3)).astype(np.float32), "index_scale": (4.0, 3.0, 2.0),
                        "index_offset": (0.0, 0.0, 0.0), "intensity": 1.5}
        w, h = (16, 12) if case.startswith("cube") else (19, 13)
        if not case.startswith("cube"):
            sd["mask"] = (np.arange(w * h) % 4 != 1).astype(np.uint8)
    ref = _same_solve(oracle, sd, w, h, 12, 64, 2e-3)
    assert np.any(ref["field"] != 0)
    if case != "ball":
        assert ref["neumann_hits"] > 0
        assert not np.array_equal(ref["field"][:, 0], ref["field"][:, 1])      # the random source grid differs per channel


def test_3d_debug_channels_match_oracle(oracle):
    """renderDirichletSDF / renderSilhouetteSDF / renderSource of the 3-D integrator: one query per pixel at its evaluation point"""
    from test_oracle_3d import _unit_source
    sd = cube_scene3(n=3, d_faces=(0, 1), n_faces=(2, 3, 4, 5), value=lambda x, y, z: x, flux=lambda x, y, z, f: 0.0)
    sd["probe"] = (0.8, (0.5, 0.5, 0.4), (0.0, 0.6, 0.8), (1.0, 0.0, 0.0))      # a tilted slice that leaves the cube
    sd["source"] = dict(_unit_source(n=4, lo=-0.2, hi=1.2), intensity=2.0)
    sd["source"]["rgb"] = np.random.default_rng(1).uniform(0, 1, (4, 4, 4, 3)).astype(np.float32)
    it = _it(sd, 21, 17, 1, 4, 1e-3)
    for which in (0, 1):
        got, want = it.render_sdf(which), oracle.render_sdf3(sd, 21, 17, which)
        assert np.array_equal(got, want) and np.isfinite(got).all() and got.max() > 0.1
    assert np.array_equal(it.render_source(), oracle.render_source3(sd, 21, 17))
    it.close()
    # without the meshes / the source: +inf and zeros
    only_d = sphere_scene3(subdiv=1, value=lambda x, y, z: 1.0)
    it = _it(only_d, 8, 8, 1, 4, 1e-3)
    assert np.all(np.isinf(it.render_sdf(1))) and np.all(it.render_source() == 0)
    it.close()


@pytest.mark.parametrize("knobs", [{"WOST3_WAVE": "0", "WOST3_COOP": "0"}, {"WOST3_WAVE": "0", "WOST3_COOP": "1"}, {"WOST3_WAVE": "1", "WOST3_COOP": "0"},
                                   {"WOST3_WAVE": "0", "WOST3_COOP": "2"}, {"WOST3_COOP": "1"},
                                   {"WOST3_POOL_CAP": "96"}, {"WOST3_POOL_CAP": "200", "WOST3_RAY_TRIGGER": "1", "WOST3_CP_TRIGGER": "1"}])
def test_3d_wave_cooperative_queries_and_their_fallbacks_match_oracle(oracle, monkeypatch, knobs):
    """the tree queries of a walk answered by the wave through its LDS task pools (closest_triangle_pool, closest_silhouette3_wave,
    ray_closest3_wave: the default) against the per-lane descents they replace, each side alone and both; pools so small that
    the waves cannot take a batch and answer the old way (96 tasks: the 64 roots leave room for ten node tasks); slot tasks
    served as soon as one exists.  Every variant: the oracle's field and counters."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    _same_solve(oracle, _shell_scene(2, 3), 14, 12, 6, 64, 2e-3)
    _same_solve(oracle, _shell_scene(2, 3, flux=lambda x, y, z: 0.3 * y), 12, 10, 4, 48, 2e-3)
    _same_solve(oracle, sphere_scene3(subdiv=3, radius=1.0, value=lambda x, y, z: x * y), 24, 20, 6, 32, 2e-3)
    sd = cube_scene3(n=6, d_faces=(0, 1), n_faces=(2, 3, 4, 5), value=lambda x, y, z: x, flux=lambda x, y, z, f: 0.1 * f)
    _same_solve(oracle, sd, 12, 12, 8, 48, 2e-3)


def test_3d_queries_and_walks_far_outside_the_meshes(oracle):
    """an OPEN Neumann shell lets walkers escape: positions and radii of 10^3 .. 10^7 mesh units occur, where the
    rounding of box and primitive distances grows with |q| -- the tree queries must still give the flat answers"""
    sd = _shell_scene(2, 3)
    keep = np.ones(len(sd["n_tris"]), bool)
    keep[::7] = False                                   # holes all over the shell
    sd["n_tris"] = sd["n_tris"][keep]
    assert len(sd["n_tris"]) > 1000
    it = _it(sd, 8, 8, 1, 4, 1e-3)
    rng = np.random.default_rng(11)
    d = rng.normal(size=(4000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = (d * 10.0 ** rng.uniform(0.5, 7.0, (4000, 1))).astype(np.float32)
    for x, y in zip(it.closest_point(pts), oracle.closest_point3(sd["d_verts"], sd["d_tris"], pts)):
        assert np.array_equal(x, y)
    V, T = sd["n_verts"], sd["n_tris"]
    assert np.array_equal(it.closest_silhouette(pts), oracle.closest_silhouette3(V, T, pts))
    aim = (-pts / np.linalg.norm(pts, axis=1, keepdims=True) + rng.normal(scale=1e-7, size=pts.shape)).astype(np.float32)
    tmax = (np.linalg.norm(pts, axis=1) * 2.0).astype(np.float32)
    got, ref = it.ray_intersect(pts, aim, tmax), oracle.ray_intersect3(V, T, pts, aim, tmax)
    assert np.array_equal(got[0], ref[0])
    hit = ref[0] != 0
    assert np.array_equal(got[1][hit], ref[1][hit]) and np.array_equal(got[2][hit], ref[2][hit])
    it.close()
    ref = _same_solve(oracle, sd, 14, 12, 8, 96, 2e-3)
    assert ref["walks_truncated"] > 0 and ref["neumann_hits"] > 0


def test_3d_random_scenes_match_the_oracle():
    """tools/fuzz/fuzz_parity3d.py: random bumpy icospheres of 20 .. 1280 triangles on either boundary kind, holes, emissive or
    not, doubled and zero-area triangles, scales 1e-3 .. 1e3, probes that look at the scene from 40 scene sizes away"""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz", "fuzz_parity3d.py"), "0", "40"], capture_output=True, text=True,
                         timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "fuzz3d 0..39: 0 mismatches" in out.stdout, out.stdout[-3000:]
