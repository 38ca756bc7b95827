import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_GPU_ORDER = [
    # BASELINE.json configs 1 -> 5, the headline parity evidence
    ("test_gpu_parity.py", ("test_config1_", "test_refill_launch_matches_oracle", "test_full_size_properties_config2",
                            "test_full_size_config3_")),
    ("test_guided_integrator.py", ("test_gpu_config4_", "test_gpu_config5_")),
    # rows a21 - a27: the guided path against the oracle
    ("test_guided_integrator.py", ("test_gpu_training_end_to_end", "test_gpu_first_pass_records", "test_gpu_frozen_network",
                                   "test_gpu_guided_edge_cases", "test_gpu_unguided_depths", "test_gpu_uniform_fraction",
                                   "test_gpu_guided_with_source", "test_gpu_fused_sample_kernel", "test_gpu_train_pixel",
                                   "test_gpu_training_pixel", "test_gpu_sharded", "test_gpu_trained_solve")),
    ("test_guided_distribution.py", ("",)),
    ("test_guided_network.py", ("",)),
    # rows a1 - a20: the uniform path, its queries and edge cases
    ("test_gpu_parity.py", ("",)),
    ("test_guided_integrator.py", ("",)),
    # (b) the host mirror, f1 - f4
    ("test_host_exec.py", ("",)),
    ("test_gpu_3d.py", ("",)),
    ("test_vmf.py", ("",)),
    ("test_guided_3d.py", ("",)),
    ("test_gpu_build3.py", ("",)),
    ("test_gpu_far_trees.py", ("",)),
]
# the long statistical / fuzz / multi-process tests go behind everything else whatever file they are in
# (the half-precision mode's run-to-run checks too: they are strict -- two solves, same bits -- and a box that misbehaves there must not
# cost the parity tests behind them under -x; the fp32 case of the same test, the parity mode, stays in the first rank)
_GPU_LAST = ("test_gpu_config4_at_full_size[16]", "test_gpu_half_precision_training_kernels_repeat", "test_random_scenes", "test_gpu_random_scenes", "test_3d_random_scenes", "test_gpu_tree_sized", "test_gpu_guiding_reduces",
             "test_gpu_reordered_training", "test_gpu_half_precision_network_mode_is_unbiased", "test_gpu_guided_on_ladybug_agrees",
             "test_gpu_full_frame_guided_properties", "test_bench_", "test_gpu_two_ranks", "test_ground_truth_sample_count",
             "test_gpu_guided3_half_precision_solve_is_unbiased")


def _gpu_rank(item):
    if item.get_closest_marker("gpu") is None:
        return -1
    fname, name = os.path.basename(str(item.fspath)), item.name
    if any(name.startswith(p) for p in _GPU_LAST):
        return len(_GPU_ORDER)
    for rank, (f, prefixes) in enumerate(_GPU_ORDER):
        if f == fname and any(name.startswith(p) for p in prefixes):
            return rank
    return len(_GPU_ORDER) - 1


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not failed) where torch reports that there is no HIP device, so a
    plain `pytest tests` works in the build container too (WOST_SKIP_GPU_TESTS=1 forces the skip).  With a
    device present nothing is skipped -- a missing libwost_hip.so must fail loudly there -- and a torch
    that cannot be imported or asked is an error, not a skip: a broken environment on the GPU box must not
    turn into a green run with zero GPU tests."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu") is not None]
    if not gpu_items:
        return
    # a run that is cut off (the driver's clock, a slow box) must lose the least important tests: the BASELINE configurations
    # first, then the parity tests of the path's rows, unit queries, the host mirror, the 3-D variants, and last the fuzzers
    # and the statistical full-frame tests.  (stable: the order inside a rank is the collection order)
    items.sort(key=_gpu_rank)
    reason = None
    if os.environ.get("WOST_SKIP_GPU_TESTS") == "1":
        reason = "WOST_SKIP_GPU_TESTS=1"
    else:
        import torch                                 # an ImportError here fails the collection on purpose
        if torch.cuda.device_count() <= 0:           # does not initialise the GPU
            reason = "no HIP device"
    if reason:
        skip = pytest.mark.skip(reason=reason)
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def oracle_libm():
    from oracle.oracle import Oracle
    return Oracle(libm=True)


@pytest.fixture(scope="session")
def ladybug():
    from elaina_amd import Problem
    return Problem.load_scene("ladybug")


@pytest.fixture(scope="session")
def fille():
    from elaina_amd import Problem
    return Problem.load_scene("fille")


def box_problem(lo=0.0, hi=1.0, n_per_side=8, d_sides=(0, 1, 2, 3), n_sides=(), value=None, flux=None,
                probe=None):
    """Axis-aligned CCW box [lo,hi]^2 split into n_per_side segments per side.
    sides: 0 bottom (y=lo), 1 right (x=hi), 2 top (y=hi), 3 left (x=lo).
    value(x,y) -> Dirichlet value, flux(x,y,side) -> Neumann colour."""
    from elaina_amd import Problem
    corners = [(lo, lo), (hi, lo), (hi, hi), (lo, hi)]

    def side_mesh(sides, fn):
        verts, segs, cols = [], [], []
        for s in sides:
            a, b = np.array(corners[s]), np.array(corners[(s + 1) % 4])
            base = len(verts)
            for k in range(n_per_side + 1):
                p = a + (b - a) * (k / n_per_side)
                verts.append(p)
                v = 0.0 if fn is None else fn(float(p[0]), float(p[1]), s)
                cols.append([v, v, v, v, v, v])
            for k in range(n_per_side):
                segs.append((base + k, base + k + 1))
        if not verts:
            return None, None, None
        return (np.asarray(verts, np.float32), np.asarray(segs, np.int32), np.asarray(cols, np.float32))

    dv, ds, dc = side_mesh(d_sides, (lambda x, y, s: value(x, y)) if value else None)
    nv, ns, nc = side_mesh(n_sides, flux)
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    if probe is None:
        probe = (half * 0.9, mid, mid, 0.0, 1.0)
    return Problem(d_verts=dv, d_segs=ds, d_colors=dc, n_verts=nv, n_segs=ns, n_colors=nc, probe=probe)


def wiggly_problem(n_neumann=3000, n_dirichlet=400, emissive=False, open_gap=0):
    """A non-convex CCW Neumann boundary r(t) = 100 (1 + .2 sin 7t + .05 sin 31t) with
    n_neumann segments around a Dirichlet circle of radius 15 (n_dirichlet segments).
    open_gap > 0 removes that many Neumann segments (open polyline ends = silhouettes)."""
    from elaina_amd import Problem
    t = np.linspace(0.0, 2.0 * np.pi, n_neumann, endpoint=False)
    r = 100.0 * (1.0 + 0.2 * np.sin(7 * t) + 0.05 * np.sin(31 * t))
    nv = np.stack([r * np.cos(t), r * np.sin(t)], 1).astype(np.float32)
    ns = np.stack([np.arange(n_neumann), (np.arange(n_neumann) + 1) % n_neumann], 1).astype(np.int32)
    if open_gap:
        ns = ns[:-open_gap]
    td = np.linspace(0.0, 2.0 * np.pi, n_dirichlet, endpoint=False)
    dv = np.stack([15.0 * np.cos(td) + 5.0, 15.0 * np.sin(td) - 3.0], 1).astype(np.float32)
    ds = np.stack([np.arange(n_dirichlet), (np.arange(n_dirichlet) + 1) % n_dirichlet], 1).astype(np.int32)
    dc = np.zeros((n_dirichlet, 6), np.float32)
    dc[:, 0:3] = (0.5 + 0.5 * np.cos(td))[:, None] * np.array([1.0, 0.5, 0.25])
    dc[:, 3:6] = 0.3
    nc = None
    if emissive:
        nc = np.zeros((n_neumann, 6), np.float32)
        nc[:, 0:3] = (0.01 * np.sin(3 * t))[:, None]
        nc[:, 3:6] = nc[:, 0:3]
    return Problem(d_verts=dv, d_segs=ds, d_colors=dc, n_verts=nv, n_segs=ns, n_colors=nc,
                   probe=(110.0, 0.0, 0.0, 0.0, 1.0))


def cube_scene3(n=4, d_faces=(0, 1, 2, 3, 4, 5), n_faces=(), value=None, flux=None, lo=0.0, hi=1.0, probe=None, weld=True):
    """Axis-aligned cube [lo,hi]^3 as a 3-D scene dict for the oracle / Problem3: every face is an
    n x n grid of quads cut into two triangles, outward normals.  Faces: 0 x=lo, 1 x=hi, 2 y=lo,
    3 y=hi, 4 z=lo, 5 z=hi.  value(x,y,z) -> Dirichlet value, flux(x,y,z,face) -> Neumann colour
    (derivative along the inward normal).  The probe is the slice z = mid.  weld: faces of one mesh share
    the vertices on the cube's edges (watertight); per-face colours that jump across an edge need weld=False."""
    def face_mesh(faces, fn):
        verts, tris, cols = [], [], []
        for f in faces:
            axis, side = f // 2, f % 2
            base = len(verts)
            for j in range(n + 1):
                for i in range(n + 1):
                    a, b = lo + (hi - lo) * i / n, lo + (hi - lo) * j / n
                    p = [0.0, 0.0, 0.0]
                    p[axis] = hi if side else lo
                    p[(axis + 1) % 3], p[(axis + 2) % 3] = a, b
                    verts.append(p)
                    v = 0.0 if fn is None else fn(p[0], p[1], p[2], f)
                    cols.append([v] * 6)
            for j in range(n):
                for i in range(n):
                    v00, v10 = base + j * (n + 1) + i, base + j * (n + 1) + i + 1
                    v01, v11 = v00 + n + 1, v10 + n + 1
                    # (axis+1, axis+2, axis) is right-handed: counter-clockwise in (a, b) faces +axis
                    quad = [(v00, v10, v11), (v00, v11, v01)] if side else [(v00, v11, v10), (v00, v01, v11)]
                    tris += quad
        if not verts:
            return None, None, None
        # weld the vertices the faces share along the cube's edges (a closed, watertight surface)
        index, wv, wc, remap = {}, [], [], []
        for p, c in zip(verts, cols):
            k = tuple(round(x, 9) for x in p) if weld else len(wv)
            if k not in index:
                index[k] = len(wv)
                wv.append(p)
                wc.append(c)
            remap.append(index[k])
        tris = [(remap[a], remap[b], remap[c]) for a, b, c in tris]
        return np.asarray(wv, np.float32), np.asarray(tris, np.int32), np.asarray(wc, np.float32)

    dv, dt, dc = face_mesh(d_faces, (lambda x, y, z, f: value(x, y, z)) if value else None)
    nv, nt, nc = face_mesh(n_faces, flux)
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    if probe is None:
        probe = (0.9 * half, (mid, mid, mid), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0))
    return {"d_verts": dv, "d_tris": dt, "d_colors": dc, "n_verts": nv, "n_tris": nt, "n_colors": nc, "probe": probe,
            "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}


def sphere_scene3(subdiv=2, radius=1.0, value=None, probe=None):
    """Icosphere (20 * 4^subdiv triangles, outward normals) with Dirichlet values value(x,y,z)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    verts = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[k] = len(verts) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    V = (np.asarray(verts) * radius).astype(np.float32)
    T = np.asarray(f, np.int32)
    cols = np.zeros((len(V), 6), np.float32)
    if value is not None:
        cols[:] = np.asarray([value(*p) for p in V], np.float32)[:, None]
    if probe is None:
        probe = (0.8 * radius, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0))
    return {"d_verts": V, "d_tris": T, "d_colors": cols, "n_verts": None, "n_tris": None, "n_colors": None, "probe": probe,
            "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
