import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped (not failed) where there is no HIP device, so a plain
    `pytest tests` works in the build container too.  With a device present nothing is skipped: a
    missing libwost_hip.so must fail loudly there."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu") is not None]
    if not gpu_items:
        return
    reason = None
    try:
        import torch
        if torch.cuda.device_count() <= 0:      # does not initialise the GPU
            reason = "no HIP device"
    except Exception as e:                       # pragma: no cover
        reason = "torch unavailable: %r" % (e,)
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
