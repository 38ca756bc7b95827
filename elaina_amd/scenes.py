"""Synthetic scenes for the variance checks of tests/ and bench.py (no reference counterpart: the
reference ships data files only)."""
import numpy as np

from .problem import Problem

BRIGHT_DISC_CENTRE = (78.0, 72.0)
BRIGHT_DISC_AABB = ((-1.0, -1.0), (101.0, 101.0))


def _circle(c, r, n, value):
    t = np.linspace(0.0, 2.0 * np.pi, n, endpoint=False)
    v = np.stack([c[0] + r * np.cos(t), c[1] + r * np.sin(t)], 1)
    s = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1)
    return v, s, np.full((n, 6), value, np.float32)


def bright_disc_scene(bright_radius=3.0, dark_radius=14.0):
    """A small bright Dirichlet disc (value 1) and a large dark one (value 0) inside a reflecting
    (zero-flux Neumann) box [0, 100]^2: u(x) = P(a walk from x reaches the bright disc first), 0.31 on
    average and small over most of the frame -- a scene where sampling the direction of a walk step in
    proportion to what comes back from it must lower the variance (the purpose of the guided
    integrator, reference integrator/guided/): from most points the contribution arrives from the
    direction of the bright disc."""
    bv, bs, bc = _circle(BRIGHT_DISC_CENTRE, bright_radius, 64, 1.0)
    dv, ds, dc = _circle((32.0, 36.0), dark_radius, 128, 0.0)
    d_verts = np.concatenate([bv, dv]).astype(np.float32)
    d_segs = np.concatenate([bs, ds + len(bv)]).astype(np.int32)
    d_cols = np.concatenate([bc, dc]).astype(np.float32)
    k = 8
    pts = []
    for a, b in (((0, 0), (100, 0)), ((100, 0), (100, 100)), ((100, 100), (0, 100)), ((0, 100), (0, 0))):      # counter-clockwise
        for i in range(k):
            pts.append((a[0] + (b[0] - a[0]) * i / k, a[1] + (b[1] - a[1]) * i / k))
    n_verts = np.asarray(pts, np.float32)
    n_segs = np.stack([np.arange(4 * k), (np.arange(4 * k) + 1) % (4 * k)], 1).astype(np.int32)
    return Problem(d_verts=d_verts, d_segs=d_segs, d_colors=d_cols, n_verts=n_verts, n_segs=n_segs,
                   probe=(49.0, 50.0, 50.0, 0.0, 1.0))


def mixture_mean_direction(raw33):
    """mean direction of the von Mises mixture behind one row of raw network outputs (8 x (lambda, kappa,
    mu.x, mu.y) + selection logit; reference guided/train.h:60-79, guided/distribution.h:136-160):
    sum_i w_i A(kappa_i) mu_i with A = I1 / I0, the mean resultant length of a von Mises lobe"""
    from scipy.special import ive
    r = np.asarray(raw33, np.float64)
    lam = np.exp(np.clip(r[0:32:4], -10.0, 15.0))
    kap = np.exp(np.clip(r[1:32:4], -10.0, 15.0))
    mu = np.stack([r[2:32:4], r[3:32:4]], 1)
    n = np.linalg.norm(mu, axis=1, keepdims=True)
    mu = np.where(n > 0, mu / np.where(n > 0, n, 1.0), mu)
    a = ive(1, kap) / ive(0, kap)
    return ((lam / lam.sum())[:, None] * a[:, None] * mu).sum(0)
