"""Simulate near-first closest-point queries of a WoS walk on ladybug for implicit k-ary Morton trees (k = 4, 8), AABB nodes, leaves of 4 segments:
visits (= dependent node fetches), box tests and segment tests per query, with the previous step's closest segment as the starting bound."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from elaina_amd import Problem
p = Problem.load_scene("ladybug")
V = np.asarray(p.d_verts, np.float64); S = np.asarray(p.d_segs, np.int64)
P0 = V[S[:, 0]]; P1 = V[S[:, 1]]
n = len(S)
# Morton order of segment centres
C = 0.5 * (P0 + P1); lo = C.min(0); hi = C.max(0)
q = ((C - lo) / (hi - lo).max() * 65535).astype(np.uint64)
def part(x):
    x = x & 0xffff; x = (x | (x << 8)) & 0x00ff00ff; x = (x | (x << 4)) & 0x0f0f0f0f; x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555; return x
code = part(q[:, 0]) | (part(q[:, 1]) << np.uint64(1))
order = np.argsort(code, kind='stable'); P0 = P0[order]; P1 = P1[order]
LEAF = 4
n_leaves = (n + LEAF - 1) // LEAF
pad = n_leaves * LEAF - n
P0p = np.concatenate([P0, np.repeat(P0[-1:], pad, 0)]); P1p = np.concatenate([P1, np.repeat(P1[-1:], pad, 0)])
leaf_lo = np.minimum(P0p, P1p).reshape(n_leaves, LEAF, 2).min(1); leaf_hi = np.maximum(P0p, P1p).reshape(n_leaves, LEAF, 2).max(1)
def build(k):
    levels = [(leaf_lo, leaf_hi)]
    while len(levels[-1][0]) > 1:
        l, h = levels[-1]; m = (len(l) + k - 1) // k; padn = m * k - len(l)
        l2 = np.concatenate([l, np.full((padn, 2), np.inf)]).reshape(m, k, 2).min(1); h2 = np.concatenate([h, np.full((padn, 2), -np.inf)]).reshape(m, k, 2).max(1)
        levels.append((l2, h2))
    return levels[::-1]      # root first; last = leaves
def seg_d2(q, a, b):
    e = b - a; w = q - a; L2 = (e * e).sum(-1); t = np.clip((w * e).sum(-1) / np.where(L2 > 0, L2, 1), 0, 1); d = w - t[..., None] * e; return (d * d).sum(-1)
def box_d2(q, l, h):
    d = np.maximum(np.maximum(l - q, q - h), 0); return (d * d).sum(-1)
def query(levels, k, Q, hint):
    """vectorised near-first traversal with per-walker stacks; returns best d2, best seg, counters"""
    W = len(Q); nl = len(levels)
    best = seg_d2(Q, P0p[hint], P1p[hint]); bseg = hint.copy()
    cap = (k - 1) * nl + 2
    st_key = np.full((W, cap), np.inf); st_lvl = np.zeros((W, cap), np.int64); st_idx = np.zeros((W, cap), np.int64); sp = np.zeros(W, np.int64)
    cur_lvl = np.zeros(W, np.int64); cur_idx = np.zeros(W, np.int64); active = np.ones(W, bool)
    inner = np.zeros(W, np.int64); leafv = np.zeros(W, np.int64)
    while active.any():
        a = np.nonzero(active)[0]
        is_leaf = cur_lvl[a] == nl - 1
        # leaves: test LEAF segments
        al = a[is_leaf]
        if len(al):
            leafv[al] += 1
            base = cur_idx[al] * LEAF
            for j in range(LEAF):
                d = seg_d2(Q[al], P0p[base + j], P1p[base + j]); better = d < best[al]
                best[al] = np.where(better, d, best[al]); bseg[al] = np.where(better, base + j, bseg[al])
        ai = a[~is_leaf]
        nxt_lvl = np.zeros(len(a), np.int64); nxt_idx = np.zeros(len(a), np.int64); has_next = np.zeros(len(a), bool)
        if len(ai):
            inner[ai] += 1
            lv = cur_lvl[ai] + 1
            keys = np.full((len(ai), k), np.inf); idxs = np.zeros((len(ai), k), np.int64)
            for c in range(k):
                ci = cur_idx[ai] * k + c
                d = np.full(len(ai), np.inf)
                for L in np.unique(lv):
                    m = lv == L; l, h = levels[L]; ok = m & (ci < len(l))
                    if ok.any(): d[ok] = box_d2(Q[ai][ok], l[ci[ok]], h[ci[ok]])
                keys[:, c] = d; idxs[:, c] = ci
            keys = np.where(keys < best[ai][:, None], keys, np.inf)
            o = np.argsort(keys, 1); keys = np.take_along_axis(keys, o, 1); idxs = np.take_along_axis(idxs, o, 1)
            # nearest child continues, the others are pushed farthest first
            for c in range(k - 1, 0, -1):
                okp = np.isfinite(keys[:, c]); w = ai[okp]
                st_key[w, sp[w]] = keys[okp, c]; st_lvl[w, sp[w]] = lv[okp]; st_idx[w, sp[w]] = idxs[okp, c]; sp[w] += 1
            pos = np.searchsorted(a, ai)
            has_next[pos] = np.isfinite(keys[:, 0]); nxt_lvl[pos] = lv; nxt_idx[pos] = idxs[:, 0]
        # walkers without a next node pop (stale entries culled)
        need = ~has_next
        w = a[need]
        while len(w):
            empty = sp[w] == 0
            active[w[empty]] = False
            w = w[~empty]
            if not len(w): break
            sp[w] -= 1
            kk = st_key[w, sp[w]]; good = kk < best[w]
            g = w[good]
            cur_lvl[g] = st_lvl[g, sp[g]]; cur_idx[g] = st_idx[g, sp[g]]
            w = w[~good]
        hn = a[has_next]
        cur_lvl[hn] = nxt_lvl[has_next]; cur_idx[hn] = nxt_idx[has_next]
    return best, bseg, inner, leafv
rng = np.random.default_rng(0)
W = 4000
probe = p.probe      # (scale, posx, posy, upx, upy)
scale, px, py, ux, uy = [float(v) for v in probe]
pix = rng.integers(0, 1024, (W, 2)); ndc = 2.0 * pix / 1024.0 - 1.0
Q0 = scale * (ndc[:, :1] * np.array([[uy, -ux]]) + ndc[:, 1:] * np.array([[ux, uy]])) + np.array([[px, py]])
res = {}
for k in (4, 8):
    lv = build(k)
    Q = Q0.copy(); hint = np.zeros(W, np.int64)
    _, hint, _, _ = query(lv, k, Q, hint)       # depth-0 query (cached in the kernel): not counted
    tot_inner = tot_leaf = tot_q = 0
    depth = np.zeros(W, np.int64)
    for step in range(40):
        d2, seg, inner, leafv = query(lv, k, Q, hint)
        if step > 0: tot_inner += inner.sum(); tot_leaf += leafv.sum(); tot_q += W
        R = np.sqrt(d2); absorbed = R < 1.0
        th = rng.uniform(0, 2 * np.pi, W); Qn = Q + 0.99 * np.maximum(R, 1e-4)[:, None] * np.stack([np.cos(th), np.sin(th)], 1)
        restart = absorbed | (depth >= 63)
        Q = np.where(restart[:, None], Q0, Qn); depth = np.where(restart, 0, depth + 1); hint = np.where(restart, hint, seg)
        rng = np.random.default_rng(step + 1)     # the same directions for both arities
    res[k] = (tot_inner / tot_q, tot_leaf / tot_q, len(lv))
    print("k=%d: %d levels; per query: %.2f inner + %.2f leaf visits = %.2f dependent fetches; box tests %.1f; segment tests %.1f" % (
        k, len(lv), tot_inner / tot_q, tot_leaf / tot_q, (tot_inner + tot_leaf) / tot_q, k * tot_inner / tot_q, LEAF * tot_leaf / tot_q))
