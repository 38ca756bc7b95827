// lbvh_build.cpp -- host builder for the implicit 4-ary LBVH (see lbvh.h).
// Compiled with -ffp-contract=off: the per-segment constants (e, 1/|e|^2, |e|, n) are
// part of the arithmetic contract of the queries and must round exactly as specified
// in DESIGN.md ("segment record").
#include "lbvh.h"
#include "lbvh_fit.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>

namespace wost {

static inline uint32_t part1by1(uint32_t x)
{
    x &= 0x0000ffff;
    x = (x | (x << 8)) & 0x00FF00FF;
    x = (x | (x << 4)) & 0x0F0F0F0F;
    x = (x | (x << 2)) & 0x33333333;
    x = (x | (x << 1)) & 0x55555555;
    return x;
}

static inline float dot2(float ax, float ay, float bx, float by) { return std::fmaf(ax, bx, ay * by); }

// Oriented box around n points (x0 y0 x1 y1 ..., fp32 coordinates): the tightest (by half perimeter) of the principal axis of
// their moments `sums` and the sixteen fixed directions, measured in the frame of the fp32 axis it will be stored with,
// inflated by a relative 1e-6 plus `pad`.  out = cx cy ux uy hl hw.  The arithmetic is lbvh_fit.h's: the device builder
// (wost_build2.hip) computes the same bits.
void fit_obb(const float *P, size_t n, const FitSums &sums, double obb_pad, float out[6])
{
    double best = std::numeric_limits<double>::infinity();
    for (int a = -1; a < kFitDirs; ++a) {
        float uxf, uyf;
        if (a < 0) fit_pca_axis(sums, uxf, uyf);
        else { uxf = fit_dir(a).c; uyf = fit_dir(a).s; }
        FitExtent e = fit_extent_empty();
        for (size_t i = 0; i < n; ++i) fit_extent_add(e, uxf, uyf, P[2 * i], P[2 * i + 1]);
        const double score = fit_score(e);
        if (score < best) {
            best = score;
            fit_box(e, uxf, uyf, obb_pad, out);
        }
    }
}

int build_tree(int32_t n_verts, const float *verts, int32_t n_segs, const int32_t *segs,
               const float *colors, HostTree *t, bool refine, int extra_levels)
{
    const float inf_f = std::numeric_limits<float>::infinity();
    *t = HostTree();
    t->n_segs = n_segs;
    t->n_verts = n_verts;
    if (n_segs <= 0) return 0;
    if (!verts || !segs || n_verts <= 0) return -1;

    // ---- flat records in original order ------------------------------------------------
    t->flat.resize(n_segs);
    t->flatCol.assign((size_t)n_segs * 12, 0.0f);
    std::vector<int32_t> vprev(n_verts, -1), vnext(n_verts, -1);
    float lox = std::numeric_limits<float>::infinity(), loy = lox, hix = -lox, hiy = -lox;
    for (int i = 0; i < n_segs; ++i) {
        int i0 = segs[2 * i], i1 = segs[2 * i + 1];
        if (i0 < 0 || i1 < 0 || i0 >= n_verts || i1 >= n_verts) return -1;
        FlatSeg &s = t->flat[i];
        s.ax = verts[2 * i0];
        s.ay = verts[2 * i0 + 1];
        float bx = verts[2 * i1], by = verts[2 * i1 + 1];
        s.ex = bx - s.ax;
        s.ey = by - s.ay;
        float len2 = dot2(s.ex, s.ey, s.ex, s.ey);
        s.inv_len2 = (len2 > 0.0f) ? 1.0f / len2 : 0.0f;
        s.len = std::sqrt(len2);
        if (s.len > 0.0f) {
            s.nx = s.ey / s.len;
            s.ny = -s.ex / s.len;
        } else {
            s.nx = 0.0f;
            s.ny = 0.0f;
        }
        // distance form (DESIGN.md "segment distance")
        s.cx = std::fmaf(0.5f, s.ex, s.ax);
        s.cy = std::fmaf(0.5f, s.ey, s.ay);
        if (s.len > 0.0f) {
            s.ux = s.ex / s.len;
            s.uy = s.ey / s.len;
        } else {
            s.ux = 1.0f;
            s.uy = 0.0f;
        }
        s.hl = 0.5f * s.len;
        s.pad0 = s.pad1 = s.pad2 = 0.0f;
        if (vnext[i0] < 0) vnext[i0] = i;  // lowest segment index wins
        if (vprev[i1] < 0) vprev[i1] = i;
        lox = std::min(lox, std::min(s.ax, bx));
        hix = std::max(hix, std::max(s.ax, bx));
        loy = std::min(loy, std::min(s.ay, by));
        hiy = std::max(hiy, std::max(s.ay, by));
        if (colors) {
            float *c = &t->flatCol[(size_t)i * 12];
            for (int k = 0; k < 3; ++k) {
                c[0 + k] = colors[6 * i0 + k];      // left  colour of vertex i0
                c[3 + k] = colors[6 * i1 + k];      // left  colour of vertex i1
                c[6 + k] = colors[6 * i0 + 3 + k];  // right colour of vertex i0
                c[9 + k] = colors[6 * i1 + 3 + k];  // right colour of vertex i1
            }
        }
    }
    t->aabb[0] = lox; t->aabb[1] = loy; t->aabb[2] = hix; t->aabb[3] = hiy;
    t->sil.resize(n_verts);
    for (int v = 0; v < n_verts; ++v) t->sil[v] = SilVertex{verts[2 * v], verts[2 * v + 1], vprev[v], vnext[v]};

    // ---- Morton order of centroids -----------------------------------------------------
    const double sx = (hix > lox) ? 65535.0 / ((double)hix - lox) : 0.0;
    const double sy = (hiy > loy) ? 65535.0 / ((double)hiy - loy) : 0.0;
    std::vector<uint32_t> code(n_segs);
    for (int i = 0; i < n_segs; ++i) {
        const FlatSeg &s = t->flat[i];
        double cx = (double)s.ax + 0.5 * (double)s.ex, cy = (double)s.ay + 0.5 * (double)s.ey;
        uint32_t qx = (uint32_t)std::min(65535.0, std::max(0.0, (cx - lox) * sx));
        uint32_t qy = (uint32_t)std::min(65535.0, std::max(0.0, (cy - loy) * sy));
        code[i] = part1by1(qx) | (part1by1(qy) << 1);
    }
    std::vector<int32_t> order(n_segs);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return code[a] < code[b]; });

    // ---- tree shape -------------------------------------------------------------------
    t->n_leaves = (n_segs + kLeafSize - 1) / kLeafSize;
    int levels = 1;
    int cap = kArity;
    while (cap < t->n_leaves) {
        cap *= kArity;
        ++levels;
    }
    levels += extra_levels;
    for (int i = 0; i < extra_levels; ++i) cap *= kArity;
    t->levels = levels;
    t->n_leaves_cap = cap;
    t->first_leaf = (cap - 1) / (kArity - 1);
    const size_t n_slots = (size_t)cap * kLeafSize;

    // ---- assignment of segments to leaves -----------------------------------------------
    // slot_of[k] = original segment stored in slot k (-1 = padding).  Two policies:
    //  * plain LBVH: consecutive groups of 4 in Morton order fill the leaves left to right;
    //  * refined (default): top-down, every node splits its Morton-ordered set into four
    //    subsets by two rounds of a perimeter-weighted sweep (the surface-area heuristic in
    //    2-D), constrained so that each subset fits the fixed capacity of its implicit subtree.
    std::vector<int32_t> slot_of(n_slots, -1);
    if (!refine) {
        for (int k = 0; k < n_segs; ++k) slot_of[k] = order[k];
    } else {
        struct Item { float cx, cy, lox, loy, hix, hiy; int32_t idx; };
        std::vector<Item> items(n_segs);
        for (int k = 0; k < n_segs; ++k) {
            int o = order[k];
            const FlatSeg &s = t->flat[o];
            int i1 = segs[2 * o + 1];
            float bx = verts[2 * i1], by = verts[2 * i1 + 1];
            items[k] = Item{s.ax + 0.5f * s.ex, s.ay + 0.5f * s.ey, std::min(s.ax, bx), std::min(s.ay, by),
                            std::max(s.ax, bx), std::max(s.ay, by), o};
        }
        std::vector<float> suffix;
        // split items[b,e) into [b,m) and [m,e) with both sides <= cap_side items; returns m
        auto split2 = [&](int b, int e, int cap_side) -> int {
            const int n = e - b;
            const int lo = std::max(0, n - cap_side), hi = std::min(n, cap_side);
            if (n <= 1) return std::min(e, b + hi);
            double best_cost = std::numeric_limits<double>::infinity();
            int best_axis = 0, best_k = (lo + hi) / 2;
            for (int axis = 0; axis < 2; ++axis) {
                std::stable_sort(items.begin() + b, items.begin() + e, [axis](const Item &p, const Item &q) {
                    return axis == 0 ? p.cx < q.cx : p.cy < q.cy;
                });
                suffix.assign(n + 1, 0.0f);
                float lx = inf_f, ly = inf_f, hx = -inf_f, hy = -inf_f;
                for (int i = n - 1; i >= 0; --i) {
                    const Item &it = items[b + i];
                    lx = std::min(lx, it.lox); ly = std::min(ly, it.loy);
                    hx = std::max(hx, it.hix); hy = std::max(hy, it.hiy);
                    suffix[i] = (hx - lx) + (hy - ly);
                }
                // cost in leaf visits: leaves needed on each side times the box perimeter;
                // k == 0 / k == n (no split, everything in one child) compete when they fit
                auto leaves = [](int c) { return (double)((c + kLeafSize - 1) / kLeafSize); };
                if (axis == 0 && hi == n) {
                    const double cost = leaves(n) * suffix[0];
                    if (cost < best_cost) { best_cost = cost; best_axis = 0; best_k = n; }
                }
                lx = inf_f; ly = inf_f; hx = -inf_f; hy = -inf_f;
                for (int k = 1; k < n; ++k) {
                    const Item &it = items[b + k - 1];
                    lx = std::min(lx, it.lox); ly = std::min(ly, it.loy);
                    hx = std::max(hx, it.hix); hy = std::max(hy, it.hiy);
                    if (k < lo || k > hi) continue;
                    const double cost = leaves(k) * ((hx - lx) + (hy - ly)) + leaves(n - k) * suffix[k];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_k = k; }
                }
            }
            if (best_axis == 0)
                std::stable_sort(items.begin() + b, items.begin() + e,
                                 [](const Item &p, const Item &q) { return p.cx < q.cx; });
            return b + std::min(std::max(best_k, lo), hi);
        };
        // recursive descent over the implicit tree
        struct Job { int b, e, level, pos; };
        std::vector<Job> todo{{0, n_segs, 0, 0}};
        while (!todo.empty()) {
            Job j = todo.back();
            todo.pop_back();
            if (j.level == levels) {
                for (int k = j.b; k < j.e; ++k) slot_of[(size_t)j.pos * kLeafSize + (k - j.b)] = items[k].idx;
                continue;
            }
            int child_cap = kLeafSize;  // segments one child subtree can hold
            for (int l = j.level + 1; l < levels; ++l) child_cap *= kArity;
            const int m = split2(j.b, j.e, 2 * child_cap);
            const int ml = split2(j.b, m, child_cap);
            const int mr = split2(m, j.e, child_cap);
            todo.push_back({j.b, ml, j.level + 1, 4 * j.pos + 0});
            todo.push_back({ml, m, j.level + 1, 4 * j.pos + 1});
            todo.push_back({m, mr, j.level + 1, 4 * j.pos + 2});
            todo.push_back({mr, j.e, j.level + 1, 4 * j.pos + 3});
        }
    }

    // ---- leaves ------------------------------------------------------------------------
    t->segA.assign(n_slots * 4, 0.0f);
    t->segInv.assign(n_slots, 0.0f);
    t->segOrig.assign(n_slots, kFarIndex);
    t->segCol.assign(n_slots * 12, 0.0f);
    t->origToSlot.assign(n_segs, -1);
    for (size_t s = 0; s < n_slots; ++s) {
        t->segA[4 * s + 0] = kFarCoord;
        t->segA[4 * s + 1] = kFarCoord;
    }
    for (size_t k = 0; k < n_slots; ++k) {
        const int o = slot_of[k];
        if (o < 0) continue;
        const FlatSeg &s = t->flat[o];
        t->segA[4 * k + 0] = s.ax;
        t->segA[4 * k + 1] = s.ay;
        t->segA[4 * k + 2] = s.ex;
        t->segA[4 * k + 3] = s.ey;
        t->segInv[k] = s.inv_len2;
        t->segOrig[k] = o;
        t->origToSlot[o] = (int32_t)k;
        std::memcpy(&t->segCol[k * 12], &t->flatCol[(size_t)o * 12], 12 * sizeof(float));
    }

    // ---- boxes, bottom-up ----------------------------------------------------------------
    // Padding: the computed closest point of a segment lies within half an ulp of the
    // largest coordinate of its exact bounding box, and box / segment distances carry a
    // relative rounding error of a few 2^-24; ext * 2^-20 covers both for every query
    // within ~16 scene extents (DESIGN.md "pruning slack").
    const float ext = std::max(std::max(std::fabs(lox), std::fabs(hix)), std::max(std::fabs(loy), std::fabs(hiy)));
    const float pad = ext * 0x1p-20f + 1e-30f;
    t->pad = pad;
    const int n_nodes = t->first_leaf + cap;  // heap indices 0 .. n_nodes-1
    std::vector<float> nb((size_t)n_nodes * 4);
    std::vector<char> empty(n_nodes, 1);
    const float inf = std::numeric_limits<float>::infinity();
    for (int g = 0; g < n_nodes; ++g) {
        nb[4 * (size_t)g + 0] = inf; nb[4 * (size_t)g + 1] = inf;
        nb[4 * (size_t)g + 2] = -inf; nb[4 * (size_t)g + 3] = -inf;
    }
    for (size_t k = 0; k < n_slots; ++k) {
        const int o = slot_of[k];
        if (o < 0) continue;
        int g = t->first_leaf + (int)(k / kLeafSize);
        const FlatSeg &s = t->flat[o];
        // second endpoint exactly as given (not a + e)
        int i1 = segs[2 * o + 1];
        float bx = verts[2 * i1], by = verts[2 * i1 + 1];
        float *b = &nb[4 * (size_t)g];
        b[0] = std::min(b[0], std::min(s.ax, bx));
        b[1] = std::min(b[1], std::min(s.ay, by));
        b[2] = std::max(b[2], std::max(s.ax, bx));
        b[3] = std::max(b[3], std::max(s.ay, by));
        empty[g] = 0;
    }
    for (int g = t->first_leaf - 1; g >= 0; --g) {
        float *b = &nb[4 * (size_t)g];
        for (int j = 1; j <= kArity; ++j) {
            int c = kArity * g + j;
            if (empty[c]) continue;
            const float *cb = &nb[4 * (size_t)c];
            b[0] = std::min(b[0], cb[0]); b[1] = std::min(b[1], cb[1]);
            b[2] = std::max(b[2], cb[2]); b[3] = std::max(b[3], cb[3]);
            empty[g] = 0;
        }
    }
    t->boxes.resize((size_t)(n_nodes - 1) * 4);
    for (int g = 1; g < n_nodes; ++g) {
        float *o = &t->boxes[4 * (size_t)(g - 1)];
        const float *b = &nb[4 * (size_t)g];
        if (empty[g]) {
            o[0] = o[1] = o[2] = o[3] = kFarCoord;
        } else {
            o[0] = b[0] - pad; o[1] = b[1] - pad; o[2] = b[2] + pad; o[3] = b[3] + pad;
        }
    }
    // ---- oriented child boxes for every node -------------------------------------------
    {
        const int n_all = t->first_leaf + cap;   // nodes that have children (leaves included)
        t->nodes.assign((size_t)n_all * WOST_NODE_FLOATS, 0.0f);
        // endpoints under every node, gathered bottom-up (leaves first), and their moments on the 2^20 grid of the mesh's box
        std::vector<std::vector<float>> pts(n_all);
        std::vector<FitSums> sums(n_all, FitSums{0, 0, 0, 0, 0, 0});
        const double grid = fit_grid_scale(lox, loy, hix, hiy);
        for (int k = 0; k < cap; ++k) {
            std::vector<float> &P = pts[t->first_leaf + k];
            for (int j = 0; j < kLeafSize; ++j) {
                const int o = slot_of[(size_t)k * kLeafSize + j];
                if (o < 0) continue;
                for (int e = 0; e < 2; ++e) {
                    const int v = segs[2 * o + e];
                    P.push_back(verts[2 * v]); P.push_back(verts[2 * v + 1]);
                    fit_add_point(sums[t->first_leaf + k], verts[2 * v], verts[2 * v + 1], lox, loy, grid);
                }
            }
        }
        for (int g = t->first_leaf - 1; g >= 1; --g)
            for (int j = 1; j <= kArity; ++j) {
                const std::vector<float> &C = pts[kArity * g + j];
                pts[g].insert(pts[g].end(), C.begin(), C.end());
                fit_add_sums(sums[g], sums[kArity * g + j]);
            }
        const char *e_pad = getenv("WOST_OBB_PAD_LOG2");   // developer knob: absolute pad = ext * 2^-k
        const double obb_pad = (double)ext * std::ldexp(1.0, -(e_pad ? atoi(e_pad) : 21)) + 1e-30;
        t->obb_pad = obb_pad;
        auto set_child = [&](int parent, int j, float cx, float cy, float ux, float uy, float hl, float hw) {
            float *nd = &t->nodes[(size_t)parent * WOST_NODE_FLOATS];
            nd[0 + j] = cx; nd[4 + j] = cy; nd[8 + j] = ux; nd[12 + j] = uy; nd[16 + j] = hl; nd[20 + j] = hw;
        };
        // inner levels: fit an oriented box around the endpoints of each child subtree
        for (int g = 0; g < t->first_leaf; ++g) {
            for (int j = 0; j < kArity; ++j) {
                const std::vector<float> &P = pts[kArity * g + 1 + j];
                const size_t n = P.size() / 2;
                if (n == 0) {
                    set_child(g, j, kFarCoord, kFarCoord, 1.0f, 0.0f, 0.0f, 0.0f);
                    continue;
                }
                float ob[6];
                fit_obb(P.data(), n, sums[kArity * g + 1 + j], obb_pad, ob);
                set_child(g, j, ob[0], ob[1], ob[2], ob[3], ob[4], ob[5]);
            }
        }
        // last level: the children are the segments, exact records, no padding
        for (int k = 0; k < cap; ++k)
            for (int j = 0; j < kLeafSize; ++j) {
                const int o = slot_of[(size_t)k * kLeafSize + j];
                if (o < 0) {
                    set_child(t->first_leaf + k, j, kFarCoord, kFarCoord, 1.0f, 0.0f, 0.0f, 0.0f);
                } else {
                    const FlatSeg &s = t->flat[o];
                    set_child(t->first_leaf + k, j, s.cx, s.cy, s.ux, s.uy, s.hl, 0.0f);
                }
            }
    }
    // ---- SNCH normal cones + per-slot vertex ids ------------------------------------------
    {
        const int n_all = t->first_leaf + cap;
        t->cones.assign((size_t)n_all * 20, 0.0f);
        t->segVerts.assign(n_slots * 2, -1);
        for (size_t k = 0; k < n_slots; ++k) {
            const int o = slot_of[k];
            if (o < 0) continue;
            t->segVerts[2 * k] = segs[2 * o];
            t->segVerts[2 * k + 1] = segs[2 * o + 1];
        }
        // per node (heap index): the normals its cone must cover -- those of its segments and of the segments that share a
        // vertex with them --, "has an open end", the end points
        struct Acc { std::vector<float> nrm; ConeSums cs{0, 0, 0, 0}; std::vector<float> pts; };
        std::vector<Acc> acc(n_all);
        auto add_seg_normal = [&](Acc &a, int sidx) {
            const FlatSeg &s = t->flat[sidx];
            if (s.len > 0.0f) {
                a.nrm.push_back(s.nx); a.nrm.push_back(s.ny);
                cone_add_normal(a.cs, s.nx, s.ny);
            }
        };
        for (int k = 0; k < cap; ++k) {
            Acc &a = acc[t->first_leaf + k];
            for (int j = 0; j < kLeafSize; ++j) {
                const int o = slot_of[(size_t)k * kLeafSize + j];
                if (o < 0) continue;
                add_seg_normal(a, o);
                for (int e = 0; e < 2; ++e) {
                    const int v = segs[2 * o + e];
                    a.pts.push_back(verts[2 * v]); a.pts.push_back(verts[2 * v + 1]);
                    if (vprev[v] < 0 || vnext[v] < 0) a.cs.open = 1;
                    if (vprev[v] >= 0) add_seg_normal(a, vprev[v]);
                    if (vnext[v] >= 0) add_seg_normal(a, vnext[v]);
                }
            }
        }
        for (int g = t->first_leaf - 1; g >= 1; --g)
            for (int j = 1; j <= kArity; ++j) {
                const Acc &c = acc[kArity * g + j];
                acc[g].nrm.insert(acc[g].nrm.end(), c.nrm.begin(), c.nrm.end());
                acc[g].pts.insert(acc[g].pts.end(), c.pts.begin(), c.pts.end());
                cone_add_sums(acc[g].cs, c.cs);
            }
        auto set_cone = [&](int parent, int j, float ax, float ay, float ch, float sh, float rad) {
            float *cn = &t->cones[(size_t)parent * 20];
            cn[0 + j] = ax; cn[4 + j] = ay; cn[8 + j] = ch; cn[12 + j] = sh; cn[16 + j] = rad;
        };
        auto fit = [&](int parent, int j, Acc &a, float cx, float cy) {
            double rad = 0.0;
            for (size_t i = 0; i + 1 < a.pts.size(); i += 2) {
                const double dx = (double)a.pts[i] - (double)cx, dy = (double)a.pts[i + 1] - (double)cy;
                rad = std::max(rad, std::sqrt(dx * dx + dy * dy));
            }
            const float radf = cone_radius(rad, ext);
            double ax, ay;
            if (!cone_axis(a.cs, ax, ay)) { set_cone(parent, j, 1.0f, 0.0f, -1.0f, 0.0f, radf); return; }
            double cmin = 1.0;
            for (size_t i = 0; i + 1 < a.nrm.size(); i += 2) cmin = std::min(cmin, cone_cos_to(ax, ay, a.nrm[i], a.nrm[i + 1]));
            float c4[4];
            if (!cone_finish(ax, ay, cmin, c4)) { set_cone(parent, j, 1.0f, 0.0f, -1.0f, 0.0f, radf); return; }
            set_cone(parent, j, c4[0], c4[1], c4[2], c4[3], radf);
        };
        for (int g = 0; g < t->first_leaf; ++g)
            for (int j = 0; j < kArity; ++j) {
                const float *nd = &t->nodes[(size_t)g * WOST_NODE_FLOATS];
                fit(g, j, acc[kArity * g + 1 + j], nd[0 + j], nd[4 + j]);
            }
        // last level: the "children" are single segments; their candidates are their two
        // endpoints, tested exactly by the query, so no cone is needed (cannot prune)
        for (int k = 0; k < cap; ++k)
            for (int j = 0; j < kLeafSize; ++j) set_cone(t->first_leaf + k, j, 1.0f, 0.0f, -1.0f, 0.0f, 0.0f);
    }
    return 0;
}

}  // namespace wost
