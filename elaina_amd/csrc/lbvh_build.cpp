// lbvh_build.cpp -- host builder for the implicit 4-ary LBVH (see lbvh.h).
// Compiled with -ffp-contract=off: the per-segment constants (e, 1/|e|^2, |e|, n) are
// part of the arithmetic contract of the queries and must round exactly as specified
// in DESIGN.md ("segment record").
#include "lbvh.h"

#include <algorithm>
#include <cmath>
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

int build_tree(int32_t n_verts, const float *verts, int32_t n_segs, const int32_t *segs,
               const float *colors, HostTree *t)
{
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
    for (int v = 0; v < n_verts; ++v) {
        if (vprev[v] < 0 && vnext[v] < 0) continue;
        t->sil.push_back(SilVertex{verts[2 * v], verts[2 * v + 1], vprev[v], vnext[v]});
    }

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

    // ---- leaves ------------------------------------------------------------------------
    t->n_leaves = (n_segs + kLeafSize - 1) / kLeafSize;
    int levels = 1;
    int cap = kArity;
    while (cap < t->n_leaves) {
        cap *= kArity;
        ++levels;
    }
    t->levels = levels;
    t->n_leaves_cap = cap;
    t->first_leaf = (cap - 1) / (kArity - 1);
    const size_t n_slots = (size_t)cap * kLeafSize;
    t->segA.assign(n_slots * 4, 0.0f);
    t->segInv.assign(n_slots, 0.0f);
    t->segOrig.assign(n_slots, kFarIndex);
    t->segCol.assign(n_slots * 12, 0.0f);
    t->origToSlot.assign(n_segs, -1);
    for (size_t s = 0; s < n_slots; ++s) {
        t->segA[4 * s + 0] = kFarCoord;
        t->segA[4 * s + 1] = kFarCoord;
    }
    for (int k = 0; k < n_segs; ++k) {
        int o = order[k];
        const FlatSeg &s = t->flat[o];
        t->segA[4 * (size_t)k + 0] = s.ax;
        t->segA[4 * (size_t)k + 1] = s.ay;
        t->segA[4 * (size_t)k + 2] = s.ex;
        t->segA[4 * (size_t)k + 3] = s.ey;
        t->segInv[k] = s.inv_len2;
        t->segOrig[k] = o;
        t->origToSlot[o] = k;
        std::memcpy(&t->segCol[(size_t)k * 12], &t->flatCol[(size_t)o * 12], 12 * sizeof(float));
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
    for (int k = 0; k < n_segs; ++k) {
        int g = t->first_leaf + k / kLeafSize;
        const FlatSeg &s = t->flat[order[k]];
        // second endpoint exactly as given (not a + e)
        int i1 = segs[2 * order[k] + 1];
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
    return 0;
}

}  // namespace wost
