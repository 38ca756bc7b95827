// wost_build2.hip -- the segment LBVH of the 2-D path built on the device (Problem<2>::build_bvh / compute_silhouettes; the
// reference builds its trees on the device: core/problem.cu:31-37, 48-54).  Same output as the host builder lbvh_build.cpp,
// bit for bit -- segment records, Morton order, the perimeter-weighted (SAH) top-down assignment of the segments to the leaves
// of the implicit 4-ary tree, oriented child boxes, normal cones, the compact scan copies -- with the host builder kept as the
// checker (wost_mesh_build_check, tests/test_gpu_build2.py) and behind WOST_HOST_BUILD=1.
//
// How the two agree:
//   * every stored floating-point number is a chain of correctly rounded operations executed the same way on both sides
//     (lbvh_fit.h; no contraction), an exact min / max, or derived from an integer sum: nothing depends on the order in which
//     threads arrive;
//   * the top-down refinement -- per node: sort the node's segments by centroid x, sweep the split position, the same by y,
//     keep the cheaper, and that twice more for the two halves -- runs LEVEL-synchronously: the ranges of all nodes of a level
//     are sorted at once by a stable radix sort on (range, orderable float) keys, the sweeps are segmented scans of bounding
//     boxes (exact), a candidate's cost is the host's expression in double precision, and the host's "first strict minimum in
//     sweep order" is an atomic minimum on (cost bits) followed by one on the sweep position among the minima.  A stable sort
//     on equal keys keeps the previous order, exactly like std::stable_sort in the host's recursion.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/wost.h"
#include "lbvh.h"
#include "lbvh_fit.h"
#include "wost_device.h"
#include "wost_internal.h"
#include "wost_build2.h"

namespace wost {

namespace {

struct B2Shape {
    int n, nv, levels, cap, first_leaf, n_all;      // n_all = nodes that have children (first_leaf + cap)
    long long n_slots;
};

struct B2Meta {
    uint32_t lo[2], hi[2];      // the mesh's box, floats in an order-preserving unsigned encoding (atomicMin / atomicMax)
    int32_t bad;                // an index out of range
    int32_t emissive;           // any non-zero colour
    int32_t n_scan;             // occupied slots
};

__device__ __forceinline__ uint32_t b2_enc(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float b2_dec(uint32_t e)
{
    const uint32_t u = (e >> 31) ? (e & 0x7fffffffu) : ~e;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// a float as an unsigned key that sorts like operator< (-0 and +0 are one key: a stable sort must not tell them apart)
__device__ __forceinline__ uint32_t b2_okey(float f)
{
    if (f == 0.0f) f = 0.0f;
    return b2_enc(f);
}
__device__ __forceinline__ float b2_ext(const B2Meta *m)
{
    return fmaxf(fmaxf(fabsf(b2_dec(m->lo[0])), fabsf(b2_dec(m->hi[0]))), fmaxf(fabsf(b2_dec(m->lo[1])), fabsf(b2_dec(m->hi[1]))));
}
__device__ __forceinline__ uint32_t b2_part1by1(uint32_t x)
{
    x &= 0x0000ffff;
    x = (x | (x << 8)) & 0x00FF00FF;
    x = (x | (x << 4)) & 0x0F0F0F0F;
    x = (x | (x << 2)) & 0x33333333;
    x = (x | (x << 1)) & 0x55555555;
    return x;
}

__global__ void b2_init_kernel(B2Meta *m)
{
    m->lo[0] = m->lo[1] = 0xffffffffu;
    m->hi[0] = m->hi[1] = 0u;
    m->bad = 0; m->emissive = 0; m->n_scan = 0;
}

// ---- segment records in original order, vertex adjacency, the mesh's box (lbvh_build.cpp "flat records") -------------------
__global__ __launch_bounds__(256) void b2_flat_kernel(const float *verts, const int32_t *segs, const float *colors, int n, int nv, DevFlatSeg *flat, float *flatCol,
                                                      int32_t *vprev, int32_t *vnext, B2Meta *meta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int i0 = segs[2 * i], i1 = segs[2 * i + 1];
    if (i0 < 0 || i1 < 0 || i0 >= nv || i1 >= nv) {
        meta->bad = 1;
        return;
    }
    DevFlatSeg s;
    s.ax = verts[2 * i0];
    s.ay = verts[2 * i0 + 1];
    const float bx = verts[2 * i1], by = verts[2 * i1 + 1];
    s.ex = bx - s.ax;
    s.ey = by - s.ay;
    const float len2 = __builtin_fmaf(s.ex, s.ex, s.ey * s.ey);
    s.inv_len2 = (len2 > 0.0f) ? 1.0f / len2 : 0.0f;
    s.len = sqrtf(len2);
    if (s.len > 0.0f) {
        s.nx = s.ey / s.len;
        s.ny = -s.ex / s.len;
    } else {
        s.nx = 0.0f;
        s.ny = 0.0f;
    }
    s.cx = __builtin_fmaf(0.5f, s.ex, s.ax);
    s.cy = __builtin_fmaf(0.5f, s.ey, s.ay);
    if (s.len > 0.0f) {
        s.ux = s.ex / s.len;
        s.uy = s.ey / s.len;
    } else {
        s.ux = 1.0f;
        s.uy = 0.0f;
    }
    s.hl = 0.5f * s.len;
    s.pad0 = s.pad1 = s.pad2 = 0.0f;
    flat[i] = s;
    atomicMin(&vnext[i0], i);      // lowest segment index wins
    atomicMin(&vprev[i1], i);
    atomicMin(&meta->lo[0], b2_enc(fminf(s.ax, bx)));
    atomicMax(&meta->hi[0], b2_enc(fmaxf(s.ax, bx)));
    atomicMin(&meta->lo[1], b2_enc(fminf(s.ay, by)));
    atomicMax(&meta->hi[1], b2_enc(fmaxf(s.ay, by)));
    float *c = flatCol + (size_t)i * 12;
    bool any = false;
    for (int k = 0; k < 3; ++k) {
        const float c0 = colors ? colors[6 * i0 + k] : 0.0f, c1 = colors ? colors[6 * i1 + k] : 0.0f;
        const float c2 = colors ? colors[6 * i0 + 3 + k] : 0.0f, c3 = colors ? colors[6 * i1 + 3 + k] : 0.0f;
        c[0 + k] = c0; c[3 + k] = c1; c[6 + k] = c2; c[9 + k] = c3;
        any = any || c0 != 0.0f || c1 != 0.0f || c2 != 0.0f || c3 != 0.0f;
    }
    if (any) meta->emissive = 1;
}

__global__ __launch_bounds__(256) void b2_sil_kernel(const float *verts, int nv, int32_t *vprev, int32_t *vnext, const DevFlatSeg *flat, DevSilVertex *sil, float4 *silN,
                                                     const B2Meta *meta)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv || meta->bad) return;
    const int p = vprev[v] == 0x7fffffff ? -1 : vprev[v], q = vnext[v] == 0x7fffffff ? -1 : vnext[v];
    vprev[v] = p;
    vnext[v] = q;
    sil[v] = DevSilVertex{verts[2 * v], verts[2 * v + 1], p, q};
    float4 nn = {0.0f, 0.0f, 0.0f, 0.0f};
    if (p >= 0) { nn.x = flat[p].nx; nn.y = flat[p].ny; }
    if (q >= 0) { nn.z = flat[q].nx; nn.w = flat[q].ny; }
    silN[v] = nn;
}

// ---- Morton codes of the centroids, and the refinement's items in original order ---------------------------------------------
struct B2Items {
    float *cx, *cy;       // centroid as the refinement sorts by it: a + 0.5f e in fp32
    float4 *box;          // lo.x lo.y hi.x hi.y of the segment's end points
};
__global__ __launch_bounds__(256) void b2_code_kernel(const DevFlatSeg *flat, const float *verts, const int32_t *segs, int n, const B2Meta *meta, uint32_t *code, int32_t *idx,
                                                      B2Items it)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || meta->bad) return;
    const DevFlatSeg s = flat[i];
    const float lox = b2_dec(meta->lo[0]), loy = b2_dec(meta->lo[1]), hix = b2_dec(meta->hi[0]), hiy = b2_dec(meta->hi[1]);
    const double sx = (hix > lox) ? 65535.0 / ((double)hix - lox) : 0.0;
    const double sy = (hiy > loy) ? 65535.0 / ((double)hiy - loy) : 0.0;
    const double cx = (double)s.ax + 0.5 * (double)s.ex, cy = (double)s.ay + 0.5 * (double)s.ey;
    const uint32_t qx = (uint32_t)fmin(65535.0, fmax(0.0, (cx - lox) * sx));
    const uint32_t qy = (uint32_t)fmin(65535.0, fmax(0.0, (cy - loy) * sy));
    code[i] = b2_part1by1(qx) | (b2_part1by1(qy) << 1);
    idx[i] = i;
    const int i1 = segs[2 * i + 1];
    const float bx = verts[2 * i1], by = verts[2 * i1 + 1];
    it.cx[i] = s.ax + 0.5f * s.ex;
    it.cy[i] = s.ay + 0.5f * s.ey;
    it.box[i] = float4{fminf(s.ax, bx), fminf(s.ay, by), fmaxf(s.ax, bx), fmaxf(s.ay, by)};
}

// ---- the refinement: one round of split2 over all ranges of a level ---------------------------------------------------------
// perm[k] = the original index of the item at position k; a round's ranges [rb[r], re[r]) are disjoint, in order, and cover [0, n)
__global__ __launch_bounds__(256) void b2_rid_kernel(const int32_t *re, int n_ranges, int n, uint32_t *rid)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int lo = 0, hi = n_ranges - 1;      // first range whose end lies beyond k
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (re[mid] > k) hi = mid;
        else lo = mid + 1;
    }
    rid[k] = (uint32_t)lo;
}
// axis 0 / 1: the centroid coordinate; axis 2: the coordinate of the axis each range has chosen (the host's last stable sort)
__global__ __launch_bounds__(256) void b2_key_kernel(const int32_t *perm, const uint32_t *rid, B2Items it, const int32_t *range_axis, int axis, int n, uint64_t *keys)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int o = perm[k];
    const uint32_t r = rid[k];
    const int a = axis == 2 ? range_axis[r] : axis;
    keys[k] = ((uint64_t)r << 32) | b2_okey(a == 0 ? it.cx[o] : it.cy[o]);
}
struct B2BoxOp {
    __host__ __device__ float4 operator()(const float4 &a, const float4 &b) const { return float4{fminf(a.x, b.x), fminf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }
};
// the boxes of the items in position order and in reversed position order (a suffix sweep = a prefix sweep of the mirror image)
__global__ __launch_bounds__(256) void b2_vals_kernel(const int32_t *perm, const uint32_t *rid, B2Items it, int n, float4 *fwd, float4 *rev, uint32_t *rid_rev)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float4 b = it.box[perm[k]];
    fwd[k] = b;
    rev[n - 1 - k] = b;
    rid_rev[n - 1 - k] = rid[k];
}
__device__ __forceinline__ float b2_perim(float4 b) { return (b.z - b.x) + (b.w - b.y); }
__device__ __forceinline__ double b2_leaves(int c) { return (double)((c + kLeafSize - 1) / kLeafSize); }
// a split candidate of position k of its range: cost in leaf visits and its place in the host's sweep order (0 = "no split")
__device__ __forceinline__ bool b2_candidate(const int32_t *rb, const int32_t *re, const uint32_t *rid, const float4 *pre, const float4 *suf_rev, int n, int cap_side, int k,
                                             bool special, double &cost, uint32_t &ord, uint32_t &r_out)
{
    const uint32_t r = rid[k];
    const int b = rb[r], e = re[r], cnt = e - b, i = k - b;
    r_out = r;
    if (cnt <= 1) return false;
    const int lo = max(0, cnt - cap_side), hi = min(cnt, cap_side);
    if (special) {
        // everything in one child: competes on the first axis when it fits
        if (i != 0 || hi != cnt) return false;
        cost = b2_leaves(cnt) * (double)b2_perim(suf_rev[n - 1 - b]);
        ord = 0u;
        return true;
    }
    const int kk = i + 1;       // items b .. k on the left, k + 1 .. e - 1 on the right
    if (kk >= cnt || kk < lo || kk > hi) return false;
    cost = b2_leaves(kk) * (double)b2_perim(pre[k]) + b2_leaves(cnt - kk) * (double)b2_perim(suf_rev[n - 1 - (k + 1)]);
    ord = (uint32_t)kk;
    return true;
}
__global__ __launch_bounds__(256) void b2_cost_min_kernel(const int32_t *rb, const int32_t *re, const uint32_t *rid, const float4 *pre, const float4 *suf_rev, int n, int cap_side,
                                                          int axis, unsigned long long *best_cost)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    // costs are not negative: their bit patterns order like the numbers.  The lanes of a wave mostly belong to ONE range (the top
    // levels' ranges are thousands of items long): their minimum goes to memory as one atomic, not sixty-four on the same word
    unsigned long long mine = ~0ull;
    uint32_t r = 0xffffffffu;
    if (k < n) {
        r = rid[k];
        for (int special = (axis == 0 ? 1 : 0); special >= 0; --special) {
            double cost;
            uint32_t ord, r2;
            if (b2_candidate(rb, re, rid, pre, suf_rev, n, cap_side, k, special != 0, cost, ord, r2) && cost < INFINITY)
                mine = min(mine, (unsigned long long)__double_as_longlong(cost));
        }
    }
    const uint32_t r0 = __shfl(r, 0);
    if (__all(r == r0 || k >= n)) {
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned int lo = __shfl_xor((unsigned int)(mine & 0xffffffffull), m), hi = __shfl_xor((unsigned int)(mine >> 32), m);
            mine = min(mine, ((unsigned long long)hi << 32) | lo);
        }
        if ((threadIdx.x & 63) == 0 && mine != ~0ull && r0 != 0xffffffffu) atomicMin(&best_cost[r0], mine);
    } else if (mine != ~0ull) {
        atomicMin(&best_cost[r], mine);
    }
}
__global__ __launch_bounds__(256) void b2_cost_arg_kernel(const int32_t *rb, const int32_t *re, const uint32_t *rid, const float4 *pre, const float4 *suf_rev, int n, int cap_side,
                                                          int axis, const unsigned long long *best_cost, uint32_t *best_ord)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    for (int special = (axis == 0 ? 1 : 0); special >= 0; --special) {
        double cost;
        uint32_t ord, r;
        if (b2_candidate(rb, re, rid, pre, suf_rev, n, cap_side, k, special != 0, cost, ord, r) && cost < INFINITY &&
            (unsigned long long)__double_as_longlong(cost) == best_cost[r])
            atomicMin(&best_ord[r], ord);
    }
}
__global__ __launch_bounds__(256) void b2_reset_kernel(unsigned long long *c0, unsigned long long *c1, uint32_t *o0, uint32_t *o1, int n_ranges)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ranges) return;
    c0[r] = c1[r] = 0x7ff0000000000000ull;
    o0[r] = o1[r] = 0xffffffffu;
}
// the split of every range: the host's split2 return value, and the axis the range is left sorted by
__global__ __launch_bounds__(256) void b2_decide_kernel(const int32_t *rb, const int32_t *re, int n_ranges, int cap_side, const unsigned long long *c0, const unsigned long long *c1,
                                                        const uint32_t *o0, const uint32_t *o1, int32_t *split, int32_t *range_axis)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ranges) return;
    const int b = rb[r], e = re[r], cnt = e - b;
    const int lo = max(0, cnt - cap_side), hi = min(cnt, cap_side);
    if (cnt <= 1) {
        split[r] = min(e, b + hi);
        range_axis[r] = 0;
        return;
    }
    double best = INFINITY;
    int axis = 0, bk = (lo + hi) / 2;
    if (o0[r] != 0xffffffffu) {
        best = __longlong_as_double((long long)c0[r]);
        bk = o0[r] == 0u ? cnt : (int)o0[r];
    }
    if (o1[r] != 0xffffffffu && __longlong_as_double((long long)c1[r]) < best) {
        axis = 1;
        bk = (int)o1[r];
    }
    split[r] = b + min(max(bk, lo), hi);
    range_axis[r] = axis;
}
// round A -> round B: (b, m), (m, e) of every range; round B -> the next level: four children per node
__global__ __launch_bounds__(256) void b2_halves_kernel(const int32_t *rb, const int32_t *re, const int32_t *split, int n_ranges, int32_t *rb2, int32_t *re2)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ranges) return;
    rb2[2 * r] = rb[r]; re2[2 * r] = split[r];
    rb2[2 * r + 1] = split[r]; re2[2 * r + 1] = re[r];
}
__global__ __launch_bounds__(256) void b2_slot_of_kernel(const int32_t *rb, const int32_t *re, const uint32_t *rid, const int32_t *perm, int n, int32_t *slot_of)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t r = rid[k];
    slot_of[(size_t)r * kLeafSize + (k - rb[r])] = perm[k];
}

// ---- leaves: slot records, the last level's child records, the moments and normal sums of every leaf -------------------------
struct B2Sums {
    FitSums fit;
    ConeSums cone;
};
// extents of a node's end points along the sixteen FIXED candidate directions: minima and maxima, so a node's are the union of its
// children's -- made bottom-up with the sums, not by a sweep of the node's segments per direction (which was 17 sweeps per child, a
// quarter of the whole build); only the principal axis, a different one for every node, still needs its own sweep
struct B2Fixed {
    FitExtent e[kFitDirs];
};
__device__ __forceinline__ void b2_add_seg_normal(ConeSums &c, const DevFlatSeg *flat, int s)
{
    if (flat[s].len > 0.0f) cone_add_normal(c, flat[s].nx, flat[s].ny);
}
__global__ __launch_bounds__(256) void b2_leaf_kernel(B2Shape S, const int32_t *slot_of, const DevFlatSeg *flat, const float *flatCol, const float *verts, const int32_t *segs,
                                                      const int32_t *vprev, const int32_t *vnext, const B2Meta *meta, float4 *segA, float *segInv, int32_t *segOrig, float *segCol,
                                                      int2 *segVerts, float *nodes, float *cones, B2Sums *sums, B2Fixed *fixed)
{
    const int L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= S.cap || meta->bad) return;
    const float lox = b2_dec(meta->lo[0]), loy = b2_dec(meta->lo[1]), hix = b2_dec(meta->hi[0]), hiy = b2_dec(meta->hi[1]);
    const double grid = fit_grid_scale(lox, loy, hix, hiy);
    B2Sums a{};
    B2Fixed fx;
    for (int d = 0; d < kFitDirs; ++d) fx.e[d] = fit_extent_empty();
    float *nd = nodes + (size_t)(S.first_leaf + L) * WOST_NODE_FLOATS;
    float *cn = cones + (size_t)(S.first_leaf + L) * 20;
    for (int j = 0; j < kLeafSize; ++j) {
        const size_t k = (size_t)L * kLeafSize + j;
        const int o = slot_of[k];
        // the last level's "children" are single segments: exact records, no cone
        cn[0 + j] = 1.0f; cn[4 + j] = 0.0f; cn[8 + j] = -1.0f; cn[12 + j] = 0.0f; cn[16 + j] = 0.0f;
        if (o < 0) {
            segA[k] = float4{kFarCoord, kFarCoord, 0.0f, 0.0f};
            segInv[k] = 0.0f;
            segOrig[k] = kFarIndex;
            for (int c = 0; c < 12; ++c) segCol[k * 12 + c] = 0.0f;
            segVerts[k] = int2{-1, -1};
            nd[0 + j] = kFarCoord; nd[4 + j] = kFarCoord; nd[8 + j] = 1.0f; nd[12 + j] = 0.0f; nd[16 + j] = 0.0f; nd[20 + j] = 0.0f;
            continue;
        }
        const DevFlatSeg s = flat[o];
        segA[k] = float4{s.ax, s.ay, s.ex, s.ey};
        segInv[k] = s.inv_len2;
        segOrig[k] = o;
        for (int c = 0; c < 12; ++c) segCol[k * 12 + c] = flatCol[(size_t)o * 12 + c];
        segVerts[k] = int2{segs[2 * o], segs[2 * o + 1]};
        nd[0 + j] = s.cx; nd[4 + j] = s.cy; nd[8 + j] = s.ux; nd[12 + j] = s.uy; nd[16 + j] = s.hl; nd[20 + j] = 0.0f;
        b2_add_seg_normal(a.cone, flat, o);
        for (int e = 0; e < 2; ++e) {
            const int v = segs[2 * o + e];
            fit_add_point(a.fit, verts[2 * v], verts[2 * v + 1], lox, loy, grid);
            for (int d = 0; d < kFitDirs; ++d) fit_extent_add(fx.e[d], fit_dir(d).c, fit_dir(d).s, verts[2 * v], verts[2 * v + 1]);
            if (vprev[v] < 0 || vnext[v] < 0) a.cone.open = 1;
            if (vprev[v] >= 0) b2_add_seg_normal(a.cone, flat, vprev[v]);
            if (vnext[v] >= 0) b2_add_seg_normal(a.cone, flat, vnext[v]);
        }
    }
    sums[S.first_leaf + L] = a;
    fixed[S.first_leaf + L] = fx;
}
__global__ __launch_bounds__(256) void b2_inner_sums_kernel(int level_first, int level_count, const B2Meta *meta, B2Sums *sums, B2Fixed *fixed)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= level_count || meta->bad) return;
    const int g = level_first + p;
    B2Sums a{};
    B2Fixed fx;
    for (int d = 0; d < kFitDirs; ++d) fx.e[d] = fit_extent_empty();
    for (int j = 1; j <= kArity; ++j) {
        const B2Sums c = sums[kArity * g + j];
        fit_add_sums(a.fit, c.fit);
        cone_add_sums(a.cone, c.cone);
        for (int d = 0; d < kFitDirs; ++d) {
            const FitExtent ce = fixed[kArity * g + j].e[d];
            FitExtent &e = fx.e[d];
            e.umin = fmin(e.umin, ce.umin); e.umax = fmax(e.umax, ce.umax);
            e.vmin = fmin(e.vmin, ce.vmin); e.vmax = fmax(e.vmax, ce.vmax);
        }
    }
    sums[g] = a;
    fixed[g] = fx;
}

__device__ __forceinline__ double b2_shfl_xor(double v, int m)
{
    const long long x = __double_as_longlong(v);
    const int lo = __shfl_xor((int)(x & 0xffffffffll), m), hi = __shfl_xor((int)(x >> 32), m);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ void b2_child_span(const B2Shape &S, int c4, long long &s0, long long &s1)
{
    int level = 0;
    for (long long first = 0, count = 1; c4 >= first + count; first += count, count *= 4) ++level;
    const long long pos = c4 - (((1ll << (2 * level)) - 1) / 3);
    long long span = kLeafSize;
    for (int i = level; i < S.levels; ++i) span *= 4;
    s0 = pos * span;
    s1 = s0 + span;
}

// ---- the oriented box of every child of every inner node, one wave per child (lbvh_build.cpp fit_obb) -------------------------
__global__ __launch_bounds__(256) void b2_obb_kernel(B2Shape S, const int32_t *slot_of, const float *verts, const int32_t *segs, const B2Meta *meta, const B2Sums *sums,
                                                     const B2Fixed *fixed, double obb_pad_rel, float *nodes)
{
    const long long w = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= 4ll * S.first_leaf || meta->bad) return;
    const int g = (int)(w >> 2), j = (int)(w & 3), c4 = 4 * g + 1 + j;
    float *nd = nodes + (size_t)g * WOST_NODE_FLOATS;
    const FitSums fs = sums[c4].fit;
    if (fs.n == 0) {
        if (lane == 0) { nd[0 + j] = kFarCoord; nd[4 + j] = kFarCoord; nd[8 + j] = 1.0f; nd[12 + j] = 0.0f; nd[16 + j] = 0.0f; nd[20 + j] = 0.0f; }
        return;
    }
    long long s0, s1;
    b2_child_span(S, c4, s0, s1);
    const double obb_pad = (double)b2_ext(meta) * obb_pad_rel + 1e-30;
    double best = INFINITY;
    float out[6] = {0, 0, 0, 0, 0, 0};
    for (int a = -1; a < kFitDirs; ++a) {
        float uxf, uyf;
        if (a < 0) fit_pca_axis(fs, uxf, uyf);
        else { uxf = fit_dir(a).c; uyf = fit_dir(a).s; }
        FitExtent e = fit_extent_empty();
        if (a < 0) {
            for (long long k = s0 + lane; k < s1; k += 64) {
                const int o = slot_of[k];
                if (o < 0) continue;
                for (int q = 0; q < 2; ++q) {
                    const int v = segs[2 * o + q];
                    fit_extent_add(e, uxf, uyf, verts[2 * v], verts[2 * v + 1]);
                }
            }
            for (int m = 32; m >= 1; m >>= 1) {
                e.umin = fmin(e.umin, b2_shfl_xor(e.umin, m)); e.umax = fmax(e.umax, b2_shfl_xor(e.umax, m));
                e.vmin = fmin(e.vmin, b2_shfl_xor(e.vmin, m)); e.vmax = fmax(e.vmax, b2_shfl_xor(e.vmax, m));
            }
        } else {
            e = fixed[c4].e[a];      // the union of the children's extents along this direction (minima and maxima: exact)
        }
        const double score = fit_score(e);
        if (score < best) {
            best = score;
            fit_box(e, uxf, uyf, obb_pad, out);
        }
    }
    if (lane == 0) { nd[0 + j] = out[0]; nd[4 + j] = out[1]; nd[8 + j] = out[2]; nd[12 + j] = out[3]; nd[16 + j] = out[4]; nd[20 + j] = out[5]; }
}

// ---- the normal cone of every child of every inner node, one wave per child ---------------------------------------------------
__global__ __launch_bounds__(256) void b2_cone_kernel(B2Shape S, const int32_t *slot_of, const DevFlatSeg *flat, const float *verts, const int32_t *segs, const int32_t *vprev,
                                                      const int32_t *vnext, const B2Meta *meta, const B2Sums *sums, const float *nodes, float *cones)
{
    const long long w = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= 4ll * S.first_leaf || meta->bad) return;
    const int g = (int)(w >> 2), j = (int)(w & 3), c4 = 4 * g + 1 + j;
    float *cn = cones + (size_t)g * 20;
    const float *nd = nodes + (size_t)g * WOST_NODE_FLOATS;
    const float cx = nd[0 + j], cy = nd[4 + j];
    long long s0, s1;
    b2_child_span(S, c4, s0, s1);
    const ConeSums cs = sums[c4].cone;
    double ax = 0.0, ay = 0.0;
    const bool prunes = cone_axis(cs, ax, ay);
    double cmin = 1.0, rad = 0.0;
    for (long long k = s0 + lane; k < s1; k += 64) {
        const int o = slot_of[k];
        if (o < 0) continue;
        if (prunes && flat[o].len > 0.0f) cmin = fmin(cmin, cone_cos_to(ax, ay, flat[o].nx, flat[o].ny));
        for (int q = 0; q < 2; ++q) {
            const int v = segs[2 * o + q];
            const double dx = (double)verts[2 * v] - (double)cx, dy = (double)verts[2 * v + 1] - (double)cy;
            rad = fmax(rad, sqrt(dx * dx + dy * dy));
            if (!prunes) continue;
            const int p = vprev[v], n2 = vnext[v];
            if (p >= 0 && flat[p].len > 0.0f) cmin = fmin(cmin, cone_cos_to(ax, ay, flat[p].nx, flat[p].ny));
            if (n2 >= 0 && flat[n2].len > 0.0f) cmin = fmin(cmin, cone_cos_to(ax, ay, flat[n2].nx, flat[n2].ny));
        }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        cmin = fmin(cmin, b2_shfl_xor(cmin, m));
        rad = fmax(rad, b2_shfl_xor(rad, m));
    }
    if (lane != 0) return;
    const float radf = cone_radius(rad, b2_ext(meta));
    float c4v[4];
    if (!prunes || !cone_finish(ax, ay, cmin, c4v)) {
        cn[0 + j] = 1.0f; cn[4 + j] = 0.0f; cn[8 + j] = -1.0f; cn[12 + j] = 0.0f; cn[16 + j] = radf;
        return;
    }
    cn[0 + j] = c4v[0]; cn[4 + j] = c4v[1]; cn[8 + j] = c4v[2]; cn[12 + j] = c4v[3]; cn[16 + j] = radf;
}

// ---- closest_point_wave's compact copies of the occupied slots, in slot order ---------------------------------------------------
__global__ __launch_bounds__(256) void b2_occupied_kernel(const int32_t *slot_of, long long n_slots, int32_t *flag)
{
    const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (k < n_slots) flag[k] = slot_of[k] >= 0 ? 1 : 0;
}
__global__ __launch_bounds__(256) void b2_scan_copy_kernel(B2Shape S, const int32_t *slot_of, const int32_t *dst, const float *nodes, long long n_scan_padded, float4 *scanBox,
                                                           float *scanHl, int2 *scanId)
{
    const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (k < S.n_slots && slot_of[k] >= 0) {
        const float *nd = nodes + ((size_t)S.first_leaf + k / 4) * WOST_NODE_FLOATS + (k & 3);
        const int d = dst[k];
        scanBox[d] = float4{nd[0], nd[4], nd[8], nd[12]};
        scanHl[d] = nd[16];
        scanId[d] = int2{(int)k, slot_of[k]};
    }
    // padding to a multiple of 256 with records that cannot win (every occupied slot lies below S.n)
    if (k >= S.n && k < n_scan_padded) {
        scanBox[k] = float4{1.0e18f, 1.0e18f, 1.0f, 0.0f};
        scanHl[k] = 0.0f;
        scanId[k] = int2{-1, kFarIndex};
    }
}
__global__ __launch_bounds__(256) void b2_fill_i32_kernel(int32_t *p, long long n, int32_t v)
{
    const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (k < n) p[k] = v;
}

struct B2Buffer {
    char *base = nullptr;
    size_t size = 0;
    size_t take(size_t bytes)
    {
        const size_t off = size;
        size += (bytes + 255) / 256 * 256;
        return off;
    }
    template <class T>
    T *at(size_t off) const { return reinterpret_cast<T *>(base + off); }
    ~B2Buffer()
    {
        if (base) (void)hipFree(base);
    }
};

}  // namespace

#define B2_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return set_error(WOST_ERR_DEVICE, std::string("mesh build: ") + #expr + ": " + hipGetErrorString(e_)); \
    } while (0)

static inline dim3 b2_grid(long long count) { return dim3((unsigned)std::max<long long>(1, (count + 255) / 256)); }

int build_tree_device(const wost_mesh_desc &d, DeviceTree2 &out_tree)
{
    DevMesh &v = out_tree.view;
    v = DevMesh{};
    v.n_segs = d.n_segs;
    if (d.n_segs == 0) return WOST_OK;
    if (!d.verts || !d.segs || d.n_verts <= 0) return set_error(WOST_ERR_INVALID, "mesh: segment index out of range or null arrays");
    if (d.n_segs > (1 << 26)) return set_error(WOST_ERR_UNSUPPORTED, "mesh: more than 2^26 segments");
    const int n = d.n_segs, nv = d.n_verts;
    // an index out of range is refused HERE, before any launch: the kernels below leave their outputs unwritten when the code
    // kernel reports it, and the sorts and gathers that follow would read stale keys as positions (round-5 advisor finding);
    // the indices are in host memory, the loop is O(n)
    for (int i = 0; i < 2 * n; ++i)
        if (d.segs[i] < 0 || d.segs[i] >= nv) return set_error(WOST_ERR_INVALID, "mesh: segment index out of range or null arrays");
    B2Shape S{};
    S.n = n; S.nv = nv;
    const int n_leaves = (n + kLeafSize - 1) / kLeafSize;
    S.levels = 1; S.cap = kArity;
    while (S.cap < n_leaves) { S.cap *= kArity; ++S.levels; }
    S.first_leaf = (S.cap - 1) / (kArity - 1);
    S.n_all = S.first_leaf + S.cap;
    S.n_slots = (long long)S.cap * kLeafSize;
    const size_t n_slots = (size_t)S.n_slots;
    const size_t n_scan_pad = ((size_t)n + 255) / 256 * 256;
    int rid_bits = 1;
    while ((1ll << rid_bits) < 2ll * S.cap) ++rid_bits;

    B2Buffer out, tmp;
    const size_t o_nodes = out.take((size_t)S.n_all * WOST_NODE_FLOATS * 4), o_cones = out.take((size_t)S.n_all * 20 * 4), o_segA = out.take(n_slots * 16), o_segInv = out.take(n_slots * 4),
                 o_segOrig = out.take(n_slots * 4), o_segCol = out.take(n_slots * 48), o_segVerts = out.take(n_slots * 8), o_flat = out.take((size_t)n * sizeof(DevFlatSeg)),
                 o_flatCol = out.take((size_t)n * 48), o_sil = out.take((size_t)nv * sizeof(DevSilVertex)), o_silN = out.take((size_t)nv * 16),
                 o_scanBox = out.take(n_scan_pad * 16), o_scanHl = out.take(n_scan_pad * 4), o_scanId = out.take(n_scan_pad * 8);
    size_t sort_a = 0, sort_b = 0, scan_c = 0, scan_d = 0;
    B2_TRY(rocprim::radix_sort_pairs(nullptr, sort_a, (uint32_t *)nullptr, (uint32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, 0u, 32u));
    B2_TRY(rocprim::radix_sort_pairs(nullptr, sort_b, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, 0u, (unsigned)(32 + rid_bits)));
    B2_TRY(rocprim::inclusive_scan_by_key(nullptr, scan_c, (uint32_t *)nullptr, (float4 *)nullptr, (float4 *)nullptr, (size_t)n, B2BoxOp()));
    B2_TRY(rocprim::exclusive_scan(nullptr, scan_d, (int32_t *)nullptr, (int32_t *)nullptr, 0, n_slots, rocprim::plus<int32_t>()));
    const size_t max_ranges = (size_t)2 * S.cap;
    const size_t t_meta = tmp.take(sizeof(B2Meta)), t_verts = tmp.take((size_t)nv * 8), t_segs = tmp.take((size_t)n * 8), t_colors = tmp.take(d.colors ? (size_t)nv * 24 : 0),
                 t_vprev = tmp.take((size_t)nv * 4), t_vnext = tmp.take((size_t)nv * 4), t_code = tmp.take((size_t)n * 4), t_code2 = tmp.take((size_t)n * 4),
                 t_idx = tmp.take((size_t)n * 4), t_perm = tmp.take((size_t)n * 4), t_perm2 = tmp.take((size_t)n * 4), t_icx = tmp.take((size_t)n * 4),
                 t_icy = tmp.take((size_t)n * 4), t_ibox = tmp.take((size_t)n * 16), t_rid = tmp.take((size_t)n * 4), t_rid_rev = tmp.take((size_t)n * 4),
                 t_keys = tmp.take((size_t)n * 8), t_keys2 = tmp.take((size_t)n * 8), t_fwd = tmp.take((size_t)n * 16), t_rev = tmp.take((size_t)n * 16),
                 t_pre = tmp.take((size_t)n * 16), t_suf = tmp.take((size_t)n * 16), t_rb = tmp.take(max_ranges * 4), t_re = tmp.take(max_ranges * 4),
                 t_rb2 = tmp.take(max_ranges * 4), t_re2 = tmp.take(max_ranges * 4), t_split = tmp.take(max_ranges * 4), t_axis = tmp.take(max_ranges * 4),
                 t_c0 = tmp.take(max_ranges * 8), t_c1 = tmp.take(max_ranges * 8), t_o0 = tmp.take(max_ranges * 4), t_o1 = tmp.take(max_ranges * 4),
                 t_slot_of = tmp.take(n_slots * 4), t_flag = tmp.take(n_slots * 4), t_dst = tmp.take(n_slots * 4), t_sums = tmp.take((size_t)S.n_all * sizeof(B2Sums)), t_fixed = tmp.take((size_t)S.n_all * sizeof(B2Fixed)),
                 t_rp = tmp.take(std::max(std::max(sort_a, sort_b), std::max(scan_c, scan_d)));
    B2_TRY(hipMalloc((void **)&out.base, out.size));
    B2_TRY(hipMalloc((void **)&tmp.base, tmp.size));
    hipStream_t st = nullptr;
    B2Meta *meta = tmp.at<B2Meta>(t_meta);
    float *verts = tmp.at<float>(t_verts);
    int32_t *segs = tmp.at<int32_t>(t_segs);
    float *colors = d.colors ? tmp.at<float>(t_colors) : nullptr;
    int32_t *vprev = tmp.at<int32_t>(t_vprev), *vnext = tmp.at<int32_t>(t_vnext);
    DevFlatSeg *flat = out.at<DevFlatSeg>(o_flat);
    float *flatCol = out.at<float>(o_flatCol);
    B2_TRY(hipMemcpyAsync(verts, d.verts, (size_t)nv * 8, hipMemcpyHostToDevice, st));
    B2_TRY(hipMemcpyAsync(segs, d.segs, (size_t)n * 8, hipMemcpyHostToDevice, st));
    if (d.colors) B2_TRY(hipMemcpyAsync(colors, d.colors, (size_t)nv * 24, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(b2_init_kernel, dim3(1), dim3(1), 0, st, meta);
    hipLaunchKernelGGL(b2_fill_i32_kernel, b2_grid(nv), dim3(256), 0, st, vprev, (long long)nv, 0x7fffffff);
    hipLaunchKernelGGL(b2_fill_i32_kernel, b2_grid(nv), dim3(256), 0, st, vnext, (long long)nv, 0x7fffffff);
    hipLaunchKernelGGL(b2_flat_kernel, b2_grid(n), dim3(256), 0, st, verts, segs, colors, n, nv, flat, flatCol, vprev, vnext, meta);
    hipLaunchKernelGGL(b2_sil_kernel, b2_grid(nv), dim3(256), 0, st, verts, nv, vprev, vnext, flat, out.at<DevSilVertex>(o_sil), out.at<float4>(o_silN), meta);
    B2Items items{tmp.at<float>(t_icx), tmp.at<float>(t_icy), tmp.at<float4>(t_ibox)};
    hipLaunchKernelGGL(b2_code_kernel, b2_grid(n), dim3(256), 0, st, flat, verts, segs, n, meta, tmp.at<uint32_t>(t_code), tmp.at<int32_t>(t_idx), items);
    int32_t *perm = tmp.at<int32_t>(t_perm), *perm2 = tmp.at<int32_t>(t_perm2);
    size_t rp_bytes = sort_a;
    B2_TRY(rocprim::radix_sort_pairs(tmp.at<void>(t_rp), rp_bytes, tmp.at<uint32_t>(t_code), tmp.at<uint32_t>(t_code2), tmp.at<int32_t>(t_idx), perm, (size_t)n, 0u, 32u, st));

    // ---- the refinement, level by level: round A splits every node's range in two, round B each half again ----
    int32_t *rb = tmp.at<int32_t>(t_rb), *re = tmp.at<int32_t>(t_re), *rb2 = tmp.at<int32_t>(t_rb2), *re2 = tmp.at<int32_t>(t_re2);
    int32_t *split = tmp.at<int32_t>(t_split), *range_axis = tmp.at<int32_t>(t_axis);
    uint32_t *rid = tmp.at<uint32_t>(t_rid), *rid_rev = tmp.at<uint32_t>(t_rid_rev);
    unsigned long long *c0 = tmp.at<unsigned long long>(t_c0), *c1 = tmp.at<unsigned long long>(t_c1);
    uint32_t *o0 = tmp.at<uint32_t>(t_o0), *o1 = tmp.at<uint32_t>(t_o1);
    uint64_t *keys = tmp.at<uint64_t>(t_keys), *keys2 = tmp.at<uint64_t>(t_keys2);
    float4 *fwd = tmp.at<float4>(t_fwd), *rev = tmp.at<float4>(t_rev), *pre = tmp.at<float4>(t_pre), *suf = tmp.at<float4>(t_suf);
    {
        const int32_t r0[2] = {0, n};
        B2_TRY(hipMemcpyAsync(rb, &r0[0], 4, hipMemcpyHostToDevice, st));
        B2_TRY(hipMemcpyAsync(re, &r0[1], 4, hipMemcpyHostToDevice, st));
    }
    auto sort_by = [&](int axis) -> int {
        hipLaunchKernelGGL(b2_key_kernel, b2_grid(n), dim3(256), 0, st, perm, rid, items, range_axis, axis, n, keys);
        size_t bytes = sort_b;
        B2_TRY(rocprim::radix_sort_pairs(tmp.at<void>(t_rp), bytes, keys, keys2, perm, perm2, (size_t)n, 0u, (unsigned)(32 + rid_bits), st));
        std::swap(perm, perm2);
        return WOST_OK;
    };
    auto round = [&](int n_ranges, int cap_side) -> int {
        hipLaunchKernelGGL(b2_rid_kernel, b2_grid(n), dim3(256), 0, st, re, n_ranges, n, rid);
        hipLaunchKernelGGL(b2_reset_kernel, b2_grid(n_ranges), dim3(256), 0, st, c0, c1, o0, o1, n_ranges);
        for (int axis = 0; axis < 2; ++axis) {
            const int rc = sort_by(axis);
            if (rc != WOST_OK) return rc;
            hipLaunchKernelGGL(b2_vals_kernel, b2_grid(n), dim3(256), 0, st, perm, rid, items, n, fwd, rev, rid_rev);
            size_t bytes = scan_c;
            B2_TRY(rocprim::inclusive_scan_by_key(tmp.at<void>(t_rp), bytes, rid, fwd, pre, (size_t)n, B2BoxOp(), rocprim::equal_to<uint32_t>(), st));
            bytes = scan_c;
            B2_TRY(rocprim::inclusive_scan_by_key(tmp.at<void>(t_rp), bytes, rid_rev, rev, suf, (size_t)n, B2BoxOp(), rocprim::equal_to<uint32_t>(), st));
            hipLaunchKernelGGL(b2_cost_min_kernel, b2_grid(n), dim3(256), 0, st, rb, re, rid, pre, suf, n, cap_side, axis, axis == 0 ? c0 : c1);
            hipLaunchKernelGGL(b2_cost_arg_kernel, b2_grid(n), dim3(256), 0, st, rb, re, rid, pre, suf, n, cap_side, axis, axis == 0 ? c0 : c1, axis == 0 ? o0 : o1);
        }
        hipLaunchKernelGGL(b2_decide_kernel, b2_grid(n_ranges), dim3(256), 0, st, rb, re, n_ranges, cap_side, c0, c1, o0, o1, split, range_axis);
        return sort_by(2);      // the order the host's recursion leaves behind: by x where x won, by y elsewhere (stable: untouched)
    };
    int n_ranges = 1;
    for (int level = 0; level < S.levels; ++level) {
        int child_cap = kLeafSize;
        for (int l = level + 1; l < S.levels; ++l) child_cap *= kArity;
        int rc = round(n_ranges, 2 * child_cap);
        if (rc != WOST_OK) return rc;
        hipLaunchKernelGGL(b2_halves_kernel, b2_grid(n_ranges), dim3(256), 0, st, rb, re, split, n_ranges, rb2, re2);
        std::swap(rb, rb2);
        std::swap(re, re2);
        n_ranges *= 2;
        rc = round(n_ranges, child_cap);
        if (rc != WOST_OK) return rc;
        hipLaunchKernelGGL(b2_halves_kernel, b2_grid(n_ranges), dim3(256), 0, st, rb, re, split, n_ranges, rb2, re2);
        std::swap(rb, rb2);
        std::swap(re, re2);
        n_ranges *= 2;
    }
    // n_ranges == cap: the ranges are the leaves, in position order
    int32_t *slot_of = tmp.at<int32_t>(t_slot_of);
    hipLaunchKernelGGL(b2_fill_i32_kernel, b2_grid((long long)n_slots), dim3(256), 0, st, slot_of, (long long)n_slots, -1);
    hipLaunchKernelGGL(b2_rid_kernel, b2_grid(n), dim3(256), 0, st, re, n_ranges, n, rid);
    hipLaunchKernelGGL(b2_slot_of_kernel, b2_grid(n), dim3(256), 0, st, rb, re, rid, perm, n, slot_of);

    // ---- leaves, sums bottom-up, boxes, cones, the scan copies ----
    float *nodes = out.at<float>(o_nodes), *cones = out.at<float>(o_cones);
    B2Sums *sums = tmp.at<B2Sums>(t_sums);
    B2Fixed *fixed = tmp.at<B2Fixed>(t_fixed);
    B2_TRY(hipMemsetAsync(nodes, 0, (size_t)S.n_all * WOST_NODE_FLOATS * 4, st));
    B2_TRY(hipMemsetAsync(cones, 0, (size_t)S.n_all * 80, st));
    hipLaunchKernelGGL(b2_leaf_kernel, b2_grid(S.cap), dim3(256), 0, st, S, slot_of, flat, flatCol, verts, segs, vprev, vnext, meta, out.at<float4>(o_segA),
                       out.at<float>(o_segInv), out.at<int32_t>(o_segOrig), out.at<float>(o_segCol), out.at<int2>(o_segVerts), nodes, cones, sums, fixed);
    {
        int first = S.first_leaf, count = S.cap;
        for (int l = S.levels - 1; l >= 0; --l) {
            count /= kArity;
            first -= count;
            hipLaunchKernelGGL(b2_inner_sums_kernel, b2_grid(count), dim3(256), 0, st, first, count, meta, sums, fixed);
        }
    }
    const char *e_pad = getenv("WOST_OBB_PAD_LOG2");   // developer knob: absolute pad = ext * 2^-k
    const double obb_pad_rel = std::ldexp(1.0, -(e_pad ? atoi(e_pad) : 21));
    hipLaunchKernelGGL(b2_obb_kernel, b2_grid(256ll * S.first_leaf), dim3(256), 0, st, S, slot_of, verts, segs, meta, sums, fixed, obb_pad_rel, nodes);
    hipLaunchKernelGGL(b2_cone_kernel, b2_grid(256ll * S.first_leaf), dim3(256), 0, st, S, slot_of, flat, verts, segs, vprev, vnext, meta, sums, nodes, cones);
    int32_t *flag = tmp.at<int32_t>(t_flag), *dst = tmp.at<int32_t>(t_dst);
    hipLaunchKernelGGL(b2_occupied_kernel, b2_grid((long long)n_slots), dim3(256), 0, st, slot_of, (long long)n_slots, flag);
    rp_bytes = scan_d;
    B2_TRY(rocprim::exclusive_scan(tmp.at<void>(t_rp), rp_bytes, flag, dst, 0, n_slots, rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(b2_scan_copy_kernel, b2_grid((long long)std::max(n_slots, n_scan_pad)), dim3(256), 0, st, S, slot_of, dst, nodes, (long long)n_scan_pad,
                       out.at<float4>(o_scanBox), out.at<float>(o_scanHl), out.at<int2>(o_scanId));
    B2_TRY(hipGetLastError());
    B2Meta hm{};
    B2_TRY(hipMemcpy(&hm, meta, sizeof(hm), hipMemcpyDeviceToHost));         // the one wait of the build
    if (hm.bad) return set_error(WOST_ERR_INVALID, "mesh: segment index out of range or null arrays");
    const float ext = std::max(std::max(std::fabs(b2_dec(hm.lo[0])), std::fabs(b2_dec(hm.hi[0]))), std::max(std::fabs(b2_dec(hm.lo[1])), std::fabs(b2_dec(hm.hi[1]))));
    v.n_sil = nv; v.levels = S.levels; v.first_leaf = S.first_leaf; v.emissive = hm.emissive;
    v.far2 = 2.25f * ext * ext;
    v.huge2 = 4096.0f * ext * ext;
    v.nodes = out.at<float4>(o_nodes); v.cones = out.at<float4>(o_cones); v.segA = out.at<float4>(o_segA); v.segInv = out.at<float>(o_segInv);
    v.segOrig = out.at<int32_t>(o_segOrig); v.segCol = out.at<float>(o_segCol); v.segVerts = out.at<int2>(o_segVerts); v.flat = flat; v.flatCol = flatCol;
    v.sil = out.at<DevSilVertex>(o_sil); v.silN = out.at<float4>(o_silN);
    v.scanBox = out.at<float4>(o_scanBox); v.scanHl = out.at<float>(o_scanHl); v.scanId = out.at<int2>(o_scanId); v.n_scan = (int32_t)n_scan_pad;
    out_tree.alloc = out.base;
    out_tree.bytes = out.size;
    out.base = nullptr;                     // owned by the caller from here
    return WOST_OK;
}

}  // namespace wost
