// wost_build3.hip -- the triangle LBVH of the 3-D path (Problem<3>::build_bvh; reference core/problem.cu:31-37, 48-54 builds its
// trees on the device): triangle and edge records, Morton order, the implicit 4-ary tree of child boxes, the normal cones of
// the silhouette query, the run boxes of the emissive sampler.
//
// Two builders with the same output, bit for bit:
//   * build_mesh3_device  -- the product path: HIP kernels + rocPRIM radix sorts / scan on the context's device; the only
//                            host work is the tree shape (a function of the triangle count) and one read-back of 48 bytes;
//   * build_mesh3         -- the host builder of rounds 2-3, kept as the checker of the device build
//                            (wost3_mesh_build_check, tests/test_gpu_build3.py) and behind WOST3_HOST_BUILD=1.
// Every floating-point result that is stored is either a single correctly rounded operation chain executed the same way on
// both sides (no contraction: -ffp-contract=off), an exact min / max, or an integer sum: nothing depends on the order in
// which threads arrive.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/wost.h"
#include "lbvh.h"
#include "wost_device.h"
#include "wost_internal.h"
#include "wost_device3.h"
#include "wost_internal3.h"

namespace wost {

// the cone axis is the sum of the normals in fixed point (2^-36: 2^27 normals fit 63 bits)
constexpr double kNormalFix = 68719476736.0;
// cos(10^-4), sin(10^-4): the pad of a cone's half angle
constexpr double kConePadCos = 0.999999995000000004166666665277778, kConePadSin = 9.99999998333333341666666646825397e-5;

// ---- host: LBVH over triangles ----------------------------------------------------------------------

static inline uint32_t part1by2(uint32_t x)
{
    x &= 0x3ff;
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
static inline float hdot3(const float *a, const float *b) { return std::fmaf(a[0], b[0], std::fmaf(a[1], b[1], a[2] * b[2])); }

// returns 0, or -1 on an index out of range
static int build_mesh3(const wost3_mesh_desc &d, HostMesh3 *out)
{
    HostMesh3 &h = *out;
    h = HostMesh3();
    h.n_tris = d.n_tris;
    if (d.n_tris <= 0) return 0;
    if (!d.verts || !d.tris || d.n_verts <= 0) return -1;
    const int n = d.n_tris;
    h.flat.resize(n);
    std::vector<float> cen((size_t)n * 3);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int t = 0; t < n; ++t) {
        const int32_t *iv = d.tris + 3 * (size_t)t;
        for (int k = 0; k < 3; ++k)
            if (iv[k] < 0 || iv[k] >= d.n_verts) return -1;
        DevTri &T = h.flat[t];
        for (int c = 0; c < 3; ++c) {
            T.p0[c] = d.verts[3 * (size_t)iv[0] + c]; T.p1[c] = d.verts[3 * (size_t)iv[1] + c]; T.p2[c] = d.verts[3 * (size_t)iv[2] + c];
            lo[c] = std::min(lo[c], std::min(T.p0[c], std::min(T.p1[c], T.p2[c])));
            hi[c] = std::max(hi[c], std::max(T.p0[c], std::max(T.p1[c], T.p2[c])));
            cen[3 * (size_t)t + c] = (float)(((double)T.p0[c] + T.p1[c] + T.p2[c]) / 3.0);
        }
        // e0, e1, nraw = cross3(e0, e1), unit normal, area: the triangle record of DESIGN.md 2.3
        float e0[3], e1[3];
        for (int c = 0; c < 3; ++c) { e0[c] = T.p1[c] - T.p0[c]; e1[c] = T.p2[c] - T.p0[c]; }
        T.nraw[0] = std::fmaf(e0[1], e1[2], -(e0[2] * e1[1]));
        T.nraw[1] = std::fmaf(e0[2], e1[0], -(e0[0] * e1[2]));
        T.nraw[2] = std::fmaf(e0[0], e1[1], -(e0[1] * e1[0]));
        const float l = std::sqrt(hdot3(T.nraw, T.nraw));
        T.area = 0.5f * l;
        for (int c = 0; c < 3; ++c) T.n[c] = l > 0.0f ? T.nraw[c] / l : 0.0f;
    }
    if (d.colors) {
        h.colors.assign(d.colors, d.colors + (size_t)d.n_verts * 6);
        for (float c : h.colors)
            if (c != 0.0f) h.emissive = true;
    }
    // edges: the first two incident triangles in index order, direction of the first
    std::vector<int32_t> edge_of;
    {
        struct Key { int a, b, t, k; };
        std::vector<Key> keys;
        keys.reserve((size_t)n * 3);
        for (int t = 0; t < n; ++t)
            for (int k = 0; k < 3; ++k) {
                const int a = d.tris[3 * (size_t)t + k], b = d.tris[3 * (size_t)t + (k + 1) % 3];
                keys.push_back(Key{std::min(a, b), std::max(a, b), t, k});
            }
        std::sort(keys.begin(), keys.end(), [](const Key &x, const Key &y) {
            if (x.a != y.a) return x.a < y.a;
            if (x.b != y.b) return x.b < y.b;
            if (x.t != y.t) return x.t < y.t;
            return x.k < y.k;
        });
        edge_of.assign((size_t)n * 3, -1);        // the edge record of side k of triangle t
        for (size_t i = 0; i < keys.size();) {
            size_t j = i;
            while (j < keys.size() && keys[j].a == keys[i].a && keys[j].b == keys[i].b) ++j;
            const int t0 = keys[i].t, k0 = keys[i].k;
            const int a = d.tris[3 * (size_t)t0 + k0], b = d.tris[3 * (size_t)t0 + (k0 + 1) % 3];
            if (a != b) {
                DevEdge3 E{};
                for (int c = 0; c < 3; ++c) { E.pa[c] = d.verts[3 * (size_t)a + c]; E.pb[c] = d.verts[3 * (size_t)b + c]; }
                E.t0 = t0;
                E.t1 = (j - i >= 2) ? keys[i + 1].t : -1;
                for (size_t q = i; q < j; ++q) edge_of[3 * (size_t)keys[q].t + keys[q].k] = (int32_t)h.edges.size();
                h.edges.push_back(E);
            }
            i = j;
        }
        h.n_edges = (int32_t)h.edges.size();
    }
    h.flatVerts.assign(d.tris, d.tris + (size_t)n * 3);
    // Morton order of the centroids, leaves of 4, implicit complete 4-ary tree (lbvh.h in 3-D)
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    {
        std::vector<uint32_t> code(n);
        for (int t = 0; t < n; ++t) {
            uint32_t q[3];
            for (int c = 0; c < 3; ++c) {
                const double s = hi[c] > lo[c] ? 1023.0 / ((double)hi[c] - lo[c]) : 0.0;
                q[c] = (uint32_t)std::min(1023.0, std::max(0.0, ((double)cen[3 * (size_t)t + c] - lo[c]) * s));
            }
            code[t] = part1by2(q[0]) | (part1by2(q[1]) << 1) | (part1by2(q[2]) << 2);
        }
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return code[a] < code[b]; });
    }
    const int n_leaves = (n + 3) / 4;
    int levels = 1, cap = 4;
    while (cap < n_leaves) { cap *= 4; ++levels; }
    h.levels = levels;
    h.first_leaf = (cap - 1) / 3;
    const size_t n_slots = (size_t)cap * 4;
    h.tri.assign(n_slots * 12, 1.0e18f);
    h.triOrig.assign(n_slots, kFarIndex);
    h.triVerts.assign(n_slots * 3, 0);
    std::vector<char> edge_listed(h.edges.size(), 0);
    h.slotEdges.assign(n_slots * 3 * 16, 0.0f);
    for (int k = 0; k < n; ++k) {
        const int o = order[k];
        for (int c = 0; c < 3; ++c) {
            const int32_t e = edge_of[3 * (size_t)o + c];
            if (e >= 0 && !edge_listed[e]) {
                edge_listed[e] = 1;
                const DevEdge3 &E = h.edges[e];
                float *r = &h.slotEdges[(3 * (size_t)k + c) * 16];
                for (int x = 0; x < 3; ++x) {
                    r[x] = E.pa[x]; r[4 + x] = E.pb[x];
                    r[8 + x] = h.flat[E.t0].n[x];
                    r[12 + x] = E.t1 >= 0 ? h.flat[E.t1].n[x] : 0.0f;
                }
                r[3] = E.t1 >= 0 ? 1.0f : 2.0f;
            }
        }
        const DevTri &T = h.flat[o];
        float *r = &h.tri[(size_t)k * 12];
        for (int c = 0; c < 3; ++c) { r[c] = T.p0[c]; r[4 + c] = T.p1[c]; r[8 + c] = T.p2[c]; }
        r[3] = r[7] = r[11] = 0.0f;
        h.triOrig[k] = o;
        for (int c = 0; c < 3; ++c) h.triVerts[3 * (size_t)k + c] = d.tris[3 * (size_t)o + c];
    }
    // boxes bottom-up, padded (pruning slack)
    float ext = 0.0f;
    for (int c = 0; c < 3; ++c) ext = std::max(ext, std::max(std::fabs(lo[c]), std::fabs(hi[c])));
    const float pad = ext * 0x1p-18f + 1e-30f;
    h.ext = ext;
    const int n_nodes = h.first_leaf + cap;
    std::vector<float> nb((size_t)n_nodes * 6);
    std::vector<char> empty(n_nodes, 1);
    for (int g = 0; g < n_nodes; ++g)
        for (int c = 0; c < 3; ++c) { nb[6 * (size_t)g + c] = INFINITY; nb[6 * (size_t)g + 3 + c] = -INFINITY; }
    for (int k = 0; k < n; ++k) {
        const int g = h.first_leaf + k / 4;
        const DevTri &T = h.flat[order[k]];
        for (int c = 0; c < 3; ++c) {
            nb[6 * (size_t)g + c] = std::min(nb[6 * (size_t)g + c], std::min(T.p0[c], std::min(T.p1[c], T.p2[c])));
            nb[6 * (size_t)g + 3 + c] = std::max(nb[6 * (size_t)g + 3 + c], std::max(T.p0[c], std::max(T.p1[c], T.p2[c])));
        }
        empty[g] = 0;
    }
    for (int g = h.first_leaf - 1; g >= 0; --g)
        for (int j = 1; j <= 4; ++j) {
            const int c4 = 4 * g + j;
            if (empty[c4]) continue;
            for (int c = 0; c < 3; ++c) {
                nb[6 * (size_t)g + c] = std::min(nb[6 * (size_t)g + c], nb[6 * (size_t)c4 + c]);
                nb[6 * (size_t)g + 3 + c] = std::max(nb[6 * (size_t)g + 3 + c], nb[6 * (size_t)c4 + 3 + c]);
            }
            empty[g] = 0;
        }
    if (h.emissive && n > WOST3_FLAT_MAX) {
        size_t prev_off = 0, prev_n = 0;
        for (int l = 0; l < 12; ++l) {
            const size_t run = (size_t)4 << (2 * l), n_runs = ((size_t)n + run - 1) / run;
            h.obox_off[l] = (int32_t)(h.obox.size() / 8);
            for (size_t r = 0; r < n_runs; ++r) {
                float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
                if (l == 0) {
                    for (size_t i = r * 4; i < std::min<size_t>(r * 4 + 4, (size_t)n); ++i)
                        for (int c = 0; c < 3; ++c) {
                            const DevTri &T = h.flat[i];
                            blo[c] = std::min(blo[c], std::min(T.p0[c], std::min(T.p1[c], T.p2[c])));
                            bhi[c] = std::max(bhi[c], std::max(T.p0[c], std::max(T.p1[c], T.p2[c])));
                        }
                    for (int c = 0; c < 3; ++c) { blo[c] -= pad; bhi[c] += pad; }
                } else {
                    for (size_t c4 = r * 4; c4 < std::min(r * 4 + 4, prev_n); ++c4)
                        for (int c = 0; c < 3; ++c) {
                            blo[c] = std::min(blo[c], h.obox[(prev_off + c4) * 8 + c]);
                            bhi[c] = std::max(bhi[c], h.obox[(prev_off + c4) * 8 + 4 + c]);
                        }
                }
                h.obox.insert(h.obox.end(), {blo[0], blo[1], blo[2], 0.0f, bhi[0], bhi[1], bhi[2], 0.0f});
            }
            prev_off = (size_t)h.obox_off[l];
            prev_n = n_runs;
            h.obox_levels = l + 1;
            if (n_runs <= 1) break;
        }
    }
    // inner nodes (levels 0 .. levels-1) store the boxes of their four children; a leaf (level == levels) stores the
    // boxes of its four triangles
    h.nodes.assign((size_t)n_nodes * 24, 0.0f);
    for (int g = h.first_leaf; g < n_nodes; ++g) {
        float *nd = &h.nodes[(size_t)g * 24];
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * (g - h.first_leaf) + j;
            for (int c = 0; c < 3; ++c) {
                if (k < n) {
                    const DevTri &T = h.flat[order[k]];
                    nd[4 * c + j] = std::min(T.p0[c], std::min(T.p1[c], T.p2[c])) - pad;
                    nd[12 + 4 * c + j] = std::max(T.p0[c], std::max(T.p1[c], T.p2[c])) + pad;
                } else {
                    nd[4 * c + j] = nd[12 + 4 * c + j] = 1.0e18f;
                }
            }
        }
    }
    for (int g = 0; g < h.first_leaf; ++g)
        for (int j = 0; j < 4; ++j) {
            const int c4 = 4 * g + 1 + j;
            float *nd = &h.nodes[(size_t)g * 24];
            for (int c = 0; c < 3; ++c) {
                nd[4 * c + j] = empty[c4] ? 1.0e18f : nb[6 * (size_t)c4 + c] - pad;
                nd[12 + 4 * c + j] = empty[c4] ? 1.0e18f : nb[6 * (size_t)c4 + 3 + c] + pad;
            }
        }
    // normal cones of the children of every inner node (cone3_may_hold_silhouette): the normals of both triangles of
    // every edge of the subtree's triangles, the end points of those edges; a boundary edge is always a silhouette
    {
        struct Acc { std::vector<double> nrm, pts; bool open = false; };
        std::vector<Acc> acc(n_nodes);
        for (int k = 0; k < n; ++k) {
            Acc &a = acc[h.first_leaf + k / 4];
            const int o = order[k];
            for (int c = 0; c < 3; ++c) {
                const int32_t e = edge_of[3 * (size_t)o + c];
                if (e < 0) continue;
                const DevEdge3 &E = h.edges[e];
                if (E.t1 < 0) a.open = true;
                for (int t : {E.t0, E.t1}) {
                    if (t < 0) continue;
                    const DevTri &T = h.flat[t];
                    if (hdot3(T.n, T.n) > 0.0f) a.nrm.insert(a.nrm.end(), {T.n[0], T.n[1], T.n[2]});
                }
                a.pts.insert(a.pts.end(), {E.pa[0], E.pa[1], E.pa[2], E.pb[0], E.pb[1], E.pb[2]});
            }
        }
        for (int g = h.first_leaf - 1; g >= 1; --g)
            for (int j = 1; j <= 4; ++j) {
                const Acc &c = acc[4 * g + j];
                acc[g].nrm.insert(acc[g].nrm.end(), c.nrm.begin(), c.nrm.end());
                acc[g].pts.insert(acc[g].pts.end(), c.pts.begin(), c.pts.end());
                acc[g].open = acc[g].open || c.open;
            }
        h.cones.assign((size_t)n_nodes * 24, 0.0f);
        for (int g = 0; g < h.first_leaf; ++g)
            for (int j = 0; j < 4; ++j) {
                const Acc &a = acc[4 * g + 1 + j];
                const float *nd = &h.nodes[(size_t)g * 24];
                float *cn = &h.cones[(size_t)g * 24];
                cn[12 + j] = -1.0f;                                             // cannot prune
                cn[0 + j] = 1.0f;
                if (a.open || a.nrm.empty()) continue;
                // the axis: the sum of the normals in 2^-36 fixed point, an integer sum, the same in any order (the device
                // build adds the sums of the children)
                long long sx[3] = {0, 0, 0};
                for (size_t i = 0; i < a.nrm.size(); i += 3)
                    for (int c = 0; c < 3; ++c) sx[c] += std::llrint(a.nrm[i + c] * kNormalFix);
                double ax[3];
                for (int c = 0; c < 3; ++c) ax[c] = (double)sx[c] / kNormalFix;
                const double al = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
                if (!(al > 1e-9 * (double)(a.nrm.size() / 3))) continue;
                for (int c = 0; c < 3; ++c) ax[c] /= al;
                double cmin = 1.0;
                for (size_t i = 0; i < a.nrm.size(); i += 3) {
                    const double l = std::sqrt(a.nrm[i] * a.nrm[i] + a.nrm[i + 1] * a.nrm[i + 1] + a.nrm[i + 2] * a.nrm[i + 2]);
                    cmin = std::min(cmin, (ax[0] * a.nrm[i] + ax[1] * a.nrm[i + 1] + ax[2] * a.nrm[i + 2]) / l);
                }
                // half angle = acos(cmin) + 10^-4, its cosine and sine by the addition theorem (square roots only: the same
                // bits on the host and on the device); cones wider than a right angle less 10^-3 cannot prune
                const double cc = std::max(-1.0, std::min(1.0, cmin)), ss = std::sqrt(std::max(0.0, 1.0 - cc * cc));
                const double ch = cc * kConePadCos - ss * kConePadSin, sh = ss * kConePadCos + cc * kConePadSin;
                if (ch <= 1.0e-3) continue;
                // the centre the kernel uses: the middle of the child's box, in the kernel's float arithmetic
                float cf[3];
                for (int c = 0; c < 3; ++c) cf[c] = 0.5f * (nd[4 * c + j] + nd[12 + 4 * c + j]);
                double rad = 0.0;
                for (size_t i = 0; i < a.pts.size(); i += 3)
                    rad = std::max(rad, std::sqrt((a.pts[i] - cf[0]) * (a.pts[i] - cf[0]) + (a.pts[i + 1] - cf[1]) * (a.pts[i + 1] - cf[1]) +
                                                  (a.pts[i + 2] - cf[2]) * (a.pts[i + 2] - cf[2])));
                cn[0 + j] = (float)ax[0]; cn[4 + j] = (float)ax[1]; cn[8 + j] = (float)ax[2];
                cn[12 + j] = (float)ch; cn[16 + j] = (float)sh;
                cn[20 + j] = (float)(rad * (1.0 + 1e-6) + (double)pad + 2.0 * (double)WOST_SIL_PRECISION);
            }
    }
    return 0;
}

// ---- the device build ----------------------------------------------------------------------------------------------------

namespace {

struct B3Meta {
    uint32_t lo[3], hi[3];      // the mesh bounds as ordered integers (f_enc)
    int32_t n_edges, bad;       // bad: a vertex index out of range
};
struct B3Shape {
    int32_t n, n_verts, levels, cap, first_leaf, n_nodes, vbits;
};
struct B3Sum {                  // per node: the normals of the edges below it
    long long s[3];             // their sum, 2^-36 fixed point
    int32_t cnt, open;          // how many; does a boundary edge lie below
};

// float <-> unsigned with the same order (atomicMin / atomicMax on the bounds)
__device__ __forceinline__ uint32_t f_enc(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f_dec(uint32_t e) { return __uint_as_float((e >> 31) ? (e & 0x7fffffffu) : ~e); }

__device__ __forceinline__ float b3_pad(const B3Meta *m)
{
    float ext = 0.0f;
    for (int c = 0; c < 3; ++c) ext = fmaxf(ext, fmaxf(fabsf(f_dec(m->lo[c])), fabsf(f_dec(m->hi[c]))));
    return ext * 0x1p-18f + 1e-30f;
}
__device__ __forceinline__ float ddot3(const float *a, const float *b) { return fmaf(a[0], b[0], fmaf(a[1], b[1], a[2] * b[2])); }
__device__ __forceinline__ uint32_t d_part1by2(uint32_t x)
{
    x &= 0x3ff;
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
// the first slot below heap node g of level l (4^(levels - l) leaves of four slots each)
__device__ __forceinline__ long long b3_first_slot(const B3Shape &S, int g, int l)
{
    long long first = 0, count = 1;
    for (int i = 0; i < l; ++i) { first += count; count *= 4; }
    long long span = 4;
    for (int i = l; i < S.levels; ++i) span *= 4;
    return (g - first) * span;
}
__device__ __forceinline__ double shfl_xor_f64(double v, int m)
{
    const int lo = __shfl_xor(__double2loint(v), m), hi = __shfl_xor(__double2hiint(v), m);
    return __hiloint2double(hi, lo);
}

__global__ void b3_init_kernel(B3Meta *meta)
{
    for (int c = 0; c < 3; ++c) { meta->lo[c] = 0xff800000u; meta->hi[c] = 0x007fffffu; }     // +inf, -inf
    meta->n_edges = 0;
    meta->bad = 0;
}

// triangle records, centroids, the bounds
__global__ __launch_bounds__(256) void b3_tri_kernel(const float *verts, const int32_t *tris, int n, int n_verts, DevTri *flat, float *cen,
                                                     B3Meta *meta)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (t < n) {
        const int i0 = tris[3 * (size_t)t], i1 = tris[3 * (size_t)t + 1], i2 = tris[3 * (size_t)t + 2];
        if ((unsigned)i0 >= (unsigned)n_verts || (unsigned)i1 >= (unsigned)n_verts || (unsigned)i2 >= (unsigned)n_verts) {
            meta->bad = 1;
        } else {
            DevTri T;
            for (int c = 0; c < 3; ++c) {
                T.p0[c] = verts[3 * (size_t)i0 + c]; T.p1[c] = verts[3 * (size_t)i1 + c]; T.p2[c] = verts[3 * (size_t)i2 + c];
                lo[c] = fminf(T.p0[c], fminf(T.p1[c], T.p2[c]));
                hi[c] = fmaxf(T.p0[c], fmaxf(T.p1[c], T.p2[c]));
                cen[3 * (size_t)t + c] = (float)(((double)T.p0[c] + T.p1[c] + T.p2[c]) / 3.0);
            }
            float e0[3], e1[3];
            for (int c = 0; c < 3; ++c) { e0[c] = T.p1[c] - T.p0[c]; e1[c] = T.p2[c] - T.p0[c]; }
            T.nraw[0] = fmaf(e0[1], e1[2], -(e0[2] * e1[1]));
            T.nraw[1] = fmaf(e0[2], e1[0], -(e0[0] * e1[2]));
            T.nraw[2] = fmaf(e0[0], e1[1], -(e0[1] * e1[0]));
            const float l = sqrtf(ddot3(T.nraw, T.nraw));
            T.area = 0.5f * l;
            for (int c = 0; c < 3; ++c) T.n[c] = l > 0.0f ? T.nraw[c] / l : 0.0f;
            flat[t] = T;
        }
    }
    for (int c = 0; c < 3; ++c) {
        float a = lo[c], b = hi[c];
        for (int m = 32; m >= 1; m >>= 1) { a = fminf(a, __shfl_xor(a, m)); b = fmaxf(b, __shfl_xor(b, m)); }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&meta->lo[c], f_enc(a));
            atomicMax(&meta->hi[c], f_enc(b));
        }
    }
}

__global__ __launch_bounds__(256) void b3_morton_kernel(const float *cen, int n, const B3Meta *meta, uint32_t *code, int32_t *idx)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || meta->bad) return;
    uint32_t q[3];
    for (int c = 0; c < 3; ++c) {
        const float lo = f_dec(meta->lo[c]), hi = f_dec(meta->hi[c]);
        const double s = hi > lo ? 1023.0 / ((double)hi - lo) : 0.0;
        q[c] = (uint32_t)fmin(1023.0, fmax(0.0, ((double)cen[3 * (size_t)t + c] - lo) * s));
    }
    code[t] = d_part1by2(q[0]) | (d_part1by2(q[1]) << 1) | (d_part1by2(q[2]) << 2);
    idx[t] = t;
}

// one key per triangle side: (smaller vertex, larger vertex); sorted stably, the sides of an edge stand together in
// triangle order
__global__ __launch_bounds__(256) void b3_edge_key_kernel(const int32_t *tris, int n3, int vbits, const B3Meta *meta, uint64_t *keys, int32_t *vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    uint64_t key = 0;
    if (!meta->bad) {
        const int t = i / 3, k = i - 3 * t;
        const int a = tris[3 * (size_t)t + k], b = tris[3 * (size_t)t + (k + 1) % 3];
        key = ((uint64_t)(uint32_t)min(a, b) << vbits) | (uint32_t)max(a, b);
    }
    keys[i] = key;
    vals[i] = i;
}

__device__ __forceinline__ bool b3_edge_starts(const uint64_t *keys, int i, int vbits)
{
    const uint64_t k = keys[i];
    return (i == 0 || keys[i - 1] != k) && (k >> vbits) != (k & ((1ull << vbits) - 1));
}

__global__ __launch_bounds__(256) void b3_edge_flag_kernel(const uint64_t *keys, int n3, int vbits, int32_t *flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) flags[i] = b3_edge_starts(keys, i, vbits) ? 1 : 0;
}

// edge records: the first two incident triangles in index order, direction of the first; edge_of[3 t + k] of every side
__global__ __launch_bounds__(256) void b3_edge_record_kernel(const uint64_t *keys, const int32_t *vals, const int32_t *eid, const int32_t *tris,
                                                             const float *verts, int n3, int vbits, DevEdge3 *edges, int32_t *edge_of, B3Meta *meta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3 || meta->bad) return;
    const bool starts = b3_edge_starts(keys, i, vbits);
    if (i == n3 - 1) meta->n_edges = eid[i] + (starts ? 1 : 0);
    const uint64_t key = keys[i];
    if (i > 0 && keys[i - 1] == key) return;                 // not the first side of its group
    const int e = starts ? eid[i] : -1;                       // a side between a vertex and itself has no edge
    if (e >= 0) {
        const int val = vals[i], t0 = val / 3, k0 = val - 3 * t0;
        const int a = tris[3 * (size_t)t0 + k0], b = tris[3 * (size_t)t0 + (k0 + 1) % 3];
        DevEdge3 E;
        for (int c = 0; c < 3; ++c) { E.pa[c] = verts[3 * (size_t)a + c]; E.pb[c] = verts[3 * (size_t)b + c]; }
        E.t0 = t0;
        E.t1 = (i + 1 < n3 && keys[i + 1] == key) ? vals[i + 1] / 3 : -1;
        edges[e] = E;
    }
    for (int q = i; q < n3 && keys[q] == key; ++q) edge_of[vals[q]] = e;
}

__global__ __launch_bounds__(256) void b3_fill_kernel(uint32_t *p, size_t count, uint32_t value)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) p[i] = value;
}

// the triangles in leaf order; per edge the first side (in slot order) that lists it
__global__ __launch_bounds__(256) void b3_slot_kernel(const int32_t *order, const DevTri *flat, const int32_t *tris, const int32_t *edge_of, int n,
                                                      const B3Meta *meta, float *tri, int32_t *triOrig, int32_t *triVerts, int32_t *slotOfOrig,
                                                      uint32_t *firstref)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n || meta->bad) return;
    const int o = order[k];
    const DevTri T = flat[o];
    float *r = tri + (size_t)k * 12;
    for (int c = 0; c < 3; ++c) { r[c] = T.p0[c]; r[4 + c] = T.p1[c]; r[8 + c] = T.p2[c]; }
    r[3] = r[7] = r[11] = 0.0f;
    triOrig[k] = o;
    slotOfOrig[o] = k;
    for (int c = 0; c < 3; ++c) {
        triVerts[3 * (size_t)k + c] = tris[3 * (size_t)o + c];
        const int e = edge_of[3 * (size_t)o + c];
        if (e >= 0) atomicMin(&firstref[e], (uint32_t)(3 * k + c));
    }
}

__global__ __launch_bounds__(256) void b3_slot_edge_kernel(const int32_t *order, const int32_t *edge_of, const uint32_t *firstref, const DevEdge3 *edges,
                                                           const DevTri *flat, int n, const B3Meta *meta, float *slotEdges)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n || meta->bad) return;
    const int k = i / 3, c = i - 3 * k, o = order[k];
    const int e = edge_of[3 * (size_t)o + c];
    if (e < 0 || firstref[e] != (uint32_t)i) return;
    const DevEdge3 E = edges[e];
    float *r = slotEdges + (size_t)i * 16;
    for (int x = 0; x < 3; ++x) {
        r[x] = E.pa[x]; r[4 + x] = E.pb[x];
        r[8 + x] = flat[E.t0].n[x];
        r[12 + x] = E.t1 >= 0 ? flat[E.t1].n[x] : 0.0f;
    }
    r[3] = E.t1 >= 0 ? 1.0f : 2.0f;
}

// what the edges of the triangle in slot k add to the sums of its leaf
__device__ __forceinline__ void b3_slot_sums(const int32_t *edge_of, const DevEdge3 *edges, const DevTri *flat, int o, B3Sum &a)
{
    for (int c = 0; c < 3; ++c) {
        const int e = edge_of[3 * (size_t)o + c];
        if (e < 0) continue;
        const int t0 = edges[e].t0, t1 = edges[e].t1;
        if (t1 < 0) a.open = 1;
        for (int w = 0; w < 2; ++w) {
            const int t = w ? t1 : t0;
            if (t < 0) continue;
            const float *nn = flat[t].n;
            if (ddot3(nn, nn) > 0.0f) {
                for (int x = 0; x < 3; ++x) a.s[x] += __double2ll_rn((double)nn[x] * kNormalFix);
                a.cnt += 1;
            }
        }
    }
}

// leaves: the boxes of their four triangles, their own box, their sums
__global__ __launch_bounds__(256) void b3_leaf_kernel(B3Shape S, const int32_t *order, const DevTri *flat, const int32_t *edge_of, const DevEdge3 *edges,
                                                      const B3Meta *meta, float *nb, B3Sum *nsum, float *nodes)
{
    const int L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= S.cap || meta->bad) return;
    const int g = S.first_leaf + L;
    const float pad = b3_pad(meta);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    B3Sum a{};
    float *nd = nodes + (size_t)g * 24;
    for (int j = 0; j < 4; ++j) {
        const long long k = 4ll * L + j;
        if (k < S.n) {
            const int o = order[k];
            const DevTri T = flat[o];
            for (int c = 0; c < 3; ++c) {
                const float mn = fminf(T.p0[c], fminf(T.p1[c], T.p2[c])), mx = fmaxf(T.p0[c], fmaxf(T.p1[c], T.p2[c]));
                nd[4 * c + j] = mn - pad;
                nd[12 + 4 * c + j] = mx + pad;
                lo[c] = fminf(lo[c], mn);
                hi[c] = fmaxf(hi[c], mx);
            }
            b3_slot_sums(edge_of, edges, flat, o, a);
        } else {
            for (int c = 0; c < 3; ++c) nd[4 * c + j] = nd[12 + 4 * c + j] = 1.0e18f;
        }
    }
    for (int c = 0; c < 3; ++c) { nb[6 * (size_t)g + c] = lo[c]; nb[6 * (size_t)g + 3 + c] = hi[c]; }
    nsum[g] = a;
}

// one inner level: a node stores the padded boxes of its four children, keeps their union and the sum of their sums
__global__ __launch_bounds__(256) void b3_inner_kernel(B3Shape S, int level, int level_first, int level_count, const B3Meta *meta, float *nb,
                                                       B3Sum *nsum, float *nodes)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= level_count || meta->bad) return;
    const int g = level_first + p;
    const float pad = b3_pad(meta);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    B3Sum a{};
    float *nd = nodes + (size_t)g * 24;
    for (int j = 0; j < 4; ++j) {
        const int c4 = 4 * g + 1 + j;
        const bool empty = b3_first_slot(S, c4, level + 1) >= S.n;
        for (int c = 0; c < 3; ++c) {
            const float l = nb[6 * (size_t)c4 + c], h = nb[6 * (size_t)c4 + 3 + c];
            nd[4 * c + j] = empty ? 1.0e18f : l - pad;
            nd[12 + 4 * c + j] = empty ? 1.0e18f : h + pad;
            if (!empty) { lo[c] = fminf(lo[c], l); hi[c] = fmaxf(hi[c], h); }
        }
        if (!empty) {
            const B3Sum b = nsum[c4];
            for (int x = 0; x < 3; ++x) a.s[x] += b.s[x];
            a.cnt += b.cnt;
            a.open |= b.open;
        }
    }
    for (int c = 0; c < 3; ++c) { nb[6 * (size_t)g + c] = lo[c]; nb[6 * (size_t)g + 3 + c] = hi[c]; }
    nsum[g] = a;
}

// the normal cone of every child of every inner node, one wave per child: axis from the node's sum, the widest normal and
// the farthest edge end point by a sweep of the child's slots (minima and maxima: any order)
__global__ __launch_bounds__(256) void b3_cone_kernel(B3Shape S, const int32_t *order, const int32_t *edge_of, const DevEdge3 *edges, const DevTri *flat,
                                                      const B3Meta *meta, const B3Sum *nsum, const float *nodes, float *cones)
{
    const long long w = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= 4ll * S.first_leaf || meta->bad) return;
    const int g = (int)(w >> 2), j = (int)(w & 3), c4 = 4 * g + 1 + j;
    float *cn = cones + (size_t)g * 24;
    int level = 0;
    for (long long first = 0, count = 1; c4 >= first + count; first += count, count *= 4) ++level;
    const long long s0 = b3_first_slot(S, c4, level);
    long long span = 4;
    for (int i = level; i < S.levels; ++i) span *= 4;
    const long long s1 = min(s0 + span, (long long)S.n);
    const B3Sum a = nsum[c4];
    bool prunes = s0 < S.n && !a.open && a.cnt > 0;
    double ax[3] = {0.0, 0.0, 0.0};
    if (prunes) {
        for (int c = 0; c < 3; ++c) ax[c] = (double)a.s[c] / kNormalFix;
        const double al = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
        prunes = al > 1e-9 * (double)a.cnt;
        if (prunes)
            for (int c = 0; c < 3; ++c) ax[c] /= al;
    }
    if (!prunes) {
        if (lane == 0) { cn[12 + j] = -1.0f; cn[0 + j] = 1.0f; }
        return;
    }
    const float pad = b3_pad(meta);
    const float *nd = nodes + (size_t)g * 24;
    float cf[3];
    for (int c = 0; c < 3; ++c) cf[c] = 0.5f * (nd[4 * c + j] + nd[12 + 4 * c + j]);
    double cmin = 1.0, rad = 0.0;
    for (long long k = s0 + lane; k < s1; k += 64) {
        const int o = order[k];
        for (int c = 0; c < 3; ++c) {
            const int e = edge_of[3 * (size_t)o + c];
            if (e < 0) continue;
            const DevEdge3 E = edges[e];
            for (int q = 0; q < 2; ++q) {
                const int t = q ? E.t1 : E.t0;
                if (t < 0) continue;
                const float *nn = flat[t].n;
                if (!(ddot3(nn, nn) > 0.0f)) continue;
                const double n0 = nn[0], n1 = nn[1], n2 = nn[2];
                const double l = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
                cmin = fmin(cmin, (ax[0] * n0 + ax[1] * n1 + ax[2] * n2) / l);
            }
            for (int q = 0; q < 2; ++q) {
                const float *pp = q ? E.pb : E.pa;
                const double d0 = (double)pp[0] - cf[0], d1 = (double)pp[1] - cf[1], d2 = (double)pp[2] - cf[2];
                rad = fmax(rad, sqrt(d0 * d0 + d1 * d1 + d2 * d2));
            }
        }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        cmin = fmin(cmin, shfl_xor_f64(cmin, m));
        rad = fmax(rad, shfl_xor_f64(rad, m));
    }
    if (lane != 0) return;
    const double cc = fmax(-1.0, fmin(1.0, cmin)), ss = sqrt(fmax(0.0, 1.0 - cc * cc));
    const double ch = cc * kConePadCos - ss * kConePadSin, sh = ss * kConePadCos + cc * kConePadSin;
    if (ch <= 1.0e-3) {
        cn[12 + j] = -1.0f;
        cn[0 + j] = 1.0f;
        return;
    }
    cn[0 + j] = (float)ax[0]; cn[4 + j] = (float)ax[1]; cn[8 + j] = (float)ax[2];
    cn[12 + j] = (float)ch; cn[16 + j] = (float)sh;
    cn[20 + j] = (float)(rad * (1.0 + 1e-6) + (double)pad + 2.0 * (double)WOST_SIL_PRECISION);
}

// run boxes over consecutive original triangle indices (sample_in_sphere3_tree) and the sampler's compact copies
__global__ __launch_bounds__(256) void b3_obox0_kernel(const DevTri *flat, int n, long long n_runs, const B3Meta *meta, float *obox)
{
    const long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (r >= n_runs || meta->bad) return;
    const float pad = b3_pad(meta);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long i = r * 4; i < min(r * 4 + 4, (long long)n); ++i) {
        const DevTri T = flat[i];
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], fminf(T.p0[c], fminf(T.p1[c], T.p2[c])));
            hi[c] = fmaxf(hi[c], fmaxf(T.p0[c], fmaxf(T.p1[c], T.p2[c])));
        }
    }
    float *o = obox + r * 8;
    for (int c = 0; c < 3; ++c) { o[c] = lo[c] - pad; o[4 + c] = hi[c] + pad; }
    o[3] = o[7] = 0.0f;
}
__global__ __launch_bounds__(256) void b3_obox_up_kernel(float *obox, long long prev_off, long long prev_n, long long off, long long n_runs,
                                                         const B3Meta *meta)
{
    const long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (r >= n_runs || meta->bad) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long c4 = r * 4; c4 < min(r * 4 + 4, prev_n); ++c4)
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], obox[(prev_off + c4) * 8 + c]);
            hi[c] = fmaxf(hi[c], obox[(prev_off + c4) * 8 + 4 + c]);
        }
    float *o = obox + (off + r) * 8;
    for (int c = 0; c < 3; ++c) { o[c] = lo[c]; o[4 + c] = hi[c]; }
    o[3] = o[7] = 0.0f;
}
__global__ __launch_bounds__(256) void b3_samp_kernel(const DevTri *flat, int n, long long n4, const B3Meta *meta, float *areas, float *samp)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n4 || meta->bad) return;
    float *r = samp + i * 12;
    for (int x = 0; x < 12; ++x) r[x] = 1.0e18f;
    float area = 0.0f;
    if (i < n) {
        const DevTri T = flat[i];
        area = T.area;
        for (int c = 0; c < 3; ++c) { r[c] = T.p0[c]; r[4 + c] = T.p1[c]; r[8 + c] = T.p2[c]; }
    }
    areas[i] = area;
}

struct B3Buffer {               // one allocation, carved in 256-byte steps
    char *base = nullptr;
    size_t size = 0;
    size_t take(size_t bytes)
    {
        const size_t o = size;
        size += (bytes + 255) & ~(size_t)255;
        return o;
    }
    template <class T>
    T *at(size_t off) const { return reinterpret_cast<T *>(base + off); }
    ~B3Buffer()
    {
        if (base) (void)hipFree(base);
    }
};

}  // namespace

#define B3_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return set_error(WOST_ERR_DEVICE, std::string("mesh build: ") + #expr + ": " + hipGetErrorString(e_)); \
    } while (0)

static inline dim3 b3_grid(long long count) { return dim3((unsigned)std::max<long long>(1, (count + 255) / 256)); }

// the mesh of `d` built on the current device into s.view; everything the view points at is one allocation (s.allocs)
static int build_mesh3_device(const wost3_mesh_desc &d, DeviceMesh3 &s)
{
    DevMesh3 &v = s.view;
    v = DevMesh3{};
    v.n_tris = d.n_tris;
    if (d.n_tris == 0) return WOST_OK;
    if (!d.verts || !d.tris || d.n_verts <= 0) return set_error(WOST_ERR_INVALID, "mesh: triangle index out of range or null arrays");
    if (d.n_tris > (1 << 27)) return set_error(WOST_ERR_UNSUPPORTED, "mesh: more than 2^27 triangles");
    const int n = d.n_tris, nv = d.n_verts, n3 = 3 * n;
    B3Shape S{};
    S.n = n; S.n_verts = nv;
    const int n_leaves = (n + 3) / 4;
    S.levels = 1; S.cap = 4;
    while (S.cap < n_leaves) { S.cap *= 4; ++S.levels; }
    S.first_leaf = (S.cap - 1) / 3;
    S.n_nodes = S.first_leaf + S.cap;
    S.vbits = 1;
    while ((1ll << S.vbits) < nv) ++S.vbits;
    const size_t n_slots = (size_t)S.cap * 4;
    bool emissive = false;
    if (d.colors)
        for (size_t i = 0; i < (size_t)nv * 6 && !emissive; ++i) emissive = d.colors[i] != 0.0f;
    // the run boxes of the emissive sampler: sizes are a function of n
    long long obox_off[12] = {0}, obox_runs[12] = {0}, obox_total = 0;
    int obox_levels = 0;
    if (emissive && n > WOST3_FLAT_MAX)
        for (int l = 0; l < 12; ++l) {
            const long long run = 4ll << (2 * l), n_runs = (n + run - 1) / run;
            obox_off[l] = obox_total; obox_runs[l] = n_runs; obox_total += n_runs;
            obox_levels = l + 1;
            if (n_runs <= 1) break;
        }
    const size_t n4 = ((size_t)n + 3) / 4 * 4;

    B3Buffer out, tmp;
    const size_t o_nodes = out.take((size_t)S.n_nodes * 24 * 4), o_tri = out.take(n_slots * 12 * 4), o_triOrig = out.take(n_slots * 4),
                 o_slotOf = out.take((size_t)n * 4), o_triVerts = out.take(n_slots * 3 * 4), o_colors = out.take(d.colors ? (size_t)nv * 6 * 4 : 0),
                 o_flat = out.take((size_t)n * sizeof(DevTri)), o_flatVerts = out.take((size_t)n3 * 4), o_edges = out.take((size_t)n3 * sizeof(DevEdge3)),
                 o_cones = out.take((size_t)S.n_nodes * 24 * 4), o_slotEdges = out.take(n_slots * 48 * 4), o_obox = out.take((size_t)obox_total * 32),
                 o_areas = out.take(obox_total ? n4 * 4 : 0), o_samp = out.take(obox_total ? n4 * 48 : 0);
    size_t sort_a = 0, sort_b = 0, scan_c = 0;
    B3_TRY(rocprim::radix_sort_pairs(nullptr, sort_a, (uint32_t *)nullptr, (uint32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, 0u, 30u));
    B3_TRY(rocprim::radix_sort_pairs(nullptr, sort_b, (uint64_t *)nullptr, (uint64_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (size_t)n3, 0u,
                                     (unsigned)(2 * S.vbits)));
    B3_TRY(rocprim::exclusive_scan(nullptr, scan_c, (int32_t *)nullptr, (int32_t *)nullptr, 0, (size_t)n3, rocprim::plus<int32_t>()));
    const size_t t_meta = tmp.take(sizeof(B3Meta)), t_verts = tmp.take((size_t)nv * 12), t_cen = tmp.take((size_t)n * 12), t_code = tmp.take((size_t)n * 4),
                 t_code2 = tmp.take((size_t)n * 4), t_idx = tmp.take((size_t)n * 4), t_order = tmp.take((size_t)n * 4), t_keys = tmp.take((size_t)n3 * 8),
                 t_keys2 = tmp.take((size_t)n3 * 8), t_vals = tmp.take((size_t)n3 * 4), t_vals2 = tmp.take((size_t)n3 * 4), t_flags = tmp.take((size_t)n3 * 4),
                 t_eid = tmp.take((size_t)n3 * 4), t_edge_of = tmp.take((size_t)n3 * 4), t_first = tmp.take((size_t)n3 * 4),
                 t_nb = tmp.take((size_t)S.n_nodes * 24), t_nsum = tmp.take((size_t)S.n_nodes * sizeof(B3Sum)),
                 t_rp = tmp.take(std::max(sort_a, std::max(sort_b, scan_c)));
    B3_TRY(hipMalloc((void **)&out.base, out.size));
    B3_TRY(hipMalloc((void **)&tmp.base, tmp.size));
    hipStream_t st = nullptr;
    B3Meta *meta = tmp.at<B3Meta>(t_meta);
    float *verts = tmp.at<float>(t_verts);
    int32_t *tris = out.at<int32_t>(o_flatVerts);
    DevTri *flat = out.at<DevTri>(o_flat);
    DevEdge3 *edges = out.at<DevEdge3>(o_edges);
    int32_t *order = tmp.at<int32_t>(t_order), *edge_of = tmp.at<int32_t>(t_edge_of);
    B3_TRY(hipMemcpyAsync(verts, d.verts, (size_t)nv * 12, hipMemcpyHostToDevice, st));
    B3_TRY(hipMemcpyAsync(tris, d.tris, (size_t)n3 * 4, hipMemcpyHostToDevice, st));
    if (d.colors) B3_TRY(hipMemcpyAsync(out.at<float>(o_colors), d.colors, (size_t)nv * 24, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(b3_init_kernel, dim3(1), dim3(1), 0, st, meta);
    hipLaunchKernelGGL(b3_tri_kernel, b3_grid(n), dim3(256), 0, st, verts, tris, n, nv, flat, tmp.at<float>(t_cen), meta);
    hipLaunchKernelGGL(b3_morton_kernel, b3_grid(n), dim3(256), 0, st, tmp.at<float>(t_cen), n, meta, tmp.at<uint32_t>(t_code), tmp.at<int32_t>(t_idx));
    size_t rp_bytes = sort_a;
    B3_TRY(rocprim::radix_sort_pairs(tmp.at<void>(t_rp), rp_bytes, tmp.at<uint32_t>(t_code), tmp.at<uint32_t>(t_code2), tmp.at<int32_t>(t_idx), order,
                                     (size_t)n, 0u, 30u, st));
    // edges
    hipLaunchKernelGGL(b3_edge_key_kernel, b3_grid(n3), dim3(256), 0, st, tris, n3, S.vbits, meta, tmp.at<uint64_t>(t_keys), tmp.at<int32_t>(t_vals));
    rp_bytes = sort_b;
    B3_TRY(rocprim::radix_sort_pairs(tmp.at<void>(t_rp), rp_bytes, tmp.at<uint64_t>(t_keys), tmp.at<uint64_t>(t_keys2), tmp.at<int32_t>(t_vals),
                                     tmp.at<int32_t>(t_vals2), (size_t)n3, 0u, (unsigned)(2 * S.vbits), st));
    hipLaunchKernelGGL(b3_edge_flag_kernel, b3_grid(n3), dim3(256), 0, st, tmp.at<uint64_t>(t_keys2), n3, S.vbits, tmp.at<int32_t>(t_flags));
    rp_bytes = scan_c;
    B3_TRY(rocprim::exclusive_scan(tmp.at<void>(t_rp), rp_bytes, tmp.at<int32_t>(t_flags), tmp.at<int32_t>(t_eid), 0, (size_t)n3, rocprim::plus<int32_t>(), st));
    B3_TRY(hipMemsetAsync(edges, 0, (size_t)n3 * sizeof(DevEdge3), st));
    hipLaunchKernelGGL(b3_edge_record_kernel, b3_grid(n3), dim3(256), 0, st, tmp.at<uint64_t>(t_keys2), tmp.at<int32_t>(t_vals2), tmp.at<int32_t>(t_eid), tris,
                       verts, n3, S.vbits, edges, edge_of, meta);
    // slots
    float *tri = out.at<float>(o_tri);
    hipLaunchKernelGGL(b3_fill_kernel, dim3(1024), dim3(256), 0, st, reinterpret_cast<uint32_t *>(tri) + (size_t)n * 12, (n_slots - n) * 12, 0x5d5e0b6bu);
    hipLaunchKernelGGL(b3_fill_kernel, dim3(256), dim3(256), 0, st, out.at<uint32_t>(o_triOrig) + n, n_slots - n, (uint32_t)kFarIndex);
    B3_TRY(hipMemsetAsync(out.at<char>(o_triVerts), 0, n_slots * 12, st));
    B3_TRY(hipMemsetAsync(out.at<char>(o_slotEdges), 0, n_slots * 192, st));
    B3_TRY(hipMemsetAsync(out.at<char>(o_cones), 0, (size_t)S.n_nodes * 96, st));
    B3_TRY(hipMemsetAsync(tmp.at<char>(t_first), 0xff, (size_t)n3 * 4, st));
    hipLaunchKernelGGL(b3_slot_kernel, b3_grid(n), dim3(256), 0, st, order, flat, tris, edge_of, n, meta, tri, out.at<int32_t>(o_triOrig),
                       out.at<int32_t>(o_triVerts), out.at<int32_t>(o_slotOf), tmp.at<uint32_t>(t_first));
    hipLaunchKernelGGL(b3_slot_edge_kernel, b3_grid(n3), dim3(256), 0, st, order, edge_of, tmp.at<uint32_t>(t_first), edges, flat, n, meta,
                       out.at<float>(o_slotEdges));
    // boxes and sums bottom-up, cones
    float *nodes = out.at<float>(o_nodes), *nb = tmp.at<float>(t_nb);
    B3Sum *nsum = tmp.at<B3Sum>(t_nsum);
    hipLaunchKernelGGL(b3_leaf_kernel, b3_grid(S.cap), dim3(256), 0, st, S, order, flat, edge_of, edges, meta, nb, nsum, nodes);
    {
        std::vector<int> first(S.levels + 1), count(S.levels + 1);
        int f = 0, c = 1;
        for (int l = 0; l <= S.levels; ++l) { first[l] = f; count[l] = c; f += c; c *= 4; }
        for (int l = S.levels - 1; l >= 0; --l)
            hipLaunchKernelGGL(b3_inner_kernel, b3_grid(count[l]), dim3(256), 0, st, S, l, first[l], count[l], meta, nb, nsum, nodes);
    }
    hipLaunchKernelGGL(b3_cone_kernel, b3_grid(256ll * S.first_leaf), dim3(256), 0, st, S, order, edge_of, edges, flat, meta, nsum, nodes,
                       out.at<float>(o_cones));
    if (obox_total) {
        float *obox = out.at<float>(o_obox);
        hipLaunchKernelGGL(b3_obox0_kernel, b3_grid(obox_runs[0]), dim3(256), 0, st, flat, n, obox_runs[0], meta, obox);
        for (int l = 1; l < obox_levels; ++l)
            hipLaunchKernelGGL(b3_obox_up_kernel, b3_grid(obox_runs[l]), dim3(256), 0, st, obox, obox_off[l - 1], obox_runs[l - 1], obox_off[l], obox_runs[l], meta);
        hipLaunchKernelGGL(b3_samp_kernel, b3_grid((long long)n4), dim3(256), 0, st, flat, n, (long long)n4, meta, out.at<float>(o_areas), out.at<float>(o_samp));
    }
    B3_TRY(hipGetLastError());
    B3Meta hm{};
    B3_TRY(hipMemcpy(&hm, meta, sizeof(hm), hipMemcpyDeviceToHost));         // the one wait of the build
    if (hm.bad) return set_error(WOST_ERR_INVALID, "mesh: triangle index out of range or null arrays");
    float ext = 0.0f;
    for (int c = 0; c < 3; ++c) {
        auto dec = [](uint32_t e) { const uint32_t u = (e >> 31) ? (e & 0x7fffffffu) : ~e; float f; std::memcpy(&f, &u, 4); return f; };
        ext = std::max(ext, std::max(std::fabs(dec(hm.lo[c])), std::fabs(dec(hm.hi[c]))));
    }
    v.n_edges = hm.n_edges; v.levels = S.levels; v.first_leaf = S.first_leaf; v.emissive = emissive ? 1 : 0;
    v.huge2 = 4096.0f * ext * ext;
    v.nodes = out.at<float4>(o_nodes); v.tri = out.at<float4>(o_tri); v.triOrig = out.at<int32_t>(o_triOrig); v.slotOfOrig = out.at<int32_t>(o_slotOf);
    v.triVerts = out.at<int32_t>(o_triVerts); v.colors = d.colors ? out.at<float>(o_colors) : nullptr; v.flat = flat; v.flatVerts = tris;
    v.edges = hm.n_edges > 0 ? edges : nullptr; v.cones = out.at<float4>(o_cones); v.slotEdges = out.at<float4>(o_slotEdges);
    if (obox_total) { v.obox = out.at<float4>(o_obox); v.areas = out.at<float>(o_areas); v.sampTri = out.at<float4>(o_samp); }
    for (int l = 0; l < 12; ++l) v.obox_off[l] = (int32_t)obox_off[l];
    v.obox_levels = obox_levels;
    s.allocs.push_back(out.base);
    out.base = nullptr;                     // owned by the mesh from here
    return WOST_OK;
}

// the host builder's mesh, uploaded array by array
static int upload_mesh3_host(const wost3_mesh_desc &d, DeviceMesh3 &s)
{
    if (build_mesh3(d, &s.host) != 0) return set_error(WOST_ERR_INVALID, "mesh: triangle index out of range or null arrays");
    const HostMesh3 &h = s.host;
    DevMesh3 &v = s.view;
    v = DevMesh3{};
    v.n_tris = h.n_tris;
    if (h.n_tris == 0) return WOST_OK;
    v.n_edges = h.n_edges; v.levels = h.levels; v.first_leaf = h.first_leaf; v.emissive = h.emissive ? 1 : 0;
    v.huge2 = 4096.0f * h.ext * h.ext;        // 64 extents
    W3_TRY(upload3(s.allocs, reinterpret_cast<const float4 *>(h.nodes.data()), h.nodes.size() / 4, &v.nodes));
    W3_TRY(upload3(s.allocs, reinterpret_cast<const float4 *>(h.tri.data()), h.tri.size() / 4, &v.tri));
    W3_TRY(upload3(s.allocs, h.triOrig.data(), h.triOrig.size(), &v.triOrig));
    {
        std::vector<int32_t> inv((size_t)std::max(h.n_tris, 1), 0);
        for (size_t k = 0; k < h.triOrig.size(); ++k)
            if (h.triOrig[k] != kFarIndex) inv[(size_t)h.triOrig[k]] = (int32_t)k;
        W3_TRY(upload3(s.allocs, inv.data(), inv.size(), &v.slotOfOrig));
    }
    W3_TRY(upload3(s.allocs, h.triVerts.data(), h.triVerts.size(), &v.triVerts));
    W3_TRY(upload3(s.allocs, h.colors.data(), h.colors.size(), &v.colors));
    W3_TRY(upload3(s.allocs, h.flat.data(), h.flat.size(), &v.flat));
    W3_TRY(upload3(s.allocs, h.edges.data(), h.edges.size(), &v.edges));
    W3_TRY(upload3(s.allocs, h.flatVerts.data(), h.flatVerts.size(), &v.flatVerts));
    W3_TRY(upload3(s.allocs, reinterpret_cast<const float4 *>(h.cones.data()), h.cones.size() / 4, &v.cones));
    W3_TRY(upload3(s.allocs, reinterpret_cast<const float4 *>(h.slotEdges.data()), h.slotEdges.size() / 4, &v.slotEdges));
    W3_TRY(upload3(s.allocs, reinterpret_cast<const float4 *>(h.obox.data()), h.obox.size() / 4, &v.obox));
    if (!h.obox.empty()) {
        const size_t n4 = (h.flat.size() + 3) / 4 * 4;
        std::vector<float> areas(n4, 0.0f), tri(n4 * 12, 1.0e18f);
        for (size_t i = 0; i < h.flat.size(); ++i) {
            const DevTri &T = h.flat[i];
            areas[i] = T.area;
            for (int c = 0; c < 3; ++c) { tri[12 * i + c] = T.p0[c]; tri[12 * i + 4 + c] = T.p1[c]; tri[12 * i + 8 + c] = T.p2[c]; }
        }
        W3_TRY(upload3(s.allocs, areas.data(), areas.size(), &v.areas));
        W3_TRY(upload3(s.allocs, reinterpret_cast<const float4 *>(tri.data()), n4 * 3, &v.sampTri));
    }
    for (int l = 0; l < 12; ++l) v.obox_off[l] = h.obox_off[l];
    v.obox_levels = h.obox_levels;
    return WOST_OK;
}

int upload_mesh3(const wost3_mesh_desc &d, DeviceMesh3 &s)
{
    if (d.n_tris < 0 || d.n_verts < 0) return set_error(WOST_ERR_INVALID, "negative mesh size");
    const char *host = std::getenv("WOST3_HOST_BUILD");          // developer knob: the checker's builder instead
    if (host && std::atoi(host) != 0) return upload_mesh3_host(d, s);
    return build_mesh3_device(d, s);
}

}  // namespace wost

using namespace wost;

// ---- developer / test entry: both builders on one mesh, every array of the two uploaded meshes compared byte for byte ------
namespace {
template <class T>
int64_t differing_bytes(const T *a, const T *b, size_t count, int64_t *compared)
{
    if (count == 0) return 0;
    const size_t bytes = count * sizeof(T);
    if (!a || !b) return (a || b) ? (int64_t)bytes : 0;
    std::vector<unsigned char> ha(bytes), hb(bytes);
    if (hipMemcpy(ha.data(), a, bytes, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hb.data(), b, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return (int64_t)bytes;
    int64_t diff = 0;
    for (size_t i = 0; i < bytes; ++i) diff += ha[i] != hb[i];
    *compared += (int64_t)bytes;
    return diff;
}
}  // namespace

int wost3_mesh_build_check(const wost3_mesh_desc *mesh, int device, int32_t repeat, double *host_ms, double *device_ms, int64_t *mismatch)
{
    if (!mesh || !mismatch) return set_error(WOST_ERR_INVALID, "null argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return set_error(WOST_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n_dev) return set_error(WOST_ERR_INVALID, "device index out of range");
    W3_TRY(hipSetDevice(device));
    if (mesh->n_tris < 0 || mesh->n_verts < 0) return set_error(WOST_ERR_INVALID, "negative mesh size");
    struct Owned {
        DeviceMesh3 m;
        ~Owned()
        {
            for (void *p : m.allocs) (void)hipFree(p);
        }
    };
    using clock = std::chrono::steady_clock;
    double best_host = 1e300, best_dev = 1e300;
    Owned a, b;
    for (int r = 0; r < std::max(1, repeat); ++r) {
        Owned ha, hb;
        W3_TRY(hipDeviceSynchronize());
        auto t0 = clock::now();
        int rc = upload_mesh3_host(*mesh, ha.m);
        W3_TRY(hipDeviceSynchronize());
        auto t1 = clock::now();
        if (rc != WOST_OK) return rc;
        rc = build_mesh3_device(*mesh, hb.m);
        W3_TRY(hipDeviceSynchronize());
        auto t2 = clock::now();
        if (rc != WOST_OK) return rc;
        best_host = std::min(best_host, std::chrono::duration<double, std::milli>(t1 - t0).count());
        best_dev = std::min(best_dev, std::chrono::duration<double, std::milli>(t2 - t1).count());
        std::swap(a.m, ha.m);
        std::swap(b.m, hb.m);
    }
    if (host_ms) *host_ms = best_host;
    if (device_ms) *device_ms = best_dev;
    const DevMesh3 &x = a.m.view, &y = b.m.view;
    for (int i = 0; i < 16; ++i) mismatch[i] = 0;
    int64_t scalars = (x.n_tris != y.n_tris) + (x.n_edges != y.n_edges) + (x.levels != y.levels) + (x.first_leaf != y.first_leaf) +
                      (x.emissive != y.emissive) + (std::memcmp(&x.huge2, &y.huge2, 4) != 0) + (x.obox_levels != y.obox_levels);
    for (int l = 0; l < 12; ++l) scalars += x.obox_off[l] != y.obox_off[l];
    mismatch[14] = scalars;
    if (x.n_tris == 0 || scalars) return WOST_OK;
    const size_t cap = 3 * (size_t)x.first_leaf + 1, n_nodes = x.first_leaf + cap, n_slots = cap * 4, n = (size_t)x.n_tris;
    const size_t n_colors = x.colors ? (size_t)mesh->n_verts * 6 : 0;
    size_t obox_n = 0;
    if (x.obox_levels > 0) {
        const size_t run = (size_t)4 << (2 * (x.obox_levels - 1));
        obox_n = (size_t)x.obox_off[x.obox_levels - 1] + (n + run - 1) / run;
    }
    const size_t n4 = (n + 3) / 4 * 4;
    int64_t *cmp = &mismatch[15];
    mismatch[0] = differing_bytes(x.nodes, y.nodes, n_nodes * 6, cmp);
    mismatch[1] = differing_bytes(x.tri, y.tri, n_slots * 3, cmp);
    mismatch[2] = differing_bytes(x.triOrig, y.triOrig, n_slots, cmp);
    mismatch[3] = differing_bytes(x.slotOfOrig, y.slotOfOrig, n, cmp);
    mismatch[4] = differing_bytes(x.triVerts, y.triVerts, n_slots * 3, cmp);
    mismatch[5] = differing_bytes(x.colors, y.colors, n_colors, cmp);
    mismatch[6] = differing_bytes(x.flat, y.flat, n, cmp);
    mismatch[7] = differing_bytes(x.flatVerts, y.flatVerts, n * 3, cmp);
    mismatch[8] = differing_bytes(x.edges, y.edges, (size_t)x.n_edges, cmp);
    mismatch[9] = differing_bytes(x.slotEdges, y.slotEdges, n_slots * 12, cmp);
    mismatch[10] = differing_bytes(x.cones, y.cones, n_nodes * 6, cmp);
    mismatch[11] = differing_bytes(x.obox, y.obox, obox_n * 2, cmp);
    mismatch[12] = differing_bytes(x.areas, y.areas, obox_n ? n4 : 0, cmp);
    mismatch[13] = differing_bytes(x.sampTri, y.sampTri, obox_n ? n4 * 3 : 0, cmp);
    return WOST_OK;
}
