// wost_hip3d.hip -- the 3-D uniform Walk-on-Stars path on MI355X, gfx950 only.
//
// UniformIntegrator<3> of the reference: the DIM == 3 branches of integrator/uniform/integrator.cu
// (:128-211 separateEvaluationPoint with triangles and barycentric uv :150-168, :224-231
// handleBoundary, :336-444 sampleNeumann with three draws, :465-525 oneStepWalk), EvaluationGrid<3>
// (core/evaluation_grid.h:43-70), uniformSampleSphere<3> / Hemisphere<3> (util/sampling.h:20-27,57-66),
// frameFromNormal(Vector3f) (util/transformation.h:62-67, util/math_utils.h:141-151) and
// HarmonicGreenBall<3>::eval (util/green.h:82-90), behind wost3_* of include/wost.h.
//
// Design: the same regenerating walker as the 2-D round kernel -- one lane owns one PIXEL and walks
// its samples one after the other on the pixel's PCG stream (the reference's per-pixel order) --
// but a whole solve is ONE launch: a lane runs its pixel to the end.  Closest-point queries descend
// an implicit 4-ary LBVH over the triangles (3-D Morton order, axis-aligned child boxes, 96-byte
// nodes, near-first with the per-lane LDS stack and the key format of the 2-D tree, wost_device.h).
// Neumann meshes of up to WOST3_FLAT_MAX triangles (a box, a clipped plane) are walked with wave-uniform flat
// loops; larger ones descend the same kind of tree for the silhouette and ray queries (the triangle sampling of an
// EMISSIVE Neumann mesh stays a flat loop, as in 2-D).  Source term: a dense grid, trilinear.  Arithmetic contract: DESIGN.md 2.3 (the CPU restatement the tests compare against
// follows the same contract operation for operation).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/wost.h"
#include "lbvh.h"
#include "wost_device.h"
#include "wost_internal.h"
#include "wost_pool.h"

namespace wost {

#define WOST_4PI 12.5663706143591729539f
#define WOST3_FLAT_MAX 64        // Neumann meshes up to this size are walked with flat loops, larger ones through their tree

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ V3 cross3(V3 a, V3 b)
{
    return V3{__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ V3 madd3(V3 p, float t, V3 d) { return V3{__builtin_fmaf(t, d.x, p.x), __builtin_fmaf(t, d.y, p.y), __builtin_fmaf(t, d.z, p.z)}; }
__device__ __forceinline__ V3 normalize3(V3 a)
{
    const float l = sqrtf(dot3(a, a));
    return V3{a.x / l, a.y / l, a.z / l};
}

// one triangle as the flat loops read it (original order)
struct DevTri {
    float p0[3], p1[3], p2[3];
    float nraw[3], n[3];
    float area;
};
struct DevEdge3 {
    float pa[3], pb[3];
    int32_t t0, t1;
};

struct DevMesh3 {
    const float4 *nodes;     // [n_nodes * 6] child boxes: lox[4] loy[4] loz[4] hix[4] hiy[4] hiz[4]
    const float4 *tri;       // [slots * 3] p0, p1, p2 (w unused) in leaf order; empty slots far away
    const int32_t *triOrig;  // [slots] original triangle index (WOST_FAR_INDEX = empty)
    const int32_t *slotOfOrig; // [n_tris] the slot of an original triangle (closest_triangle_pool: its minimum is taken over original indices)
    const int32_t *triVerts; // [slots * 3] vertex ids (colour lookup)
    const float *colors;     // [n_verts * 6] or nullptr
    const DevTri *flat;      // [n_tris] original order
    const int32_t *flatVerts;// [n_tris * 3] vertex ids, original order
    const DevEdge3 *edges;   // [n_edges]
    const float4 *slotEdges; // [slots * 3 * 4] the silhouette test's operands of side k of the triangle in a slot, one record:
                             // (pa, kind) (pb, -) (n0, -) (n1, -); kind 0 = degenerate side or an edge that an earlier slot
                             // already lists (every edge is tested from one triangle only), 1 = two triangles, 2 = boundary
    const float4 *cones;     // [n_nodes * 6] normal cones of the four children: ax[4] ay[4] az[4] cos[4] sin[4] rad[4]
    int32_t n_tris, n_edges, levels, first_leaf, emissive;
    float huge2;             // squared distance beyond which a closest-point query is a scan by the whole wave (closest_triangle_wave)
    // boxes over runs of consecutive ORIGINAL triangle indices (sample_in_sphere3_tree): level l holds, per run of
    // 4^(l+1) triangles, two float4 (lo.xyz, hi.xyz) at obox + 2 * (obox_off[l] + run); obox_levels = 0: not built
    const float4 *obox;
    int32_t obox_off[12];
    int32_t obox_levels;
    // compact copies for those sweeps, original order, padded to a multiple of four triangles (area 0)
    const float *areas;      // the areas alone: the sums over runs that lie inside the ball
    const float4 *sampTri;   // [n * 3] p0 p1 p2 (w unused): a run of four is fetched with loads issued together
};

struct DevProbe3 {
    float scale, pos[3], up[3], right[3];
};

// ---- closest point on one triangle (Ericson 5.1.5), squared distance ------------------------
__device__ __forceinline__ float tri_d2(V3 p0, V3 p1, V3 p2, V3 q)
{
    const V3 ab = p1 - p0, ac = p2 - p0, ap = q - p0;
    const float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    V3 c;
    if (d1 <= 0.0f && d2 <= 0.0f) c = p0;
    else {
        const V3 bp = q - p1;
        const float d3 = dot3(ab, bp), d4 = dot3(ac, bp);
        if (d3 >= 0.0f && d4 <= d3) c = p1;
        else {
            const float vc = __builtin_fmaf(d1, d4, -(d3 * d2));
            if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) c = madd3(p0, d1 / (d1 - d3), ab);
            else {
                const V3 cp = q - p2;
                const float d5 = dot3(ab, cp), d6 = dot3(ac, cp);
                if (d6 >= 0.0f && d5 <= d6) c = p2;
                else {
                    const float vb = __builtin_fmaf(d5, d2, -(d1 * d6));
                    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) c = madd3(p0, d2 / (d2 - d6), ac);
                    else {
                        const float va = __builtin_fmaf(d3, d6, -(d5 * d4));
                        if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f)
                            c = madd3(p1, (d4 - d3) / ((d4 - d3) + (d5 - d6)), p2 - p1);
                        else {
                            const float denom = 1.0f / (va + vb + vc);
                            c = madd3(madd3(p0, vb * denom, ab), vc * denom, ac);
                        }
                    }
                }
            }
        }
    }
    const V3 w = q - c;
    return dot3(w, w);
}

__device__ __forceinline__ V3 ld3(const float *p) { return V3{p[0], p[1], p[2]}; }

// ---- LBVH traversal: near-first, LDS stack, keys = box distance | level | child (wost_device.h) ----
__device__ __forceinline__ float aabb_d2(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 q)
{
    const float dx = fmaxf(fmaxf(lox - q.x, q.x - hix), 0.0f), dy = fmaxf(fmaxf(loy - q.y, q.y - hiy), 0.0f),
                dz = fmaxf(fmaxf(loz - q.z, q.z - hiz), 0.0f);
    return __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
}

// relative slack of every box-against-best comparison of the 3-D trees: 1 + 2^-15.  Box, triangle and edge distances each carry a
// few 10^-7 of relative rounding; a looser slack (10^-4 at first) is as exact but opens every box of the mesh for a walker
// thousands of scene sizes away
constexpr float kSlack3 = 1.000030517578125f;

__device__ __forceinline__ bool trav_visit3(const DevMesh3 &m, V3 q, Trav &T, const LdsColumn &stk)
{
    const uint32_t g = level_first(T.level) + (uint32_t)T.pos;
    if (T.level == m.levels) {
        // a leaf: its four triangles, exactly; ties go to the lowest ORIGINAL index.  The leaf's record holds the
        // (padded) box of every triangle: a triangle whose box is farther than the best so far cannot win or tie,
        // and the box test costs a sixth of the exact distance.
        // (the slack of kSlack3: box and triangle distances come from different formulas, and far outside the mesh --
        // open boundaries let walkers escape -- their rounding grows with |q|, beyond the padding of the boxes)
        const float4 *ld = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 96u));
        const float4 BLX = ld[0], BLY = ld[1], BLZ = ld[2], BHX = ld[3], BHY = ld[4], BHZ = ld[5];
        const float bd0 = aabb_d2(BLX.x, BLY.x, BLZ.x, BHX.x, BHY.x, BHZ.x, q), bd1 = aabb_d2(BLX.y, BLY.y, BLZ.y, BHX.y, BHY.y, BHZ.y, q);
        const float bd2 = aabb_d2(BLX.z, BLY.z, BLZ.z, BHX.z, BHY.z, BHZ.z, q), bd3 = aabb_d2(BLX.w, BLY.w, BLZ.w, BHX.w, BHY.w, BHZ.w, q);
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int slot = 4 * T.pos + j;
            const float bdj = j == 0 ? bd0 : j == 1 ? bd1 : j == 2 ? bd2 : bd3;
            if (bdj > T.best.d2 * kSlack3) continue;
            const int o = m.triOrig[slot];
            if (o == WOST_FAR_INDEX) continue;
            const float4 a = m.tri[3 * (size_t)slot], b = m.tri[3 * (size_t)slot + 1], c = m.tri[3 * (size_t)slot + 2];
            const float d = tri_d2(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), q);
            if (d < T.best.d2) {
                T.best.d2 = d; T.best.slot = slot; T.best_orig = o;
            } else if (d == T.best.d2 && slot != T.best.slot) {
                if (T.best_orig < 0) T.best_orig = (T.best.slot >= 0) ? m.triOrig[T.best.slot] : WOST_FAR_INDEX;
                if (o < T.best_orig) { T.best.slot = slot; T.best_orig = o; }
            }
        }
        return trav_pop(T, stk, T.best.d2 * kSlack3);
    }
    const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 96u));
    const float4 LX = nd[0], LY = nd[1], LZ = nd[2], HX = nd[3], HY = nd[4], HZ = nd[5];
    const float d0 = aabb_d2(LX.x, LY.x, LZ.x, HX.x, HY.x, HZ.x, q), d1 = aabb_d2(LX.y, LY.y, LZ.y, HX.y, HY.y, HZ.y, q);
    const float d2 = aabb_d2(LX.z, LY.z, LZ.z, HX.z, HY.z, HZ.z, q), d3 = aabb_d2(LX.w, LY.w, LZ.w, HX.w, HY.w, HZ.w, q);
    const float bd = T.best.d2 * kSlack3;
    const uint32_t tag = (uint32_t)(T.level + 1) << 2;
    uint32_t k0 = (d0 <= bd) ? ((__float_as_uint(d0) & ~0x3Fu) | tag | 0u) : 0xffffffffu;
    uint32_t k1 = (d1 <= bd) ? ((__float_as_uint(d1) & ~0x3Fu) | tag | 1u) : 0xffffffffu;
    uint32_t k2 = (d2 <= bd) ? ((__float_as_uint(d2) & ~0x3Fu) | tag | 2u) : 0xffffffffu;
    uint32_t k3 = (d3 <= bd) ? ((__float_as_uint(d3) & ~0x3Fu) | tag | 3u) : 0xffffffffu;
    cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2);
    int sp = T.sp;
    stk.put(sp, k3); sp += (k3 != 0xffffffffu) ? 1 : 0;
    stk.put(sp, k2); sp += (k2 != 0xffffffffu) ? 1 : 0;
    stk.put(sp, k1); sp += (k1 != 0xffffffffu) ? 1 : 0;
    T.sp = sp;
    if (k0 != 0xffffffffu) {
        T.pos = 4 * T.pos + (int)(k0 & 3u);
        T.level = T.level + 1;
        return true;
    }
    return trav_pop(T, stk, bd);
}

// seed of a query: the triangle in `slot` (temporal hint: the previous closest triangle)
__device__ __forceinline__ Closest closest_triangle(const DevMesh3 &m, V3 q, int32_t hint, const LdsColumn &stk)
{
    Trav T = trav_begin(Closest{WOST_INF, -1});
    if (hint >= 0 && m.triOrig[hint] != WOST_FAR_INDEX) {
        const float4 a = m.tri[3 * (size_t)hint], b = m.tri[3 * (size_t)hint + 1], c = m.tri[3 * (size_t)hint + 2];
        T.best = Closest{tri_d2(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), q), hint};
        T.best_orig = m.triOrig[hint];
    }
    while (trav_visit3(m, q, T, stk)) {
    }
    return T.best;
}

// checkPointSide / computeProjectionRatio for triangles
__device__ __forceinline__ int tri_side(V3 p0, V3 nraw, V3 q)
{
    const float s = dot3(nraw, q - p0);
    return (0.0f < s) - (s < 0.0f);
}
__device__ __forceinline__ void tri_uv(V3 p0, V3 e0, V3 e1, V3 q, float &u, float &v)
{
    const V3 ap = q - p0;
    const float d00 = dot3(e0, e0), d01 = dot3(e0, e1), d11 = dot3(e1, e1), d20 = dot3(ap, e0), d21 = dot3(ap, e1);
    const float denom = __builtin_fmaf(d00, d11, -(d01 * d01));
    u = __builtin_fmaf(d11, d20, -(d01 * d21)) / denom;
    v = __builtin_fmaf(d00, d21, -(d01 * d20)) / denom;
}
// computeSurfaceColor<3> + barycentric_interpolate: (a w + b u) + c v
__device__ __forceinline__ void surface_color3(const float *colors, int i0, int i1, int i2, int side, float u, float v, float out[3])
{
    const float w = 1 - u - v;
    const int off = (side >= 0) ? 0 : 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = colors ? colors[6 * (size_t)i0 + off + c] : 0.0f, b = colors ? colors[6 * (size_t)i1 + off + c] : 0.0f,
                    cc = colors ? colors[6 * (size_t)i2 + off + c] : 0.0f;
        out[c] = (a * w + b * u) + cc * v;
    }
}

// ---- Neumann mesh, flat wave-uniform loops -----------------------------------------------------
__device__ __forceinline__ float closest_silhouette3_flat(const DevMesh3 &m, V3 q, float rmax)
{
    float best2 = rmax * rmax;
    bool found = false;
    for (int i = 0; i < m.n_edges; ++i) {
        const DevEdge3 E = m.edges[i];
        const V3 pa = ld3(E.pa), pb = ld3(E.pb), e = pb - pa;
        const float ee = dot3(e, e);
        float t = ee > 0.0f ? dot3(q - pa, e) / ee : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const V3 pt = madd3(pa, t, e), view = q - pt;
        const float d2 = dot3(view, view);
        if (d2 > best2) continue;
        bool is_sil = E.t1 < 0;
        if (!is_sil) {
            const V3 n0 = ld3(m.flat[E.t0].n), n1 = ld3(m.flat[E.t1].n);
            const float d = sqrtf(d2);
            if (d <= WOST_SIL_PRECISION) {
                const float det = dot3(normalize3(e), cross3(n0, n1));
                is_sil = (-det > WOST_SIL_PRECISION);
            } else {
                const V3 vd = v3(view.x / d, view.y / d, view.z / d);
                const float dot0 = dot3(vd, n0), dot1 = dot3(vd, n1);
                is_sil = !(fabsf(dot0) <= WOST_SIL_PRECISION || fabsf(dot1) <= WOST_SIL_PRECISION) && (dot0 * dot1 < 0.0f);
            }
        }
        if (is_sil && (d2 < best2 || !found)) {
            best2 = d2;
            found = true;
        }
    }
    return found ? sqrtf(best2) : WOST_INF;
}

__device__ __forceinline__ bool tri_ray3(V3 p0, V3 p1, V3 p2, V3 o, V3 d, float tmax, float &t)
{
    const V3 e0 = p1 - p0, e1 = p2 - p0;
    const V3 pvec = cross3(d, e1);
    const float det = dot3(e0, pvec);
    if (det == 0.0f) return false;
    const V3 tvec = o - p0;
    const float sgn = det < 0.0f ? -1.0f : 1.0f, adet = fabsf(det);
    const float u = dot3(tvec, pvec) * sgn;
    if (u < 0.0f || u > adet) return false;
    const V3 qvec = cross3(tvec, e0);
    const float v = dot3(d, qvec) * sgn;
    if (v < 0.0f || u + v > adet) return false;
    const float tt = dot3(e1, qvec), ts = tt * sgn;
    if (ts < 0.0f || ts > tmax * adet) return false;
    t = tt / det;
    return true;
}
__device__ __forceinline__ bool tri_ray(const DevTri &T, V3 o, V3 d, float tmax, float &t)
{
    return tri_ray3(ld3(T.p0), ld3(T.p1), ld3(T.p2), o, d, tmax, t);
}
__device__ __forceinline__ bool ray_closest3_flat(const DevMesh3 &m, V3 o, V3 d, float tmax, float &t_out, int &idx_out)
{
    bool hit = false;
    float bt = WOST_INF;
    int bi = -1;
    for (int i = 0; i < m.n_tris; ++i) {
        float t;
        if (tri_ray(m.flat[i], o, d, tmax, t) && (!hit || t < bt)) { bt = t; bi = i; hit = true; }
    }
    t_out = bt; idx_out = bi;
    return hit;
}
__device__ __forceinline__ bool ray_any3_flat(const DevMesh3 &m, V3 o, V3 d, float tmax)
{
    bool hit = false;
    for (int i = 0; i < m.n_tris; ++i) {
        float t;
        hit = hit || tri_ray(m.flat[i], o, d, tmax, t);
    }
    return hit;
}
__device__ __forceinline__ int sample_in_sphere3_flat(const DevMesh3 &m, V3 q, float R, float u, float &pdf)
{
    const float R2 = R * R;
    float total = 0.0f;
    for (int i = 0; i < m.n_tris; ++i) {
        const DevTri T = m.flat[i];
        if (T.area > 0.0f && tri_d2(ld3(T.p0), ld3(T.p1), ld3(T.p2), q) <= R2) total += T.area;
    }
    pdf = 0.0f;
    if (!(total > 0.0f)) return -1;
    const float target = u * total;
    float cum = 0.0f;
    int last = -1;
    bool done = false;
    for (int i = 0; i < m.n_tris; ++i) {
        const DevTri T = m.flat[i];
        if (!done && T.area > 0.0f && tri_d2(ld3(T.p0), ld3(T.p1), ld3(T.p2), q) <= R2) {
            cum += T.area; last = i;
            if (target < cum) done = true;
        }
    }
    const float a = m.flat[last].area;
    pdf = (a / total) / a;
    return last;
}

#ifdef WOST3_PROFILE
// developer build: wave-clock time spent in the sections of a step, summed over waves; visit counts of the queries
__device__ unsigned long long g_prof3[16];
#endif
// ---- the same queries on the tree, for Neumann meshes too large for flat loops -------------------------
// silhouette: an edge lies inside its triangle, a triangle inside its (padded) box, so boxes farther than the best
// silhouette edge so far cannot improve it; a leaf tests the three sides of its four triangles with the body of the
// flat loop (an edge shared by two triangles is simply tested twice).  The result is a minimum: order-free.
// Normal cone of a subtree (Sawhney et al. 2023, spatialized normal cone hierarchy; the 2-D twin is
// cone_may_hold_silhouette in wost_device.h): every normal of a triangle next to an edge of the subtree lies within
// `half` of the axis, every point of those edges within `rad` of c.  A silhouette edge needs view . n0 and view . n1 of
// opposite signs, i.e. a normal of the cone perpendicular to a direction of the view cone: impossible while
// |cos(angle(axis, q - c))| > sin(half + view half angle).  Conservative (slack 1e-3 on both comparisons, rad padded
// by more than WOST_SIL_PRECISION so that a query standing on an edge is inside the ball): it only ever removes
// edges the exact test would reject, so the minimum is that of the flat loop.
__device__ __forceinline__ bool cone3_may_hold_silhouette(float ax, float ay, float az, float ch, float sh, float rad, V3 c, V3 q)
{
    if (ch <= 0.0f) return true;                       // marked "cannot prune"
    const V3 w = c - q;
    const float l2 = dot3(w, w);
    if (l2 <= rad * rad * 1.0001f) return true;        // q inside the ball: no view cone
    const float inv_l = 1.0f / sqrtf(l2);
    const float sv = fminf(rad * inv_l, 1.0f);
    const float cv = sqrtf(fmaxf(1.0f - sv * sv, 0.0f));
    const float cos_sum = ch * cv - sh * sv;
    if (cos_sum <= 1e-3f) return true;
    const float sin_sum = sh * cv + ch * sv;
    const float cs = (ax * w.x + ay * w.y + az * w.z) * inv_l;
    return fabsf(cs) <= sin_sum + 1e-3f;
}

// the flat loop's edge test on the packed record of a leaf slot (its operands in one load instead of three dependent ones)
__device__ __forceinline__ void silhouette_record_test(float4 r0, float4 r1, float4 r2, float4 r3, V3 q, float &best2, bool &found)
{
    if (r0.w == 0.0f) return;
    const V3 pa = v3(r0.x, r0.y, r0.z), pb = v3(r1.x, r1.y, r1.z), ev = pb - pa;
    const float ee = dot3(ev, ev);
    float t = ee > 0.0f ? dot3(q - pa, ev) / ee : 0.0f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const V3 pt = madd3(pa, t, ev), view = q - pt;
    const float d2 = dot3(view, view);
    if (d2 > best2) return;
    bool is_sil = r0.w == 2.0f;
    if (!is_sil) {
        const V3 n0 = v3(r2.x, r2.y, r2.z), n1 = v3(r3.x, r3.y, r3.z);
        const float d = sqrtf(d2);
        if (d <= WOST_SIL_PRECISION) {
            const float det = dot3(normalize3(ev), cross3(n0, n1));
            is_sil = (-det > WOST_SIL_PRECISION);
        } else {
            const V3 vd = v3(view.x / d, view.y / d, view.z / d);
            const float dot0 = dot3(vd, n0), dot1 = dot3(vd, n1);
            is_sil = !(fabsf(dot0) <= WOST_SIL_PRECISION || fabsf(dot1) <= WOST_SIL_PRECISION) && (dot0 * dot1 < 0.0f);
        }
    }
    if (is_sil && (d2 < best2 || !found)) {
        best2 = d2;
        found = true;
    }
}

// state of a silhouette query between node visits: the traversal (T.best.d2 = the slack pruning bound), the exact
// minimum so far (the flat loop's variable) and whether any silhouette edge was met
struct SilQuery3 {
    Trav T;
    float best2;
    bool found;
};
__device__ __forceinline__ SilQuery3 sil3_begin(float rmax)
{
    const float best2 = rmax * rmax;
    return SilQuery3{trav_begin(Closest{best2 * kSlack3, -1}), best2, false};
}
__device__ __forceinline__ float sil3_result(const SilQuery3 &Q) { return Q.found ? sqrtf(Q.best2) : WOST_INF; }

// visit ONE node; false = the query is complete
__device__ __forceinline__ bool sil3_visit(const DevMesh3 &m, V3 q, SilQuery3 &Q, const LdsColumn &stk)
{
    Trav &T = Q.T;
    float &best2 = Q.best2;
    bool &found = Q.found;
    bool more;
        const uint32_t g = level_first(T.level) + (uint32_t)T.pos;
        const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 96u));
        const float4 LX = nd[0], LY = nd[1], LZ = nd[2], HX = nd[3], HY = nd[4], HZ = nd[5];
        const float d0 = aabb_d2(LX.x, LY.x, LZ.x, HX.x, HY.x, HZ.x, q), d1 = aabb_d2(LX.y, LY.y, LZ.y, HX.y, HY.y, HZ.y, q);
        const float d2 = aabb_d2(LX.z, LY.z, LZ.z, HX.z, HY.z, HZ.z, q), d3 = aabb_d2(LX.w, LY.w, LZ.w, HX.w, HY.w, HZ.w, q);
        if (T.level == m.levels) {
            // the record of a leaf holds the boxes of its four triangles: an edge is tested from the triangle that
            // lists it, and only while that triangle's box is not beyond the best edge
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                const int slot = 4 * T.pos + j;
                const float bdj = j == 0 ? d0 : j == 1 ? d1 : j == 2 ? d2 : d3;
                if (bdj > best2 * kSlack3) continue;
                const float4 *rec = m.slotEdges + 12 * (size_t)slot;
#pragma unroll 1
                for (int k = 0; k < 3; ++k) silhouette_record_test(rec[4 * k], rec[4 * k + 1], rec[4 * k + 2], rec[4 * k + 3], q, best2, found);
            }
            T.best.d2 = best2 * kSlack3;
            more = trav_pop(T, stk);
        } else {
            const float bd = T.best.d2;
            const uint32_t tag = (uint32_t)(T.level + 1) << 2;
            const float4 *cn = m.cones + 6 * (size_t)g;
            const float4 AX = cn[0], AY = cn[1], AZ = cn[2], CH = cn[3], SH = cn[4], RD = cn[5];
            const bool c0 = d0 <= bd && cone3_may_hold_silhouette(AX.x, AY.x, AZ.x, CH.x, SH.x, RD.x, v3(0.5f * (LX.x + HX.x), 0.5f * (LY.x + HY.x), 0.5f * (LZ.x + HZ.x)), q);
            const bool c1 = d1 <= bd && cone3_may_hold_silhouette(AX.y, AY.y, AZ.y, CH.y, SH.y, RD.y, v3(0.5f * (LX.y + HX.y), 0.5f * (LY.y + HY.y), 0.5f * (LZ.y + HZ.y)), q);
            const bool c2 = d2 <= bd && cone3_may_hold_silhouette(AX.z, AY.z, AZ.z, CH.z, SH.z, RD.z, v3(0.5f * (LX.z + HX.z), 0.5f * (LY.z + HY.z), 0.5f * (LZ.z + HZ.z)), q);
            const bool c3 = d3 <= bd && cone3_may_hold_silhouette(AX.w, AY.w, AZ.w, CH.w, SH.w, RD.w, v3(0.5f * (LX.w + HX.w), 0.5f * (LY.w + HY.w), 0.5f * (LZ.w + HZ.w)), q);
            uint32_t k0 = c0 ? ((__float_as_uint(d0) & ~0x3Fu) | tag | 0u) : 0xffffffffu;
            uint32_t k1 = c1 ? ((__float_as_uint(d1) & ~0x3Fu) | tag | 1u) : 0xffffffffu;
            uint32_t k2 = c2 ? ((__float_as_uint(d2) & ~0x3Fu) | tag | 2u) : 0xffffffffu;
            uint32_t k3 = c3 ? ((__float_as_uint(d3) & ~0x3Fu) | tag | 3u) : 0xffffffffu;
            cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2);
            int sp = T.sp;
            stk.put(sp, k3); sp += (k3 != 0xffffffffu) ? 1 : 0;
            stk.put(sp, k2); sp += (k2 != 0xffffffffu) ? 1 : 0;
            stk.put(sp, k1); sp += (k1 != 0xffffffffu) ? 1 : 0;
            T.sp = sp;
            if (k0 != 0xffffffffu) {
                T.pos = 4 * T.pos + (int)(k0 & 3u);
                T.level = T.level + 1;
                more = true;
            } else {
                more = trav_pop(T, stk);
            }
        }
    return more;
}

__device__ __forceinline__ float closest_silhouette3_tree(const DevMesh3 &m, V3 q, float rmax, const LdsColumn &stk)
{
    SilQuery3 Q = sil3_begin(rmax);
#ifdef WOST3_PROFILE
    unsigned prof_inner = 0, prof_leaf = 0;
#endif
    for (;;) {
#ifdef WOST3_PROFILE
        if (Q.T.level == m.levels) ++prof_leaf; else ++prof_inner;
#endif
        if (!sil3_visit(m, q, Q, stk)) break;
    }
#ifdef WOST3_PROFILE
    atomicAdd(&g_prof3[12], (unsigned long long)prof_inner);
    atomicAdd(&g_prof3[13], (unsigned long long)prof_leaf);
    atomicAdd(&g_prof3[14], 1ull);
#endif
    return sil3_result(Q);
}

// rays: where the ray enters a child box (slabs; the boxes are padded and the comparison is slack, so a box that
// holds a hit of tri_ray is never skipped), +inf if it misses it or enters beyond `limit`
__device__ __forceinline__ float ray_aabb_entry(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 o, V3 d, V3 inv, float limit)
{
    float tmin = 0.0f, tmax = limit;
    // an axis the ray does not move along only asks whether the origin lies in the slab
    {
        const float t1 = (lox - o.x) * inv.x, t2 = (hix - o.x) * inv.x;
        const bool par = d.x == 0.0f;
        const bool out = par && (o.x < lox || o.x > hix);
        tmin = out ? WOST_INF : fmaxf(tmin, par ? tmin : fminf(t1, t2));
        tmax = par ? tmax : fminf(tmax, fmaxf(t1, t2));
    }
    {
        const float t1 = (loy - o.y) * inv.y, t2 = (hiy - o.y) * inv.y;
        const bool par = d.y == 0.0f;
        const bool out = par && (o.y < loy || o.y > hiy);
        tmin = out ? WOST_INF : fmaxf(tmin, par ? tmin : fminf(t1, t2));
        tmax = par ? tmax : fminf(tmax, fmaxf(t1, t2));
    }
    {
        const float t1 = (loz - o.z) * inv.z, t2 = (hiz - o.z) * inv.z;
        const bool par = d.z == 0.0f;
        const bool out = par && (o.z < loz || o.z > hiz);
        tmin = out ? WOST_INF : fmaxf(tmin, par ? tmin : fminf(t1, t2));
        tmax = par ? tmax : fminf(tmax, fmaxf(t1, t2));
    }
    // slack of a few ulps on the comparison: the parameter of a hit and the slab parameters are rounded independently
    return (tmin <= tmax * 1.00001f + 1e-30f) ? fminf(tmin, tmax) : WOST_INF;
}

// state of a ray query between node visits: T.best.d2 = the pruning bound (boxes entered beyond it cannot hold a better
// hit), the best hit so far
struct RayQuery3 {
    Trav T;
    V3 inv;
    float bt;
    int bi;
    bool hit;
};
__device__ __forceinline__ RayQuery3 ray3_begin(V3 d, float tmax)
{
    return RayQuery3{trav_begin(Closest{tmax * 1.00001f + 1e-30f, -1}), v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), WOST_INF, -1, false};
}

// closest hit (smallest t, lowest original index on ties: the flat loop's answer) or any hit: visit ONE node;
// false = the query is complete
template <bool ANY_HIT>
__device__ __forceinline__ bool ray3_visit(const DevMesh3 &m, V3 o, V3 d, float tmax, RayQuery3 &Q, const LdsColumn &stk)
{
    Trav &T = Q.T;
    const V3 inv = Q.inv;
    float &bt = Q.bt;
    int &bi = Q.bi;
    bool &hit = Q.hit;
    bool more;
        if (T.level == m.levels) {
            // the record of a leaf: the boxes of its four triangles; the triangles themselves in leaf order
            const uint32_t gl = level_first(T.level) + (uint32_t)T.pos;
            const float4 *ld = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(gl, 96u));
            const float4 LX = ld[0], LY = ld[1], LZ = ld[2], HX = ld[3], HY = ld[4], HZ = ld[5];
            const float bd = T.best.d2;
            const float e0 = ray_aabb_entry(LX.x, LY.x, LZ.x, HX.x, HY.x, HZ.x, o, d, inv, bd), e1 = ray_aabb_entry(LX.y, LY.y, LZ.y, HX.y, HY.y, HZ.y, o, d, inv, bd);
            const float e2 = ray_aabb_entry(LX.z, LY.z, LZ.z, HX.z, HY.z, HZ.z, o, d, inv, bd), e3 = ray_aabb_entry(LX.w, LY.w, LZ.w, HX.w, HY.w, HZ.w, o, d, inv, bd);
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                const int slot = 4 * T.pos + j;
                const float ej = j == 0 ? e0 : j == 1 ? e1 : j == 2 ? e2 : e3;
                if (!(ej <= T.best.d2)) continue;                 // empty slots lie far away
                const float4 a = m.tri[3 * (size_t)slot], b = m.tri[3 * (size_t)slot + 1], c = m.tri[3 * (size_t)slot + 2];
                float t;
                if (tri_ray3(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), o, d, tmax, t)) {
                    const int oi = m.triOrig[slot];
                    if (ANY_HIT) {
                        bt = t; bi = oi; hit = true;
                        return false;
                    }
                    if (!hit || t < bt || (t == bt && oi < bi)) {
                        bt = t; bi = oi; hit = true;
                        T.best.d2 = fminf(T.best.d2, bt * 1.00001f + 1e-30f);
                    }
                }
            }
            more = trav_pop(T, stk);
        } else {
            const uint32_t g = level_first(T.level) + (uint32_t)T.pos;
            const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 96u));
            const float4 LX = nd[0], LY = nd[1], LZ = nd[2], HX = nd[3], HY = nd[4], HZ = nd[5];
            const float bd = T.best.d2;
            const float d0 = ray_aabb_entry(LX.x, LY.x, LZ.x, HX.x, HY.x, HZ.x, o, d, inv, bd), d1 = ray_aabb_entry(LX.y, LY.y, LZ.y, HX.y, HY.y, HZ.y, o, d, inv, bd);
            const float d2 = ray_aabb_entry(LX.z, LY.z, LZ.z, HX.z, HY.z, HZ.z, o, d, inv, bd), d3 = ray_aabb_entry(LX.w, LY.w, LZ.w, HX.w, HY.w, HZ.w, o, d, inv, bd);
            const uint32_t tag = (uint32_t)(T.level + 1) << 2;
            uint32_t k0 = (d0 <= bd) ? ((__float_as_uint(d0) & ~0x3Fu) | tag | 0u) : 0xffffffffu;
            uint32_t k1 = (d1 <= bd) ? ((__float_as_uint(d1) & ~0x3Fu) | tag | 1u) : 0xffffffffu;
            uint32_t k2 = (d2 <= bd) ? ((__float_as_uint(d2) & ~0x3Fu) | tag | 2u) : 0xffffffffu;
            uint32_t k3 = (d3 <= bd) ? ((__float_as_uint(d3) & ~0x3Fu) | tag | 3u) : 0xffffffffu;
            cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2);
            int sp = T.sp;
            stk.put(sp, k3); sp += (k3 != 0xffffffffu) ? 1 : 0;
            stk.put(sp, k2); sp += (k2 != 0xffffffffu) ? 1 : 0;
            stk.put(sp, k1); sp += (k1 != 0xffffffffu) ? 1 : 0;
            T.sp = sp;
            if (k0 != 0xffffffffu) {
                T.pos = 4 * T.pos + (int)(k0 & 3u);
                T.level = T.level + 1;
                more = true;
            } else {
                more = trav_pop(T, stk);
            }
        }
    return more;
}

template <bool ANY_HIT>
__device__ __forceinline__ bool ray3_tree(const DevMesh3 &m, V3 o, V3 d, float tmax, float &t_out, int &idx_out, const LdsColumn &stk)
{
    RayQuery3 Q = ray3_begin(d, tmax);
    while (ray3_visit<ANY_HIT>(m, o, d, tmax, Q, stk)) {
    }
    t_out = Q.bt; idx_out = Q.bi;
    return Q.hit;
}

// ---- the same queries answered by a whole WAVE for all its walkers together (the loop of wost_pool.h) -----------------------
// Inside a step every lane used to run its own query to completion (closest_silhouette3_tree, ray3_tree), and its closest
// triangle in the lane machine of walk3_kernel: a wave lasts as long as its longest query, and a leaf visit -- twelve edge
// records, or four exact triangle distances, behind per-lane skips -- is executed for the whole wave whenever one lane needs
// it: 9 % (Neumann shell) and 20 % (Dirichlet icosphere) of the vector lanes did work (profiles/r03_b_*).  A frame of 512^2
// walkers cannot be cut into stage queues across the chip either: it has fewer walkers than the chip has lanes, every query
// would still sit alone in its lane.  So the work of the 64 walkers of ONE wave goes through task pools in LDS (wost_pool.h):
// node tasks measure the four children (boxes, normal cones, slabs) against their owner's bound, slot tasks evaluate one
// triangle (or its three edge records) and fold the result into the owner's words with LDS atomics.  All three queries are
// minima -- over bits(d^2) << 32 | original index, over silhouette edges within rmax, over bits(|t|) << 32 | original index --
// so the answers are the flat loops', bit for bit.
using WavePool3 = WavePool;
constexpr int kPool3OwnerWords = 10 * 64;      // per-owner words of the largest of the three queries (the ray)

// closest silhouette edge within rmax of q, for every lane with `active` (all 64 lanes must call)
__device__ __forceinline__ float closest_silhouette3_wave(const DevMesh3 &m, V3 q, float rmax, bool active, const WavePool3 &W, const LdsColumn &stk)
{
    const int lane = threadIdx.x & 63;
    float *oq = reinterpret_cast<float *>(W.own);                 // x [0, 64), y [64, 128), z [128, 192)
    uint32_t *obest = W.own + 192, *ofound = W.own + 256;         // the flat loop's best2 (bits) and `found`
    if (active) {
        oq[lane] = q.x; oq[64 + lane] = q.y; oq[128 + lane] = q.z;
        obest[lane] = __float_as_uint(rmax * rmax);
        ofound[lane] = 0u;
    }
    const bool done = pool_run(
        W, m.levels, active, 64,
        [&](uint32_t g, int owner, bool leaf, bool (&v)[4], uint32_t (&key)[4]) {
            const V3 oqv = v3(oq[owner], oq[64 + owner], oq[128 + owner]);
            const float bd = __uint_as_float(obest[owner]) * kSlack3;
            const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 96u));
            const float4 LX = nd[0], LY = nd[1], LZ = nd[2], HX = nd[3], HY = nd[4], HZ = nd[5];
            const float d0 = aabb_d2(LX.x, LY.x, LZ.x, HX.x, HY.x, HZ.x, oqv), d1 = aabb_d2(LX.y, LY.y, LZ.y, HX.y, HY.y, HZ.y, oqv);
            const float d2 = aabb_d2(LX.z, LY.z, LZ.z, HX.z, HY.z, HZ.z, oqv), d3 = aabb_d2(LX.w, LY.w, LZ.w, HX.w, HY.w, HZ.w, oqv);
            if (leaf) {
                // the record of a leaf holds the boxes of its four triangles: an edge is tested from the triangle that lists it
                v[0] = !(d0 > bd); v[1] = !(d1 > bd); v[2] = !(d2 > bd); v[3] = !(d3 > bd);
            } else {
                const float4 *cn = m.cones + 6 * (size_t)g;
                const float4 AX = cn[0], AY = cn[1], AZ = cn[2], CH = cn[3], SH = cn[4], RD = cn[5];
                const bool c0 = d0 <= bd && cone3_may_hold_silhouette(AX.x, AY.x, AZ.x, CH.x, SH.x, RD.x, v3(0.5f * (LX.x + HX.x), 0.5f * (LY.x + HY.x), 0.5f * (LZ.x + HZ.x)), oqv);
                const bool c1 = d1 <= bd && cone3_may_hold_silhouette(AX.y, AY.y, AZ.y, CH.y, SH.y, RD.y, v3(0.5f * (LX.y + HX.y), 0.5f * (LY.y + HY.y), 0.5f * (LZ.y + HZ.y)), oqv);
                const bool c2 = d2 <= bd && cone3_may_hold_silhouette(AX.z, AY.z, AZ.z, CH.z, SH.z, RD.z, v3(0.5f * (LX.z + HX.z), 0.5f * (LY.z + HY.z), 0.5f * (LZ.z + HZ.z)), oqv);
                const bool c3 = d3 <= bd && cone3_may_hold_silhouette(AX.w, AY.w, AZ.w, CH.w, SH.w, RD.w, v3(0.5f * (LX.w + HX.w), 0.5f * (LY.w + HY.w), 0.5f * (LZ.w + HZ.w)), oqv);
                key[0] = c0 ? ((__float_as_uint(d0) & ~0x3u) | 0u) : 0xffffffffu;
                key[1] = c1 ? ((__float_as_uint(d1) & ~0x3u) | 1u) : 0xffffffffu;
                key[2] = c2 ? ((__float_as_uint(d2) & ~0x3u) | 2u) : 0xffffffffu;
                key[3] = c3 ? ((__float_as_uint(d3) & ~0x3u) | 3u) : 0xffffffffu;
            }
        },
        [&](uint32_t slot, int owner) {
            const V3 oqv = v3(oq[owner], oq[64 + owner], oq[128 + owner]);
            const float b0 = __uint_as_float(obest[owner]);
            const bool f0 = ofound[owner] != 0u;
            float b = b0;
            bool f = f0;
            const float4 *rec = m.slotEdges + 12 * (size_t)slot;
#pragma unroll
            for (int c = 0; c < 3; ++c) silhouette_record_test(rec[4 * c], rec[4 * c + 1], rec[4 * c + 2], rec[4 * c + 3], oqv, b, f);
            if (f && (b < b0 || !f0)) {
                atomicMin(&obest[owner], __float_as_uint(b));
                ofound[owner] = 1u;
            }
        });
    float r = WOST_INF;
    if (!done) {
        if (active) r = closest_silhouette3_tree(m, q, rmax, stk);
    } else if (active && ofound[lane] != 0u) {
        r = sqrtf(__uint_as_float(obest[lane]));
    }
    wave_lds_fence();
    return r;
}

// the walker's ray: closest hit (smallest t, lowest original index among equal ones) for every lane with `active`
__device__ __forceinline__ bool ray_closest3_wave(const DevMesh3 &m, V3 o, V3 d, float tmax, bool active, float &t_out, int &idx_out, const WavePool3 &W,
                                                  const LdsColumn &stk, int slot_trigger)
{
    const int lane = threadIdx.x & 63;
    unsigned long long *okey = reinterpret_cast<unsigned long long *>(W.own);     // [64]: bits(|t|) << 32 | original index
    float *of = reinterpret_cast<float *>(W.own) + 128;                            // o.xyz, d.xyz, tmax: 7 x [64]
    uint32_t *obound = W.own + 128 + 7 * 64;                                       // the pruning bound (bits)
    if (active) {
        okey[lane] = ~0ull;
        of[lane] = o.x; of[64 + lane] = o.y; of[128 + lane] = o.z;
        of[192 + lane] = d.x; of[256 + lane] = d.y; of[320 + lane] = d.z;
        of[384 + lane] = tmax;
        obound[lane] = __float_as_uint(tmax * 1.00001f + 1e-30f);
    }
    const bool done = pool_run(
        W, m.levels, active, slot_trigger,
        [&](uint32_t g, int owner, bool leaf, bool (&v)[4], uint32_t (&key)[4]) {
            const V3 ro = v3(of[owner], of[64 + owner], of[128 + owner]), rd = v3(of[192 + owner], of[256 + owner], of[320 + owner]);
            const V3 inv = v3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
            const float bd = __uint_as_float(obound[owner]);
            const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 96u));
            const float4 LX = nd[0], LY = nd[1], LZ = nd[2], HX = nd[3], HY = nd[4], HZ = nd[5];
            const float d0 = ray_aabb_entry(LX.x, LY.x, LZ.x, HX.x, HY.x, HZ.x, ro, rd, inv, bd), d1 = ray_aabb_entry(LX.y, LY.y, LZ.y, HX.y, HY.y, HZ.y, ro, rd, inv, bd);
            const float d2 = ray_aabb_entry(LX.z, LY.z, LZ.z, HX.z, HY.z, HZ.z, ro, rd, inv, bd), d3 = ray_aabb_entry(LX.w, LY.w, LZ.w, HX.w, HY.w, HZ.w, ro, rd, inv, bd);
            if (leaf) {
                v[0] = d0 <= bd; v[1] = d1 <= bd; v[2] = d2 <= bd; v[3] = d3 <= bd;
            } else {
                key[0] = (d0 <= bd) ? ((__float_as_uint(d0) & ~0x3u) | 0u) : 0xffffffffu;
                key[1] = (d1 <= bd) ? ((__float_as_uint(d1) & ~0x3u) | 1u) : 0xffffffffu;
                key[2] = (d2 <= bd) ? ((__float_as_uint(d2) & ~0x3u) | 2u) : 0xffffffffu;
                key[3] = (d3 <= bd) ? ((__float_as_uint(d3) & ~0x3u) | 3u) : 0xffffffffu;
            }
        },
        [&](uint32_t slot, int owner) {
            const V3 ro = v3(of[owner], of[64 + owner], of[128 + owner]), rd = v3(of[192 + owner], of[256 + owner], of[320 + owner]);
            const float4 a = m.tri[3 * (size_t)slot], b = m.tri[3 * (size_t)slot + 1], c = m.tri[3 * (size_t)slot + 2];
            float t;
            if (tri_ray3(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), ro, rd, of[384 + owner], t)) {
                const float at = fabsf(t);       // (t may be -0)
                atomicMin(&okey[owner], ((unsigned long long)__float_as_uint(at) << 32) | (unsigned long long)(uint32_t)m.triOrig[slot]);
                atomicMin(&obound[owner], __float_as_uint(at * 1.00001f + 1e-30f));
            }
        });
    bool hit = false;
    t_out = WOST_INF;
    idx_out = -1;
    if (!done) {
        if (active) hit = ray3_tree<false>(m, o, d, tmax, t_out, idx_out, stk);
    } else if (active) {
        const unsigned long long key = okey[lane];
        if (key != ~0ull) {
            // the winner's parameter from its own test (the key holds |t|; the operands are the leaf-ordered copy's)
            idx_out = (int)(uint32_t)key;
            hit = tri_ray(m.flat[idx_out], o, d, tmax, t_out);
        }
    }
    wave_lds_fence();
    return hit;
}

// the closest triangle (closest_triangle: smallest distance, lowest original index among equal ones) for every lane with
// `active`, seeded with `seed` = (squared distance, slot) of the temporal hint or (inf, -1)
__device__ __forceinline__ Closest closest_triangle_pool(const DevMesh3 &m, V3 q, Closest seed, bool active, const WavePool3 &W, const LdsColumn &stk, int slot_trigger)
{
    const int lane = threadIdx.x & 63;
    unsigned long long *okey = reinterpret_cast<unsigned long long *>(W.own);     // [64]: bits(d2) << 32 | original index
    float *oq = reinterpret_cast<float *>(W.own) + 128;                            // q.xyz: 3 x [64]
    if (active) {
        const uint32_t so = seed.slot >= 0 ? (uint32_t)m.triOrig[seed.slot] : 0xffffffffu;
        okey[lane] = ((unsigned long long)__float_as_uint(seed.d2) << 32) | so;
        oq[lane] = q.x; oq[64 + lane] = q.y; oq[128 + lane] = q.z;
    }
    const bool done = pool_run(
        W, m.levels, active, slot_trigger,
        [&](uint32_t g, int owner, bool leaf, bool (&v)[4], uint32_t (&key)[4]) {
            const V3 oqv = v3(oq[owner], oq[64 + owner], oq[128 + owner]);
            const float bd = __uint_as_float((uint32_t)(okey[owner] >> 32)) * kSlack3;
            const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 96u));
            const float4 LX = nd[0], LY = nd[1], LZ = nd[2], HX = nd[3], HY = nd[4], HZ = nd[5];
            const float d0 = aabb_d2(LX.x, LY.x, LZ.x, HX.x, HY.x, HZ.x, oqv), d1 = aabb_d2(LX.y, LY.y, LZ.y, HX.y, HY.y, HZ.y, oqv);
            const float d2 = aabb_d2(LX.z, LY.z, LZ.z, HX.z, HY.z, HZ.z, oqv), d3 = aabb_d2(LX.w, LY.w, LZ.w, HX.w, HY.w, HZ.w, oqv);
            if (leaf) {
                v[0] = !(d0 > bd); v[1] = !(d1 > bd); v[2] = !(d2 > bd); v[3] = !(d3 > bd);
            } else {
                key[0] = (d0 <= bd) ? ((__float_as_uint(d0) & ~0x3u) | 0u) : 0xffffffffu;
                key[1] = (d1 <= bd) ? ((__float_as_uint(d1) & ~0x3u) | 1u) : 0xffffffffu;
                key[2] = (d2 <= bd) ? ((__float_as_uint(d2) & ~0x3u) | 2u) : 0xffffffffu;
                key[3] = (d3 <= bd) ? ((__float_as_uint(d3) & ~0x3u) | 3u) : 0xffffffffu;
            }
        },
        [&](uint32_t slot, int owner) {
            const int32_t o = m.triOrig[slot];
            if (o == WOST_FAR_INDEX) return;
            const float4 a = m.tri[3 * (size_t)slot], b = m.tri[3 * (size_t)slot + 1], c = m.tri[3 * (size_t)slot + 2];
            const float d = tri_d2(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), v3(oq[owner], oq[64 + owner], oq[128 + owner]));
            atomicMin(&okey[owner], ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(uint32_t)o);
        });
    Closest r = seed;
    if (!done) {
        if (active) r = closest_triangle(m, q, seed.slot, stk);
    } else if (active) {
        const unsigned long long key = okey[lane];
        const uint32_t o = (uint32_t)key;
        r = Closest{__uint_as_float((uint32_t)(key >> 32)), o == 0xffffffffu ? -1 : m.slotOfOrig[o]};
    }
    wave_lds_fence();
    return r;
}

template <bool NTREE>
__device__ __forceinline__ float closest_silhouette3(const DevMesh3 &m, V3 q, float rmax, const LdsColumn &stk)
{
    if (NTREE) return closest_silhouette3_tree(m, q, rmax, stk);
    return closest_silhouette3_flat(m, q, rmax);
}
template <bool NTREE>
__device__ __forceinline__ bool ray_closest3(const DevMesh3 &m, V3 o, V3 d, float tmax, float &t_out, int &idx_out, const LdsColumn &stk)
{
    if (NTREE) return ray3_tree<false>(m, o, d, tmax, t_out, idx_out, stk);
    return ray_closest3_flat(m, o, d, tmax, t_out, idx_out);
}
template <bool NTREE>
__device__ __forceinline__ bool ray_any3(const DevMesh3 &m, V3 o, V3 d, float tmax, const LdsColumn &stk)
{
    if (NTREE) {
        float t;
        int i;
        return ray3_tree<true>(m, o, d, tmax, t, i, stk);
    }
    return ray_any3_flat(m, o, d, tmax);
}

// The selection of sample_in_sphere3_flat for meshes too large to walk twice per step: the probabilities are defined
// over the triangles in ORIGINAL index order, so runs of consecutive indices carry boxes and an index-ordered sweep
// skips every aligned run whose box lies beyond the ball, coarsest first (wost_device.h has the 2-D twin).  The
// triangles that are tested, their order and the float sums are those of the flat loop.
// f(i): the group of four triangles from i on, of a run that touches the ball; g(i): of a run inside it (wost_device.h)
template <class F, class G>
__device__ __forceinline__ void sweep_in_sphere3(const DevMesh3 &m, V3 q, float R2, F f, G g)
{
    const float R2s = R2 * kSlack3, R2i = R2 * 0.9999f;
    int i = 0;
    while (i < m.n_tris) {
        int skip = 0, inside = 0;
        for (int l = m.obox_levels - 1; l >= 0 && (skip | inside) == 0; --l) {
            const int run = 4 << (2 * l);
            if ((i & (run - 1)) == 0) {
                const float4 lo = m.obox[2 * (m.obox_off[l] + i / run)], hi = m.obox[2 * (m.obox_off[l] + i / run) + 1];
                if (aabb_d2(lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, q) > R2s) {
                    skip = run;
                } else {
                    const float fx = fmaxf(fabsf(q.x - lo.x), fabsf(q.x - hi.x)), fy = fmaxf(fabsf(q.y - lo.y), fabsf(q.y - hi.y));
                    const float fz = fmaxf(fabsf(q.z - lo.z), fabsf(q.z - hi.z));
                    if (__builtin_fmaf(fx, fx, __builtin_fmaf(fy, fy, fz * fz)) <= R2i) inside = run;
                }
            }
        }
        if (skip) {
            i += skip;
            continue;
        }
        if (inside) {
            const int end = min(i + inside, m.n_tris);
            for (; i < end; i += 4)
                if (!g(i)) return;
            continue;
        }
        if (!f(i)) return;
        i += 4;
    }
}

// the four triangles from i on, in order: take(index, area) for those the flat loop accepts (padding has area 0)
template <bool TEST, class T>
__device__ __forceinline__ bool sample_group3(const DevMesh3 &m, int i, V3 q, float R2, T take)
{
    const float4 a = *reinterpret_cast<const float4 *>(m.areas + i);
    if (TEST) {
        float4 t[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) t[k] = m.sampTri[3 * (size_t)i + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float ak = k == 0 ? a.x : k == 1 ? a.y : k == 2 ? a.z : a.w;
            if (ak > 0.0f && tri_d2(v3(t[3 * k].x, t[3 * k].y, t[3 * k].z), v3(t[3 * k + 1].x, t[3 * k + 1].y, t[3 * k + 1].z),
                                    v3(t[3 * k + 2].x, t[3 * k + 2].y, t[3 * k + 2].z), q) <= R2 &&
                !take(i + k, ak))
                return false;
        }
    } else {
        if (a.x > 0.0f && !take(i, a.x)) return false;
        if (a.y > 0.0f && !take(i + 1, a.y)) return false;
        if (a.z > 0.0f && !take(i + 2, a.z)) return false;
        if (a.w > 0.0f && !take(i + 3, a.w)) return false;
    }
    return true;
}

__device__ __forceinline__ int sample_in_sphere3_tree(const DevMesh3 &m, V3 q, float R, float u, float &pdf)
{
    const float R2 = R * R;
    float total = 0.0f;
    auto add = [&](int, float area) {
        total += area;
        return true;
    };
    sweep_in_sphere3(
        m, q, R2, [&](int i) { return sample_group3<true>(m, i, q, R2, add); }, [&](int i) { return sample_group3<false>(m, i, q, R2, add); });
    pdf = 0.0f;
    if (!(total > 0.0f)) return -1;
    const float target = u * total;
    float cum = 0.0f;
    int last = -1;
    auto pick = [&](int i, float area) {
        cum += area;
        last = i;
        return !(target < cum);
    };
    sweep_in_sphere3(
        m, q, R2, [&](int i) { return sample_group3<true>(m, i, q, R2, pick); }, [&](int i) { return sample_group3<false>(m, i, q, R2, pick); });
    const float a = m.areas[last];
    pdf = (a / total) / a;
    return last;
}

// getPerpendicular(Vector3f) + frameFromNormal(Vector3f) + Frame<3>::toWorld
__device__ __forceinline__ V3 frame_to_world(V3 n, float lx, float ly, float lz)
{
    const float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    const uint32_t uyx = (ax - ay) < 0 ? 1u : 0u, uzx = (ax - az) < 0 ? 1u : 0u, uzy = (ay - az) < 0 ? 1u : 0u;
    const uint32_t xm = uyx & uzx, ym = (1u ^ xm) & uzy, zm = 1u ^ (xm | ym);
    const V3 t = normalize3(cross3(n, v3((float)xm, (float)ym, (float)zm))), b = normalize3(cross3(n, t));
    return V3{(t.x * lx + b.x * ly) + n.x * lz, (t.y * lx + b.y * ly) + n.y * lz, (t.z * lx + b.z * ly) + n.z * lz};
}

__device__ __forceinline__ V3 eval_point3(const DevProbe3 &p, int px, int py, int width, int height)
{
    const float ndcx = 2.0f * (float)px / (float)width + -1.0f, ndcy = 2.0f * (float)py / (float)height + -1.0f;
    return V3{p.scale * (ndcx * p.right[0] + ndcy * p.up[0]) + p.pos[0], p.scale * (ndcx * p.right[1] + ndcy * p.up[1]) + p.pos[1],
              p.scale * (ndcx * p.right[2] + ndcy * p.up[2]) + p.pos[2]};
}

constexpr int kStat3Copies = 64;
struct alignas(256) Stats3Dev {
    unsigned long long steps, started, absorbed, truncated, nhits;
};

// source term: dense grid, trilinear (wost3_source_desc)
struct DevSource3 {
    const float *rgb;          // nullptr: no source term
    int32_t nx, ny, nz;
    float sx, sy, sz, ox, oy, oz;
    float intensity;
};

__device__ __forceinline__ void source3_tap(const DevSource3 &s, int i, int j, int k, float (&v)[3])
{
    if (i < 0 || j < 0 || k < 0 || i >= s.nx || j >= s.ny || k >= s.nz) {
        v[0] = v[1] = v[2] = 0.0f;
        return;
    }
    const float *p = s.rgb + 3 * (((size_t)k * s.ny + j) * s.nx + i);
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
}

__device__ __forceinline__ void source3_eval(const DevSource3 &s, V3 q, float (&out)[3])
{
    const float gx = __builtin_fmaf(q.x, s.sx, s.ox), gy = __builtin_fmaf(q.y, s.sy, s.oy), gz = __builtin_fmaf(q.z, s.sz, s.oz);
    const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
    const float u = gx - fx, v = gy - fy, w = gz - fz;
    const int i = (int)fmaxf(fminf(fx, 1e9f), -1e9f), j = (int)fmaxf(fminf(fy, 1e9f), -1e9f), k = (int)fmaxf(fminf(fz, 1e9f), -1e9f);
    float c000[3], c001[3], c010[3], c011[3], c100[3], c101[3], c110[3], c111[3];     // [dk][dj][di]
    source3_tap(s, i, j, k, c000); source3_tap(s, i + 1, j, k, c001);
    source3_tap(s, i, j + 1, k, c010); source3_tap(s, i + 1, j + 1, k, c011);
    source3_tap(s, i, j, k + 1, c100); source3_tap(s, i + 1, j, k + 1, c101);
    source3_tap(s, i, j + 1, k + 1, c110); source3_tap(s, i + 1, j + 1, k + 1, c111);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        // x, then y, then z
        const float a00 = c000[ch] + (c001[ch] - c000[ch]) * u, a01 = c010[ch] + (c011[ch] - c010[ch]) * u;
        const float a10 = c100[ch] + (c101[ch] - c100[ch]) * u, a11 = c110[ch] + (c111[ch] - c110[ch]) * u;
        const float b0 = a00 + (a01 - a00) * v, b1 = a10 + (a11 - a10) * v;
        out[ch] = (b0 + (b1 - b0) * w) * s.intensity;
    }
}

// cube root of x in [0, 1] through the deterministic log / exp (the arithmetic contract's std::cbrt)
__device__ __forceinline__ float cbrt01(float x) { return x > 0.0f ? det_expf(det_logf(x) * (1.0f / 3.0f)) : 0.0f; }

struct Walk3Params {
    DevMesh3 dm, nm;
    DevSettings st;
    DevProbe3 probe;
    DevSource3 src;
    const uint8_t *mask;
    float *field;              // solution / spp at field[(pix - field_base) * 3]
    int32_t field_base, pixel_begin, pixel_end;
    int32_t shard_index, shard_count;
    Stats3Dev *stats;
    uint32_t *cursor;          // next unread pixel slot of the launch
    int32_t tiled;             // the range is a whole frame made of 8x8 tiles: slots follow the tiles
    int32_t wait_weight, trav_burst;
    // NTREE kernels: the Neumann-side tree queries of a step answered by the wave as a whole (closest_silhouette3_wave,
    // ray_closest3_wave); pool_cap tasks per pool and wave, behind the stack columns of the block in LDS
    int32_t coop, pool_cap, stack_words, ray_slot_trigger, cp_slot_trigger;
};

// One lane = one pixel, all its samples one after the other on the pixel's PCG stream (the reference's per-pixel
// order) -- scheduled like the 2-D round kernel: a lane is descending the Dirichlet tree (TRAV), waits with a finished
// query for the rest of its step (WAIT), or wants the next pixel of the solve (REFILL); every trip of the loop runs
// the body more lanes are ready for, persistent blocks drain one pixel cursor.  (The first version ran every lane's
// query to completion in lock step, one launch of pixel-many lanes: 4.7e8 walk-steps/s on a 1280-triangle sphere.)
struct Lane3 {
    int pid, sample, depth;
    V3 p, p_eval, nn;
    float thp;
    bool on_n;
    int32_t hint, hint0;
    Pcg rng;
    float sol[3];
    Closest d0;          // the query of the evaluation point, the same for every sample of the pixel
    bool d0_valid;
    uint32_t c_steps, c_started, c_absorbed, c_truncated, c_nhits;
};

// the rest of a step once the closest Dirichlet triangle is known (`cp`, ignored without that mesh); true = the walk
// has ended (absorbed, no boundary at all), false = L.p is the next point
#ifdef WOST3_PROFILE
#define PROF3_T0() unsigned long long prof_t = __builtin_readcyclecounter()
#define PROF3(k) do { const unsigned long long prof_n = __builtin_readcyclecounter(); if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicAdd(&g_prof3[k], prof_n - prof_t); prof_t = prof_n; } while (0)
#else
#define PROF3_T0() do {} while (0)
#define PROF3(k) do {} while (0)
#endif

// One walk step in three parts around its two tree queries -- the closest silhouette edge after part A, the walker's
// ray after part B.  The kernel answers them on the spot (step3).  Making them states of the lane machine like the
// closest-point descent (one node visit per trip: sil3_visit / ray3_visit, a trip running the body most lanes are
// ready for) was built on these parts and measured: bit-exact, but SLOWER on a 1280-triangle shell (zero flux 660 ->
// 770 ms, emissive 1570 -> 1950 ms at the best scheduler constants of each): four kinds of lanes per wave fill a body
// worse than the longest-lane wait inside the step costs, and the kernel holds both query states (165 VGPRs).
// A: the Dirichlet side.  true = absorbed.
__device__ __forceinline__ bool step3_a(const Walk3Params &P, Lane3 &L, Closest cp, float &R_D)
{
    const bool has_d = P.dm.n_tris > 0;
    const float eps = P.st.eps;
    V3 &p = L.p;
    float &thp = L.thp;
    float (&sol)[3] = L.sol;
    int32_t &hint = L.hint;
    R_D = WOST_INF;
    if (has_d) {
        hint = cp.slot;
        if (L.depth == 0) L.hint0 = cp.slot;
        const float4 a = P.dm.tri[3 * (size_t)cp.slot], b = P.dm.tri[3 * (size_t)cp.slot + 1], c = P.dm.tri[3 * (size_t)cp.slot + 2];
        const V3 p0 = v3(a.x, a.y, a.z), e0 = v3(b.x, b.y, b.z) - p0, e1 = v3(c.x, c.y, c.z) - p0;
        const int side = tri_side(p0, cross3(e0, e1), p);
        float u, v;
        tri_uv(p0, e0, e1, p, u, v);
        R_D = sqrtf(cp.d2);
        if (R_D < eps && u > 0.0f && v > 0.0f && u + v < 1.0f) {
            float col[3];
            const int32_t *tv = P.dm.triVerts + 3 * (size_t)cp.slot;
            surface_color3(P.dm.colors, tv[0], tv[1], tv[2], side, u, v, col);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                col[k] *= P.st.dirichlet_intensity;
                col[k] *= thp;
                sol[k] = col[k] + sol[k];
            }
            ++L.c_absorbed;
            return true;
        }
    }
    return false;
}

// B: the star radius, the source and Neumann samples (their rays inline), the direction of the step.  true = no boundary
// at all (the walk ends); else the walker's ray starts at `cur` along `dir` and is at most R_B long.
template <bool EMISSIVE, bool SOURCE, bool NTREE>
__device__ __forceinline__ bool step3_b(const Walk3Params &P, Lane3 &L, float R_D, float R_N, const LdsColumn &stk, float &R_B_out, V3 &dir_out, V3 &cur_out)
{
    const bool has_n = P.nm.n_tris > 0;
    const float eps = P.st.eps;
    V3 &p = L.p;
    float &thp = L.thp;
    bool &on_n = L.on_n;
    V3 &nn = L.nn;
    float (&sol)[3] = L.sol;
    Pcg &rng = L.rng;
    PROF3_T0();
            float R_B = fmaxf(WOST_R_B_FLOOR, fminf(R_D, R_N));
            R_B *= WOST_R_B_SHRINK;
            if (isinf(R_B)) return true;
            // ---- sampleSource (reference integrator/uniform/integrator.cu:235-316, DIM == 3) ----
            if (SOURCE) {
                V3 sdir;
                float dir_pdf, salpha = 1.0f;
                {
                    const float u1 = pcg_next_float(rng), u2 = pcg_next_float(rng);
                    float c, s;
                    sincos_2pi(u2, c, s);
                    if (on_n) {
                        const float z = u1, r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
                        sdir = frame_to_world(nn, r * c, r * s, z);
                        dir_pdf = 1.0f / WOST_2PI;
                        salpha = 0.5f;
                    } else {
                        const float z = 1 - 2 * u1, r = sqrtf(1 - z * z);
                        sdir = v3(r * c, r * s, z);
                        dir_pdf = 1.0f / WOST_4PI;
                    }
                }
                // how far the straight line stays inside the star-shaped region (:279-292)
                float dist = R_B;
                if (has_n) {
                    float t;
                    int hi;
                    if (ray_closest3<NTREE>(P.nm, v3(p.x + eps * sdir.x, p.y + eps * sdir.y, p.z + eps * sdir.z), sdir, dist, t, hi, stk)) dist = fminf(t, dist);
                }
                // HarmonicGreenBall<3>::sample (util/green.h:101-116): closed form, two draws
                const float g1 = pcg_next_float(rng), g2 = pcg_next_float(rng);
                float gc, gs;
                sincos_2pi(g2, gc, gs);
                float r = (1.0f + sqrtf(1.0f - cbrt01(g1 * g1)) * gc) * R_B / 2.0f;
                r = fmaxf(1e-4f, r);                                            // ELAINA_GREEN_FUNC_R_CLAMP
                if (r > R_B) r = R_B / 2.0f;
                if (r <= dist) {
                    float f[3];
                    source3_eval(P.src, v3(p.x + r * sdir.x, p.y + r * sdir.y, p.z + r * sdir.z), f);
                    const float norm = R_B * R_B / 6.0f;
                    const float c1 = (1.0f / WOST_4PI) / (r * r), c2 = dir_pdf / (r * r);     // conditionalSampleSpherePDF<3>
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float col = thp * f[k] * norm * c1 / c2 / salpha;
                        sol[k] = col + sol[k];
                    }
                }
            }
            PROF3(1);
            // ---- sampleNeumann: three draws whether or not the boundary emits ----
            if (has_n) {
                const float u0 = pcg_next_float(rng), u1 = pcg_next_float(rng), u2 = pcg_next_float(rng);
                if (EMISSIVE) {
                    float pdf;
                    const int oi = (NTREE && P.nm.obox_levels > 0) ? sample_in_sphere3_tree(P.nm, p, R_B, u0, pdf) : sample_in_sphere3_flat(P.nm, p, R_B, u0, pdf);
                    if (oi != -1 && pdf > 0) {
                        const DevTri S = P.nm.flat[oi];
                        const V3 s0 = ld3(S.p0), s1 = ld3(S.p1), s2 = ld3(S.p2);
                        const float su = sqrtf(u1), b1 = u2 * su, b0 = 1.0f - su, b2 = 1.0f - b0 - b1;
                        const V3 sp = v3((s0.x * b0 + s1.x * b1) + s2.x * b2, (s0.y * b0 + s1.y * b1) + s2.y * b2,
                                         (s0.z * b0 + s1.z * b1) + s2.z * b2);
                        const V3 rv = sp - p;
                        const float r = sqrtf(dot3(rv, rv));
                        if (r < R_B && r > 0) {
                            V3 o = p;
                            if (on_n) o = v3(p.x + eps * nn.x, p.y + eps * nn.y, p.z + eps * nn.z);
                            V3 rd = sp - o;
                            const float cd = sqrtf(dot3(rd, rd));
                            if (cd > 0) { rd.x /= cd; rd.y /= cd; rd.z /= cd; }
                            if (!ray_any3<NTREE>(P.nm, o, rd, cd - eps, stk)) {
                                int side = tri_side(s0, ld3(S.nraw), p);
                                float uu, vv;
                                tri_uv(s0, s1 - s0, s2 - s0, sp, uu, vv);
                                if (on_n) {
                                    const float dn = dot3(ld3(S.n), nn);
                                    side = (0.0f < dn) - (dn < 0.0f);
                                }
                                if (side != 0) {
                                    float col[3];
                                    const int32_t *tv = P.nm.flatVerts + 3 * (size_t)oi;
                                    surface_color3(P.nm.colors, tv[0], tv[1], tv[2], side, uu, vv, col);
                                    const float alpha = on_n ? 0.5f : 1.0f;
                                    const float G = (1.0f / r - 1.0f / R_B) / WOST_4PI;
#pragma unroll
                                    for (int k = 0; k < 3; ++k) {
                                        col[k] *= P.st.neumann_intensity;
                                        col[k] *= thp * G / alpha / pdf;
                                        sol[k] = -col[k] + sol[k];
                                    }
                                }
                            }
                        }
                    }
                }
            }
            PROF3(2);
            // ---- oneStepWalk ----
            V3 dir, cur = p;
            float pdf, alpha = 1.0f;
            {
                const float u1 = pcg_next_float(rng), u2 = pcg_next_float(rng);
                float c, s;
                sincos_2pi(u2, c, s);
                if (on_n) {
                    const float z = u1, r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
                    dir = frame_to_world(nn, r * c, r * s, z);
                    pdf = 1.0f / WOST_2PI;
                    alpha = 0.5f;
                    cur = v3(p.x + eps * nn.x, p.y + eps * nn.y, p.z + eps * nn.z);
                } else {
                    const float z = 1 - 2 * u1, r = sqrtf(1 - z * z);
                    dir = v3(r * c, r * s, z);
                    pdf = 1.0f / WOST_4PI;
                }
            }
    (void)pdf; (void)alpha;
    R_B_out = R_B; dir_out = dir; cur_out = cur;
    return false;
}

// C: where the ray ended.
__device__ __forceinline__ void step3_c(const Walk3Params &P, Lane3 &L, float R_B, V3 dir, V3 cur, bool hit, float t, int hi)
{
    V3 nxt = v3(L.p.x + R_B * dir.x, L.p.y + R_B * dir.y, L.p.z + R_B * dir.z);
    V3 hn = v3(0.0f, 0.0f, 0.0f);
    if (hit) {
        hn = ld3(P.nm.flat[hi].n);
        if (dot3(hn, dir) > 0) hn = v3(-hn.x, -hn.y, -hn.z);
        nxt = v3(cur.x + t * dir.x, cur.y + t * dir.y, cur.z + t * dir.z);
        ++L.c_nhits;
    }
    // uniformSampleSphere / Hemisphere pdf and the boundary factor of the step that was taken from L.on_n
    const float pdf = L.on_n ? 1.0f / WOST_2PI : 1.0f / WOST_4PI, alpha = L.on_n ? 0.5f : 1.0f;
    L.thp = L.thp / pdf / alpha / WOST_4PI;
    L.p = nxt; L.on_n = hit; L.nn = hn;
}

// the whole step with both queries answered on the spot; true = the walk has ended
template <bool EMISSIVE, bool SOURCE, bool NTREE>
__device__ __forceinline__ bool step3(const Walk3Params &P, Lane3 &L, Closest cp, const LdsColumn &stk)
{
    float R_D, R_B;
    if (step3_a(P, L, cp, R_D)) return true;
    float R_N = WOST_INF;
    if (P.nm.n_tris > 0) R_N = closest_silhouette3<NTREE>(P.nm, L.p, R_D, stk);
    V3 dir, cur;
    if (step3_b<EMISSIVE, SOURCE, NTREE>(P, L, R_D, R_N, stk, R_B, dir, cur)) return true;
    bool hit = false;
    float t = 0.0f;
    int hi = -1;
    if (P.nm.n_tris > 0) hit = ray_closest3<NTREE>(P.nm, cur, dir, R_B, t, hi, stk);
    step3_c(P, L, R_B, dir, cur, hit, t, hi);
    return false;
}

constexpr int kWalk3Threads = 256;

// The closest triangle to q -- the same point in all 64 lanes -- by a scan of every slot of the leaf level, the lanes sharing
// them: the exact distances of a leaf visit, the lowest ORIGINAL index among equal ones.  For a walker that strayed so far
// (DevMesh3::huge2) that all triangles lie within the rounding of one another, where the descent -- its boxes pruned with a
// relative slack -- opens every box, one lane and one node at a time.  Returns the same answer in every lane.
__device__ __forceinline__ Closest closest_triangle_wave(const DevMesh3 &m, V3 q)
{
    const int lane = threadIdx.x & 63;
    const int n_slots = 4 << (2 * m.levels);
    float bd = WOST_INF;
    int32_t bs = -1, bo = WOST_FAR_INDEX;
    for (int k = lane; k < n_slots; k += 64) {
        const int32_t o = m.triOrig[k];
        if (o == WOST_FAR_INDEX) continue;
        const float4 a = m.tri[3 * (size_t)k], b = m.tri[3 * (size_t)k + 1], c = m.tri[3 * (size_t)k + 2];
        const float d = tri_d2(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), q);
        if (d < bd || (d == bd && o < bo)) {
            bd = d; bs = k; bo = o;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(bd, off);
        const int32_t os = __shfl_xor(bs, off), oo = __shfl_xor(bo, off);
        if (od < bd || (od == bd && oo < bo)) {
            bd = od; bs = os; bo = oo;
        }
    }
    return Closest{bd, bs};
}

// WAVE = true: the closest-point queries are answered by the wave as a whole as well (closest_triangle_pool) -- every trip of
// the loop is then "all queries of the wave, then one step for every walker": no lane waits for another's descent.
template <bool EMISSIVE, bool SOURCE, bool NTREE, bool WAVE = false>
#ifndef WOST3_WAVES
#define WOST3_WAVES 1       // waves per SIMD walk3_kernel is compiled for (tuning builds override it)
#endif
__global__ __launch_bounds__(kWalk3Threads, WOST3_WAVES) void walk3_kernel(Walk3Params P)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk(lds_stack + threadIdx.x, blockDim.x);
    // the task pools of this wave (closest_silhouette3_wave & co.), behind the stack columns
    uint32_t *const pool_mem = lds_stack + P.stack_words + (threadIdx.x >> 6) * (2 * P.pool_cap + kPool3OwnerWords);
    const WavePool3 W{pool_mem + kPool3OwnerWords, pool_mem + kPool3OwnerWords + P.pool_cap, pool_mem, P.pool_cap};
#ifdef WOST3_PROFILE
    const unsigned long long prof_begin = __builtin_readcyclecounter();
#endif
    const int lane = threadIdx.x & 63;
    const bool has_d = P.dm.n_tris > 0;
    enum { MODE_TRAV = 1, MODE_QUERY = 2, MODE_WAIT = 3, MODE_DONE = 4, MODE_REFILL = 5, MODE_HUGE = 7 };
    int mode = MODE_REFILL;
    Lane3 L{};
    L.rng = Pcg{0, 1};
    Trav T = trav_begin(Closest{WOST_INF, -1});
    uint32_t pool_next = 0, pool_end = 0;
    uint32_t t_steps = 0, t_started = 0, t_absorbed = 0, t_truncated = 0, t_nhits = 0;
    const uint32_t n_slots = (uint32_t)(P.pixel_end - P.pixel_begin);

    // start the query of a step (or serve it from the cache of the evaluation point)
    auto begin_step = [&]() {
        ++L.c_steps;
        if (!has_d) {
            T.best = Closest{WOST_INF, -1};
            mode = MODE_WAIT;
        } else if (L.depth == 0 && L.d0_valid) {
            T.best = L.d0;
            mode = MODE_WAIT;
        } else {
            T = trav_begin(Closest{WOST_INF, -1});
            if (L.hint >= 0 && P.dm.triOrig[L.hint] != WOST_FAR_INDEX) {
                const float4 a = P.dm.tri[3 * (size_t)L.hint], b = P.dm.tri[3 * (size_t)L.hint + 1], c = P.dm.tri[3 * (size_t)L.hint + 2];
                T.best = Closest{tri_d2(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), L.p), L.hint};
                T.best_orig = P.dm.triOrig[L.hint];
            }
            // a walker that strayed so far that the whole mesh ties within rounding: answered by the wave (main loop)
            mode = T.best.d2 > P.dm.huge2 ? MODE_HUGE : (WAVE ? MODE_QUERY : MODE_TRAV);
        }
    };
    auto begin_sample = [&]() {
        L.p = L.p_eval;
        L.thp = 1.0f;
        L.on_n = false;
        L.nn = v3(0.0f, 0.0f, 0.0f);
        ++L.c_started;
        L.hint = L.hint0;
        L.depth = 0;
        begin_step();
    };

    for (;;) {
        const unsigned long long need = __ballot(mode == MODE_REFILL);
        if (need) {
            const uint32_t needed = (uint32_t)__popcll(need), avail = pool_end - pool_next;
            uint32_t fresh_base = 0;
            if (needed > avail) {
                if (lane == 0) fresh_base = atomicAdd(P.cursor, 64u);
                fresh_base = __shfl(fresh_base, 0);
            }
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            const uint32_t s2 = rank < avail ? pool_next + rank : fresh_base + (rank - avail);
            if (needed > avail) {
                pool_next = fresh_base + (needed - avail);
                pool_end = fresh_base + 64u;
            } else {
                pool_next += needed;
            }
            if (mode == MODE_REFILL) {
                if (s2 >= n_slots) {
                    mode = MODE_DONE;
                } else {
                    // slots walk the frame in 8x8 tiles when the range is the whole tiled frame (neighbouring walkers in a wave)
                    int pid = P.pixel_begin + (int)s2;
                    if (P.tiled) {
                        const int tiles_x = P.st.width >> 3, tile = pid >> 6, in_tile = pid & 63;
                        pid = ((tile / tiles_x) * 8 + (in_tile >> 3)) * P.st.width + (tile % tiles_x) * 8 + (in_tile & 7);
                    }
                    const int px = pid % P.st.width, py = pid / P.st.width;
                    const int tile = (py >> 3) * ((P.st.width + 7) >> 3) + (px >> 3);
                    if ((tile % P.shard_count) == P.shard_index) {
                        const bool masked = P.mask != nullptr && P.mask[pid] == 0;
                        if (masked || P.st.spp <= 0) {
                            float *f = P.field + 3 * (size_t)(pid - P.field_base);
                            const float spp = (float)P.st.spp;
                            f[0] = 0.0f / spp; f[1] = 0.0f / spp; f[2] = 0.0f / spp;
                        } else {
                            // fold the counters of the previous pixel into the lane totals (16-bit-safe: per pixel)
                            t_steps += L.c_steps; t_started += L.c_started; t_absorbed += L.c_absorbed; t_truncated += L.c_truncated; t_nhits += L.c_nhits;
                            L = Lane3{};
                            L.pid = pid;
                            L.rng = Pcg{0, 1};
                            pcg_seed_pixel(L.rng, pid, P.st.width);
                            L.p_eval = eval_point3(P.probe, px, py, P.st.width, P.st.height);
                            L.hint = L.hint0 = -1;
                            L.sample = 0;
                            begin_sample();
                        }
                    }
                    // a pixel of another shard or a masked one: the lane asks again on the next trip
                }
            }
        }
        {
            unsigned long long hb = __ballot(mode == MODE_HUGE);
            while (hb) {
                const int src = __builtin_ctzll(hb);
                const Closest r = closest_triangle_wave(P.dm, v3(__shfl(L.p.x, src), __shfl(L.p.y, src), __shfl(L.p.z, src)));
                if (lane == src) {
                    T.best = r;
                    mode = MODE_WAIT;
                }
                hb &= hb - 1;
            }
        }
        if (WAVE) {
            const bool asks = mode == MODE_QUERY;
            if (__ballot(asks)) {
                const Closest r = closest_triangle_pool(P.dm, L.p, T.best, asks, W, stk, P.cp_slot_trigger);
                if (asks) {
                    T.best = r;
                    mode = MODE_WAIT;
                }
            }
        }
        const int n_trav = __popcll(__ballot(mode == MODE_TRAV));
        const int n_wait = __popcll(__ballot(mode == MODE_WAIT));
        if (n_trav + n_wait == 0) {
            if (__ballot(mode == MODE_REFILL)) continue;
            break;
        }
        PROF3_T0();
        if (n_wait * P.wait_weight >= n_trav * 8) {
#ifdef WOST3_PROFILE
            if (lane == 0) { atomicAdd(&g_prof3[8], 1ull); atomicAdd(&g_prof3[9], (unsigned long long)n_wait); }
#endif
            const bool stepping = mode == MODE_WAIT;
            bool ended = false;
            if (stepping && has_d && L.depth == 0 && !L.d0_valid) {
                L.d0 = T.best;
                L.d0_valid = true;
            }
            if (NTREE && P.coop) {
                // the step in its three parts (step3), the two tree queries between them answered by all 64 lanes together
                float R_D = WOST_INF, R_B = 0.0f;
                V3 dir = v3(0.0f, 0.0f, 0.0f), cur = dir;
                bool mid = false, go = false;
                if (stepping) {
                    ended = step3_a(P, L, T.best, R_D);
                    mid = !ended;
                }
                const float R_N = closest_silhouette3_wave(P.nm, L.p, R_D, mid, W, stk);
                if (mid) {
                    ended = step3_b<EMISSIVE, SOURCE, NTREE>(P, L, R_D, R_N, stk, R_B, dir, cur);
                    go = !ended;
                }
                float t = 0.0f;
                int hi = -1;
                const bool hit = ray_closest3_wave(P.nm, cur, dir, R_B, go, t, hi, W, stk, P.ray_slot_trigger);
                if (go) step3_c(P, L, R_B, dir, cur, hit, t, hi);
            } else if (stepping) {
                ended = step3<EMISSIVE, SOURCE, NTREE>(P, L, T.best, stk);
            }
            if (stepping) {
                if (!ended) {
                    ++L.depth;
                    if (L.depth == P.st.max_depth) {
                        ++L.c_truncated;
                        ended = true;
                    }
                }
                if (!ended) {
                    begin_step();
                } else if (++L.sample < P.st.spp) {
                    begin_sample();
                } else {
                    float *f = P.field + 3 * (size_t)(L.pid - P.field_base);
                    const float spp = (float)P.st.spp;
                    f[0] = L.sol[0] / spp; f[1] = L.sol[1] / spp; f[2] = L.sol[2] / spp;
                    mode = MODE_REFILL;
                }
            }
            PROF3(4);
        } else {
            for (int b = 0; b < P.trav_burst; ++b) {
                if (mode == MODE_TRAV) {
                    if (!trav_visit3(P.dm, L.p, T, stk)) mode = MODE_WAIT;
                }
            }
            PROF3(5);
#ifdef WOST3_PROFILE
            if (lane == 0) { atomicAdd(&g_prof3[10], 1ull); atomicAdd(&g_prof3[11], (unsigned long long)n_trav); }
#endif
        }
    }
#ifdef WOST3_PROFILE
    if (lane == 0) atomicAdd(&g_prof3[6], __builtin_readcyclecounter() - prof_begin);
#endif
    t_steps += L.c_steps; t_started += L.c_started; t_absorbed += L.c_absorbed; t_truncated += L.c_truncated; t_nhits += L.c_nhits;
    uint32_t v[5] = {t_steps, t_started, t_absorbed, t_truncated, t_nhits};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        uint32_t x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        v[k] = x;
    }
    if (lane == 0) {
        Stats3Dev *st = P.stats + (blockIdx.x & (kStat3Copies - 1));
        if (v[0]) atomicAdd(&st->steps, (unsigned long long)v[0]);
        if (v[1]) atomicAdd(&st->started, (unsigned long long)v[1]);
        if (v[2]) atomicAdd(&st->absorbed, (unsigned long long)v[2]);
        if (v[3]) atomicAdd(&st->truncated, (unsigned long long)v[3]);
        if (v[4]) atomicAdd(&st->nhits, (unsigned long long)v[4]);
    }
}

// ---- batch queries (tests, SDF-style renders) ----------------------------------------------------
__global__ __launch_bounds__(256) void closest_point3_kernel(DevMesh3 m, const float *pts, int n, int32_t *out_idx, float *out_dist,
                                                             float *out_uv, int32_t *out_side)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk(lds_stack + threadIdx.x, blockDim.x);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 q = v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    const Closest cp = closest_triangle(m, q, -1, stk);
    const float4 a = m.tri[3 * (size_t)cp.slot], b = m.tri[3 * (size_t)cp.slot + 1], c = m.tri[3 * (size_t)cp.slot + 2];
    const V3 p0 = v3(a.x, a.y, a.z), e0 = v3(b.x, b.y, b.z) - p0, e1 = v3(c.x, c.y, c.z) - p0;
    if (out_idx) out_idx[i] = m.triOrig[cp.slot];
    if (out_dist) out_dist[i] = sqrtf(cp.d2);
    if (out_uv) tri_uv(p0, e0, e1, q, out_uv[2 * i], out_uv[2 * i + 1]);
    if (out_side) out_side[i] = tri_side(p0, cross3(e0, e1), q);
}

// debug channels at the evaluation points of the frame
__global__ __launch_bounds__(256) void render3_sdf_kernel(DevMesh3 m, DevProbe3 probe, int width, int height, int silhouette, float *out)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk(lds_stack + threadIdx.x, blockDim.x);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    const V3 q = eval_point3(probe, i % width, i / width, width, height);
    float d = WOST_INF;
    if (m.n_tris > 0) {
        if (!silhouette) d = sqrtf(closest_triangle(m, q, -1, stk).d2);
        else d = m.n_tris > WOST3_FLAT_MAX ? closest_silhouette3_tree(m, q, WOST_INF, stk) : closest_silhouette3_flat(m, q, WOST_INF);
    }
    out[i] = d;
}

__global__ __launch_bounds__(256) void render3_source_kernel(DevSource3 src, DevProbe3 probe, int width, int height, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    float f[3] = {0.0f, 0.0f, 0.0f};
    if (src.rgb) source3_eval(src, eval_point3(probe, i % width, i / width, width, height), f);
    out[3 * (size_t)i] = f[0]; out[3 * (size_t)i + 1] = f[1]; out[3 * (size_t)i + 2] = f[2];
}

template <bool NTREE>
__global__ __launch_bounds__(256) void silhouette3_kernel(DevMesh3 m, const float *pts, const float *rmax, int n, float *out)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk(lds_stack + threadIdx.x, blockDim.x);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = m.n_tris > 0 ? closest_silhouette3<NTREE>(m, v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), rmax ? rmax[i] : WOST_INF, stk) : WOST_INF;
}

template <bool NTREE>
__global__ __launch_bounds__(256) void ray3_kernel(DevMesh3 m, const float *o, const float *d, const float *tmax, int n, int32_t *out_hit,
                                                   float *out_t, int32_t *out_idx)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk(lds_stack + threadIdx.x, blockDim.x);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t;
    int idx;
    const bool hit = ray_closest3<NTREE>(m, v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i], t, idx, stk);
    out_hit[i] = hit ? 1 : 0;
    out_t[i] = t;
    out_idx[i] = idx;
}

// ---- host: LBVH over triangles ----------------------------------------------------------------------
struct HostMesh3 {
    int32_t n_tris = 0, n_edges = 0, levels = 1, first_leaf = 1;
    bool emissive = false;
    std::vector<float> nodes, tri, colors, cones, slotEdges;
    float ext = 0.0f;                 // largest coordinate
    std::vector<int32_t> triOrig, triVerts, flatVerts;
    std::vector<float> obox;          // index-ordered run boxes (emissive meshes above the flat limit)
    int32_t obox_off[12] = {0}, obox_levels = 0;
    std::vector<DevTri> flat;
    std::vector<DevEdge3> edges;
};

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
            return x.t < y.t;
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
                double ax[3] = {0.0, 0.0, 0.0};
                for (size_t i = 0; i < a.nrm.size(); i += 3)
                    for (int c = 0; c < 3; ++c) ax[c] += a.nrm[i + c];
                const double al = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
                if (!(al > 1e-9 * (double)(a.nrm.size() / 3))) continue;
                for (int c = 0; c < 3; ++c) ax[c] /= al;
                double cmin = 1.0;
                for (size_t i = 0; i < a.nrm.size(); i += 3) {
                    const double l = std::sqrt(a.nrm[i] * a.nrm[i] + a.nrm[i + 1] * a.nrm[i + 1] + a.nrm[i + 2] * a.nrm[i + 2]);
                    cmin = std::min(cmin, (ax[0] * a.nrm[i] + ax[1] * a.nrm[i + 1] + ax[2] * a.nrm[i + 2]) / l);
                }
                const double half = std::acos(std::max(-1.0, std::min(1.0, cmin))) + 1e-4;
                if (half >= 0.5 * M_PI - 1e-3) continue;
                // the centre the kernel uses: the middle of the child's box, in the kernel's float arithmetic
                float cf[3];
                for (int c = 0; c < 3; ++c) cf[c] = 0.5f * (nd[4 * c + j] + nd[12 + 4 * c + j]);
                double rad = 0.0;
                for (size_t i = 0; i < a.pts.size(); i += 3)
                    rad = std::max(rad, std::sqrt((a.pts[i] - cf[0]) * (a.pts[i] - cf[0]) + (a.pts[i + 1] - cf[1]) * (a.pts[i + 1] - cf[1]) +
                                                  (a.pts[i + 2] - cf[2]) * (a.pts[i + 2] - cf[2])));
                cn[0 + j] = (float)ax[0]; cn[4 + j] = (float)ax[1]; cn[8 + j] = (float)ax[2];
                cn[12 + j] = (float)std::cos(half); cn[16 + j] = (float)std::sin(half);
                cn[20 + j] = (float)(rad * (1.0 + 1e-6) + (double)pad + 2.0 * (double)WOST_SIL_PRECISION);
            }
    }
    return 0;
}

struct DeviceMesh3 {
    DevMesh3 view{};
    HostMesh3 host;
    std::vector<void *> allocs;
};

template <class T>
static hipError_t upload3(std::vector<void *> &allocs, const T *src, size_t count, const T **dst)
{
    *dst = nullptr;
    if (count == 0) return hipSuccess;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) return e;
    allocs.push_back(p);
    *dst = reinterpret_cast<const T *>(p);
    return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
}

// ---- von Mises-Fisher lobe on the sphere (reference util/vmf.h:21-70, the Jakob [2012] form; the lobes of
// GuidedIntegrator<3>'s mixture -- that integrator is not built, the distribution is its first piece) ----------------
constexpr float kVmfEpsilon = 1e-5f;   // M_EPSILON, core/math/include/krrmath/constants.h

// VMF::eval(cosTheta)
__device__ __forceinline__ float vmf_eval(float kappa, float cos_theta)
{
    if (kappa < kVmfEpsilon) return 1.0f / WOST_4PI;
    return det_expf(kappa * fminf(0.0f, cos_theta - 1.0f)) * kappa / (WOST_2PI * (1.0f - det_expf(-2.0f * kappa)));
}

// VMF::sample(sampler, mu): two draws, the lobe about +z turned into the frame of mu
__device__ __forceinline__ V3 vmf_sample(float kappa, V3 mu, Pcg &rng)
{
    const float u0 = pcg_next_float(rng), u1 = pcg_next_float(rng);
    float c, s;
    sincos_2pi(u1, c, s);
    V3 local;
    if (kappa < kVmfEpsilon) {
        const float z = 1 - 2 * u0, r = sqrtf(1 - z * z);               // uniformSampleSphere<3>
        local = v3(r * c, r * s, z);
    } else {
        const float cos_theta = 1.0f + det_logf(1.0f + (-u0 + det_expf(-2.0f * kappa) * u0)) / kappa;
        const float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
        local = v3(c * sin_theta, s * sin_theta, cos_theta);
    }
    return frame_to_world(mu, local.x, local.y, local.z);
}

__global__ __launch_bounds__(256) void vmf_eval_kernel(const float *kappa, const float *cos_theta, int n, float *pdf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pdf[i] = vmf_eval(kappa[i], cos_theta[i]);
}

__global__ __launch_bounds__(256) void vmf_sample_kernel(const float *kappa, const float *mu, const uint64_t *seed, int n, int per_point, float *dirs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pcg rng{0, 1};
    pcg_set_seed(rng, seed[i], 1);
    const V3 m = v3(mu[3 * i], mu[3 * i + 1], mu[3 * i + 2]);
    for (int k = 0; k < per_point; ++k) {
        const V3 w = vmf_sample(kappa[i], m, rng);
        float *o = dirs + 3 * ((size_t)i * per_point + k);
        o[0] = w.x; o[1] = w.y; o[2] = w.z;
    }
}

// ---- VMM<3,8>: the mixture of eight vMF lobes (reference integrator/guided/distribution.h:279-436, train.h:60-105 and
// 492-553 with common3d: 5 numbers per lobe -- lambda, kappa, mean vector -- and the selection logit = 41 outputs) ------
constexpr int kVmm3Lobes = 8;
struct Vmm3 {
    float lambda[kVmm3Lobes], kappa[kVmm3Lobes], weight[kVmm3Lobes], total;
    V3 mu[kVmm3Lobes], mo[kVmm3Lobes];
};

__device__ __forceinline__ float clamp_act(float v) { return fmaxf(fminf(v, 15.0f), -10.0f); }

__device__ __forceinline__ void vmm3_build(Vmm3 &m, const float *data)
{
    m.total = 0.0f;
#pragma unroll
    for (int i = 0; i < kVmm3Lobes; ++i) {
        const float *d = data + 5 * i;
        m.lambda[i] = det_expf(clamp_act(d[0]));
        m.kappa[i] = det_expf(clamp_act(d[1]));
        // Eigen normalized(): v / sqrt(z) when z = squaredNorm > 0, else v unchanged
        const float z = (d[2] * d[2] + d[3] * d[3]) + d[4] * d[4], n = sqrtf(z);
        m.mo[i] = v3(d[2], d[3], d[4]);
        m.mu[i] = z > 0.0f ? v3(d[2] / n, d[3] / n, d[4] / n) : m.mo[i];
        m.total += m.lambda[i];
    }
#pragma unroll
    for (int i = 0; i < kVmm3Lobes; ++i) m.weight[i] = m.lambda[i] / m.total;
}

__device__ __forceinline__ float vmm3_lobe_pdf(const Vmm3 &m, int i, V3 w)
{
    return vmf_eval(m.kappa[i], (w.x * m.mu[i].x + w.y * m.mu[i].y) + w.z * m.mu[i].z);
}

__device__ __forceinline__ float vmm3_pdf(const Vmm3 &m, V3 w)
{
    float val = 0.0f;
#pragma unroll
    for (int i = 0; i < kVmm3Lobes; ++i) val += m.weight[i] * vmm3_lobe_pdf(m, i, w);
    return val;
}

// one draw picks the lobe, two more the direction
__device__ __forceinline__ V3 vmm3_sample(const Vmm3 &m, Pcg &rng)
{
    float u = pcg_next_float(rng);
    int pick = 0;
    bool done = false;
#pragma unroll
    for (int i = 0; i < kVmm3Lobes; ++i) {
        if (!done && u < m.weight[i]) { pick = i; done = true; }
        if (!done) u -= m.weight[i];
    }
    float kap = m.kappa[0];
    V3 mu = m.mu[0];
#pragma unroll
    for (int i = 1; i < kVmm3Lobes; ++i)
        if (pick == i) { kap = m.kappa[i]; mu = m.mu[i]; }
    return vmf_sample(kap, mu, rng);
}

__global__ __launch_bounds__(256) void vmm3_pdf_sample_kernel(const float *raw, const float *wi, const uint64_t *seed, int n, float *pdf, float *dir)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Vmm3 m;
    vmm3_build(m, raw + 40 * (size_t)i);
    if (pdf) pdf[i] = vmm3_pdf(m, v3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]));
    if (dir) {
        Pcg rng{0, 1};
        pcg_set_seed(rng, seed[i], 1);
        const V3 w = vmm3_sample(m, rng);
        dir[3 * i] = w.x; dir[3 * i + 1] = w.y; dir[3 * i + 2] = w.z;
    }
}

// compute_dL_doutput_divergence with GuidedOutput = common3d around VMM<3,N>::gradients_probability
__global__ __launch_bounds__(256) void vmm3_loss_gradients_kernel(const float *raw, const float *dir, const float *li, const float *dir_pdf,
                                                                  const unsigned char *on_neumann, const float *normal, int n, float loss_scale,
                                                                  float *dl_draw, float *likelihood)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float eps = 1e-5f;
    const float scale = loss_scale / (float)n;
    const float *data = raw + 41 * (size_t)t;
    float *grad = dl_draw + 41 * (size_t)t;
    Vmm3 m;
    vmm3_build(m, data);
    const V3 w = v3(dir[3 * t], dir[3 * t + 1], dir[3 * t + 2]);
    const bool on_n = on_neumann ? on_neumann[t] != 0 : false;
    V3 r = v3(0.0f, 0.0f, 0.0f);
    if (on_n) {
        const V3 nn = v3(normal[3 * t], normal[3 * t + 1], normal[3 * t + 2]);
        const float d = (w.x * nn.x + w.y * nn.y) + w.z * nn.z;
        r = v3(w.x - 2 * d * nn.x, w.y - 2 * d * nn.y, w.z - 2 * d * nn.z);
    }
    float pk[kVmm3Lobes], pkr[kVmm3Lobes];
#pragma unroll
    for (int k = 0; k < kVmm3Lobes; ++k) {
        pk[k] = vmm3_lobe_pdf(m, k, w);
        pkr[k] = on_n ? vmm3_lobe_pdf(m, k, r) : 0.0f;
    }
    float probability = 0.0f;
    float g5[kVmm3Lobes][5];
#pragma unroll
    for (int sg = 0; sg < kVmm3Lobes; ++sg) {
        const float lambda = m.lambda[sg], kappa = m.kappa[sg];
        const float ox = m.mo[sg].x, oy = m.mo[sg].y, oz = m.mo[sg].z;
        const V3 mu = m.mu[sg];
        const float vmf = pk[sg];
        probability += m.weight[sg] * vmf;
        float vmfr = 0.0f;
        if (on_n) { vmfr = pkr[sg]; probability += m.weight[sg] * vmfr; }
        float dF_dlambda = (vmf + vmfr) * (m.total - lambda) / (m.total * m.total);
#pragma unroll
        for (int k = 0; k < kVmm3Lobes; ++k) {
            if (k == sg) continue;
            dF_dlambda -= m.weight[k] / m.total * pk[k];
            if (on_n) dF_dlambda -= m.weight[k] / m.total * pkr[k];
        }
        float ik;
        if (kappa < 1) ik = 0.000962f + -0.344883f * kappa + 0.030147f * (kappa * kappa);
        else ik = 1 / kappa - (1 + det_expf(-2 * kappa)) / (1 - det_expf(-2 * kappa));
        float dF_dkappa = m.weight[sg] * vmf * ((w.x * mu.x + w.y * mu.y + w.z * mu.z) + ik);
        if (on_n) dF_dkappa += m.weight[sg] * vmfr * ((r.x * mu.x + r.y * mu.y + r.z * mu.z) + ik);
        const float n2 = (ox * ox + oy * oy) + oz * oz;
        float denom = n2 * sqrtf(n2);
        if (denom < eps) denom = eps;
        const float x = w.x, y = w.y, z = w.z, xr = r.x, yr = r.y, zr = r.z;
        float dF_dx = m.weight[sg] * vmf * kappa * (-ox * oy * y - ox * oz * z + (oy * oy) * x + (oz * oz) * x) / denom;
        if (on_n) dF_dx += m.weight[sg] * vmfr * kappa * (-ox * oy * yr - ox * oz * zr + (oy * oy) * xr + (oz * oz) * xr) / denom;
        float dF_dy = m.weight[sg] * vmf * kappa * (-ox * oy * x - oy * oz * z + (ox * ox) * y + (oz * oz) * y) / denom;
        if (on_n) dF_dy += m.weight[sg] * vmfr * kappa * (-ox * oy * xr - oy * oz * zr + (ox * ox) * yr + (oz * oz) * yr) / denom;
        float dF_dz = m.weight[sg] * vmf * kappa * (-ox * oz * x - oy * oz * y + (ox * ox) * z + (oy * oy) * z) / denom;
        if (on_n) dF_dz += m.weight[sg] * vmfr * kappa * (-ox * oz * xr - oy * oz * yr + (ox * ox) * zr + (oy * oy) * zr) / denom;
        g5[sg][0] = dF_dlambda; g5[sg][1] = dF_dkappa; g5[sg][2] = dF_dx; g5[sg][3] = dF_dy; g5[sg][4] = dF_dz;
    }
    const float Li = li[t];
    const float dirPdf = dir_pdf[t] + eps;
    const float guidePdf = probability + eps;
    const float prefix = -Li / dirPdf / guidePdf * scale;
    if (likelihood) likelihood[t] = -Li / dirPdf * det_logf(guidePdf);
#pragma unroll
    for (int sg = 0; sg < kVmm3Lobes; ++sg) {
        grad[5 * sg + 0] = prefix * g5[sg][0] * det_expf(clamp_act(data[5 * sg + 0]));
        grad[5 * sg + 1] = prefix * g5[sg][1] * det_expf(clamp_act(data[5 * sg + 1]));
        grad[5 * sg + 2] = prefix * g5[sg][2];
        grad[5 * sg + 3] = prefix * g5[sg][3];
        grad[5 * sg + 4] = prefix * g5[sg][4];
    }
    const float e = 0.2f;
    const float uni = on_n ? 1.0f / WOST_2PI : 1.0f / WOST_4PI;
    const float sgm = 1.0f / (1.0f + det_expf(-data[40]));
    grad[40] = scale * (-e) * Li * (guidePdf - uni) / (dirPdf * dirPdf) * (sgm * (1 - sgm));
}

}  // namespace wost

using namespace wost;

struct wost3_context {
    int device = 0;
    wost_settings settings{};
    DevSettings dst{};
    DevProbe3 probe{};
    DeviceMesh3 dm, nm;
    uint8_t *mask = nullptr;
    DevSource3 src{};          // rgb owned by the context
    size_t n_pixels = 0;
    float *field = nullptr;
    Stats3Dev *stats = nullptr;
    uint32_t *cursor = nullptr;
    int wait_weight = 32, trav_burst = 3;   // a sweep over both constants: a leaf visit (four exact triangle distances) is dear, steps are served early
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

#define W3_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return set_error(WOST_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

static int upload_mesh3(const wost3_mesh_desc &d, DeviceMesh3 &s)
{
    if (d.n_tris < 0 || d.n_verts < 0) return set_error(WOST_ERR_INVALID, "negative mesh size");
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

static void destroy3(wost3_context *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (void *p : c->dm.allocs) (void)hipFree(p);
    for (void *p : c->nm.allocs) (void)hipFree(p);
    if (c->mask) (void)hipFree(c->mask);
    if (c->src.rgb) (void)hipFree(const_cast<float *>(c->src.rgb));
    if (c->field) (void)hipFree(c->field);
    if (c->stats) (void)hipFree(c->stats);
    if (c->cursor) (void)hipFree(c->cursor);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int run_solve3(wost3_context *c, int32_t pixel_begin, int32_t pixel_end, int32_t shard_index, int32_t shard_count, float *field_dev,
                      int32_t field_base, hipStream_t stream, wost_stats *stats)
{
    const auto t0 = std::chrono::high_resolution_clock::now();
    W3_TRY(hipSetDevice(c->device));
    W3_TRY(hipMemsetAsync(c->stats, 0, kStat3Copies * sizeof(Stats3Dev), stream));
    Walk3Params P{};
    P.dm = c->dm.view; P.nm = c->nm.view; P.st = c->dst; P.probe = c->probe; P.mask = c->mask; P.src = c->src;
    P.field = field_dev; P.field_base = field_base; P.pixel_begin = pixel_begin; P.pixel_end = pixel_end;
    P.shard_index = shard_index; P.shard_count = shard_count; P.stats = c->stats;
    P.cursor = c->cursor;
    P.tiled = (pixel_begin == 0 && pixel_end == (int32_t)c->n_pixels && ((c->settings.width | c->settings.height) & 7) == 0) ? 1 : 0;
    P.wait_weight = c->wait_weight; P.trav_burst = c->trav_burst;
    // a step that answers its Neumann queries on the tree is long and divergent: it waits until eight ninths of the
    // busy walkers of the wave stand at it (tools/probes/bench3d_shell.py: 1.8x over the Dirichlet-only setting on a 1280-triangle shell)
    if (c->nm.view.n_tris > WOST3_FLAT_MAX) P.wait_weight = 1;
    if (const char *w = std::getenv("WOST3_WAIT_WEIGHT")) P.wait_weight = std::max(1, std::atoi(w));
    if (const char *w = std::getenv("WOST3_TRAV_BURST")) P.trav_burst = std::max(1, std::atoi(w));
    W3_TRY(hipMemsetAsync(c->cursor, 0, sizeof(uint32_t), stream));
    const int bs = kWalk3Threads, n = pixel_end - pixel_begin;
    const int lv = std::max(c->dm.view.n_tris > 0 ? c->dm.view.levels : 1, c->nm.view.n_tris > 0 ? c->nm.view.levels : 1);
    size_t lds = (size_t)(3 * lv + 1) * bs * sizeof(uint32_t);
    float ms = 0.0f;
    if (n > 0) {
        W3_TRY(hipEventRecord(c->ev0, stream));
        const bool emissive = c->nm.view.n_tris > 0 && c->nm.view.emissive;
        const bool ntree = c->nm.view.n_tris > WOST3_FLAT_MAX;
        // Neumann mesh on the tree: its silhouette and ray queries are answered by the wave as a whole, through task pools in LDS
        // behind the stack columns (developer knobs: WOST3_COOP=0 for the per-lane queries, WOST3_POOL_CAP, WOST3_RAY_TRIGGER)
        P.coop = ntree ? 1 : 0;
        P.pool_cap = 512;
        P.ray_slot_trigger = 32;
        P.cp_slot_trigger = 64;
        if (const char *w = std::getenv("WOST3_CP_TRIGGER")) P.cp_slot_trigger = std::min(64, std::max(1, std::atoi(w)));
        if (const char *w = std::getenv("WOST3_COOP")) P.coop = (ntree && std::atoi(w) != 0) ? 1 : 0;
        if (const char *w = std::getenv("WOST3_POOL_CAP")) P.pool_cap = std::min(4096, std::max(96, std::atoi(w)));     // (64 roots must fit)
        if (const char *w = std::getenv("WOST3_RAY_TRIGGER")) P.ray_slot_trigger = std::min(64, std::max(1, std::atoi(w)));
        if (c->nm.view.levels > 11) P.coop = 0;      // (node and slot indices of a task: 26 bits, 4^(levels + 1) slots)
        P.stack_words = (3 * lv + 1) * bs;
        // the closest-point queries by the wave as well (WOST3_WAVE=0: the lane machine)
        bool wave = c->dm.view.n_tris > 0 && c->dm.view.levels <= 11;
        if (const char *w = std::getenv("WOST3_WAVE")) wave = wave && std::atoi(w) != 0;
        const size_t lds_pools = (size_t)(bs / 64) * (2 * (size_t)P.pool_cap + kPool3OwnerWords) * sizeof(uint32_t);
        if (lds + lds_pools > 64 * 1024) {      // (trees of millions of triangles: the stack columns alone fill the block's LDS)
            wave = false;
            P.coop = 0;
        }
        if (wave || P.coop) lds += lds_pools;
        auto kfn = ntree ? (c->src.rgb ? (emissive ? walk3_kernel<true, true, true> : walk3_kernel<false, true, true>)
                                       : (emissive ? walk3_kernel<true, false, true> : walk3_kernel<false, false, true>))
                         : (c->src.rgb ? (emissive ? walk3_kernel<true, true, false> : walk3_kernel<false, true, false>)
                                       : (emissive ? walk3_kernel<true, false, false> : walk3_kernel<false, false, false>));
        if (wave)
            kfn = ntree ? (c->src.rgb ? (emissive ? walk3_kernel<true, true, true, true> : walk3_kernel<false, true, true, true>)
                                      : (emissive ? walk3_kernel<true, false, true, true> : walk3_kernel<false, false, true, true>))
                        : (c->src.rgb ? (emissive ? walk3_kernel<true, true, false, true> : walk3_kernel<false, true, false, true>)
                                      : (emissive ? walk3_kernel<true, false, false, true> : walk3_kernel<false, false, false, true>));
        // persistent blocks: as many as the chip holds (LDS stacks and registers allow about four per CU), or fewer for small frames
        int n_cus = 256;
        (void)hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, c->device);
        int per_cu = 4;
        if (const char *w = std::getenv("WOST3_BLOCKS_PER_CU")) per_cu = std::max(1, std::atoi(w));
        const unsigned grid = (unsigned)std::min((n + bs - 1) / bs, n_cus * per_cu);
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(bs), lds, stream, P);
        W3_TRY(hipGetLastError());
        W3_TRY(hipEventRecord(c->ev1, stream));
    }
    std::vector<Stats3Dev> copies(kStat3Copies);
    W3_TRY(hipMemcpyAsync(copies.data(), c->stats, kStat3Copies * sizeof(Stats3Dev), hipMemcpyDeviceToHost, stream));
    W3_TRY(hipStreamSynchronize(stream));
    if (n > 0) W3_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
#ifdef WOST3_PROFILE
    {
        unsigned long long prof[16], zero[16] = {0};
        W3_TRY(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_prof3), sizeof(prof)));
        W3_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_prof3), zero, sizeof(zero)));
        std::fprintf(stderr, "[wost3 profile] wave cycles: silhouette %llu source %llu neumann-sample %llu walk-ray %llu | step trips %llu (cycles %llu, lanes %llu) trav trips %llu (cycles %llu, lanes %llu) | kernel total %llu, %.1f ms\n",
                     prof[0], prof[1], prof[2], prof[3], prof[8], prof[4], prof[9], prof[10], prof[5], prof[11], prof[6], ms);
        std::fprintf(stderr, "[wost3 profile] silhouette queries %llu: inner visits %llu, leaf visits %llu\n", prof[14], prof[12], prof[13]);
    }
#endif
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (const Stats3Dev &k : copies) {
            stats->walk_steps += k.steps; stats->walks_started += k.started; stats->walks_absorbed += k.absorbed;
            stats->walks_truncated += k.truncated; stats->neumann_hits += k.nhits;
        }
        stats->kernel_ms = ms;
        stats->kernel_launches = n > 0 ? 1 : 0;
        stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    }
    return WOST_OK;
}

static DeviceMesh3 *pick3(wost3_handle h, int which)
{
    return which == WOST_MESH_DIRICHLET ? &h->dm : which == WOST_MESH_NEUMANN ? &h->nm : nullptr;
}

struct Scratch3 {
    std::vector<void *> ptrs;
    ~Scratch3()
    {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <class T>
    hipError_t alloc(T **p, size_t count)
    {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, count * sizeof(T) + 16);
        if (e == hipSuccess) ptrs.push_back(q);
        *p = reinterpret_cast<T *>(q);
        return e;
    }
};

extern "C" {

int wost3_create(const wost3_scene_desc *scene, const wost_settings *settings, int device, wost3_handle *out)
{
    if (!scene || !settings || !out) return set_error(WOST_ERR_INVALID, "null argument");
    *out = nullptr;
    if (settings->width <= 0 || settings->height <= 0 || settings->spp < 0 || settings->max_depth <= 0)
        return set_error(WOST_ERR_INVALID, "bad settings");
    if ((int64_t)settings->width * settings->height > (1 << 28)) return set_error(WOST_ERR_UNSUPPORTED, "frame too large");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return set_error(WOST_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n_dev) return set_error(WOST_ERR_INVALID, "device index out of range");
    W3_TRY(hipSetDevice(device));
    wost3_context *c = new (std::nothrow) wost3_context();
    if (!c) return set_error(WOST_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->settings = *settings;
    c->dst = DevSettings{settings->width, settings->height, settings->spp, settings->max_depth, settings->eps_shell,
                         scene->dirichlet_intensity, scene->neumann_intensity};
    c->probe.scale = scene->probe_scale;
    for (int k = 0; k < 3; ++k) { c->probe.pos[k] = scene->probe_pos[k]; c->probe.up[k] = scene->probe_up[k]; c->probe.right[k] = scene->probe_right[k]; }
    c->n_pixels = (size_t)settings->width * settings->height;
    int rc = upload_mesh3(scene->dirichlet, c->dm);
    if (rc == WOST_OK) rc = upload_mesh3(scene->neumann, c->nm);
    hipError_t e = hipSuccess;
    if (rc == WOST_OK && scene->mask) {
        e = hipMalloc((void **)&c->mask, c->n_pixels);
        if (e == hipSuccess) e = hipMemcpy(c->mask, scene->mask, c->n_pixels, hipMemcpyHostToDevice);
    }
    if (rc == WOST_OK && e == hipSuccess && scene->source.nx > 0) {
        const wost3_source_desc &sd = scene->source;
        if (sd.ny <= 0 || sd.nz <= 0 || !sd.rgb) rc = set_error(WOST_ERR_INVALID, "source grid: bad size or null samples");
        else {
            const size_t bytes = (size_t)sd.nx * sd.ny * sd.nz * 3 * sizeof(float);
            float *d = nullptr;
            e = hipMalloc((void **)&d, bytes);
            if (e == hipSuccess) e = hipMemcpy(d, sd.rgb, bytes, hipMemcpyHostToDevice);
            c->src = DevSource3{d, sd.nx, sd.ny, sd.nz, sd.index_scale[0], sd.index_scale[1], sd.index_scale[2], sd.index_offset[0],
                                sd.index_offset[1], sd.index_offset[2], sd.intensity};
        }
    }
    if (rc == WOST_OK && e == hipSuccess) e = hipMalloc((void **)&c->field, c->n_pixels * 3 * sizeof(float));
    if (rc == WOST_OK && e == hipSuccess) e = hipMalloc((void **)&c->stats, kStat3Copies * sizeof(Stats3Dev));
    if (rc == WOST_OK && e == hipSuccess) e = hipMalloc((void **)&c->cursor, sizeof(uint32_t));
    if (rc == WOST_OK && e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (rc == WOST_OK && e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (rc == WOST_OK && e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (rc == WOST_OK && e != hipSuccess) rc = set_error(WOST_ERR_DEVICE, std::string("wost3_create: ") + hipGetErrorString(e));
    if (rc != WOST_OK) {
        destroy3(c);
        return rc;
    }
    *out = c;
    return WOST_OK;
}

int wost3_destroy(wost3_handle h)
{
    destroy3(h);
    return WOST_OK;
}

int wost3_solve(wost3_handle h, int32_t pixel_begin, int32_t pixel_end, float *field_rgb, wost_stats *stats)
{
    if (!h || !field_rgb) return set_error(WOST_ERR_INVALID, "null argument");
    if (pixel_begin < 0 || pixel_end > (int64_t)h->n_pixels || pixel_begin > pixel_end)
        return set_error(WOST_ERR_INVALID, "pixel range outside the frame");
    const size_t n = (size_t)(pixel_end - pixel_begin);
    if (n == 0) {
        if (stats) std::memset(stats, 0, sizeof(*stats));
        return WOST_OK;
    }
    W3_TRY(hipSetDevice(h->device));
    W3_TRY(hipMemsetAsync(h->field, 0, n * 3 * sizeof(float), h->stream));
    const int rc = run_solve3(h, pixel_begin, pixel_end, 0, 1, h->field, pixel_begin, h->stream, stats);
    if (rc != WOST_OK) return rc;
    W3_TRY(hipMemcpyAsync(field_rgb, h->field, n * 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    W3_TRY(hipStreamSynchronize(h->stream));
    return WOST_OK;
}

int wost3_solve_sharded(wost3_handle h, int32_t shard_index, int32_t shard_count, float *field_rgb_dev, void *stream, wost_stats *stats)
{
    if (!h || !field_rgb_dev) return set_error(WOST_ERR_INVALID, "null argument");
    if (shard_count <= 0 || shard_index < 0 || shard_index >= shard_count) return set_error(WOST_ERR_INVALID, "bad shard");
    return run_solve3(h, 0, (int32_t)h->n_pixels, shard_index, shard_count, field_rgb_dev, 0, reinterpret_cast<hipStream_t>(stream), stats);
}

int wost3_closest_point(wost3_handle h, int which_mesh, const float *pts, int32_t n, int32_t *out_idx, float *out_dist, float *out_uv,
                        int32_t *out_side)
{
    if (!h || !pts || n < 0) return set_error(WOST_ERR_INVALID, "null argument");
    DeviceMesh3 *m = pick3(h, which_mesh);
    if (!m || m->view.n_tris == 0) return set_error(WOST_ERR_INVALID, "mesh is empty or unknown");
    if (n == 0) return WOST_OK;
    W3_TRY(hipSetDevice(h->device));
    Scratch3 s;
    float *d_pts, *d_dist, *d_uv;
    int32_t *d_idx, *d_side;
    W3_TRY(s.alloc(&d_pts, (size_t)n * 3)); W3_TRY(s.alloc(&d_dist, n)); W3_TRY(s.alloc(&d_uv, (size_t)n * 2));
    W3_TRY(s.alloc(&d_idx, n)); W3_TRY(s.alloc(&d_side, n));
    W3_TRY(hipMemcpyAsync(d_pts, pts, (size_t)n * 12, hipMemcpyHostToDevice, h->stream));
    const int bs = 256;
    const size_t lds = (size_t)(3 * m->view.levels + 1) * bs * sizeof(uint32_t);
    hipLaunchKernelGGL(closest_point3_kernel, dim3((n + bs - 1) / bs), dim3(bs), lds, h->stream, m->view, d_pts, n, d_idx, d_dist, d_uv, d_side);
    W3_TRY(hipGetLastError());
    if (out_idx) W3_TRY(hipMemcpyAsync(out_idx, d_idx, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    if (out_dist) W3_TRY(hipMemcpyAsync(out_dist, d_dist, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    if (out_uv) W3_TRY(hipMemcpyAsync(out_uv, d_uv, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    if (out_side) W3_TRY(hipMemcpyAsync(out_side, d_side, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    W3_TRY(hipStreamSynchronize(h->stream));
    return WOST_OK;
}

int wost3_closest_silhouette(wost3_handle h, int which_mesh, const float *pts, const float *rmax, int32_t n, float *out_dist)
{
    if (!h || !pts || !out_dist || n < 0) return set_error(WOST_ERR_INVALID, "null argument");
    DeviceMesh3 *m = pick3(h, which_mesh);
    if (!m) return set_error(WOST_ERR_INVALID, "unknown mesh selector");
    if (n == 0) return WOST_OK;
    W3_TRY(hipSetDevice(h->device));
    Scratch3 s;
    float *d_pts, *d_rmax = nullptr, *d_out;
    W3_TRY(s.alloc(&d_pts, (size_t)n * 3)); W3_TRY(s.alloc(&d_out, n));
    W3_TRY(hipMemcpyAsync(d_pts, pts, (size_t)n * 12, hipMemcpyHostToDevice, h->stream));
    if (rmax) {
        W3_TRY(s.alloc(&d_rmax, n));
        W3_TRY(hipMemcpyAsync(d_rmax, rmax, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    }
    {
        const size_t lds = (size_t)(3 * (m->view.n_tris > 0 ? m->view.levels : 1) + 1) * 256 * sizeof(uint32_t);
        if (m->view.n_tris > WOST3_FLAT_MAX) hipLaunchKernelGGL((silhouette3_kernel<true>), dim3((n + 255) / 256), dim3(256), lds, h->stream, m->view, d_pts, d_rmax, n, d_out);
        else hipLaunchKernelGGL((silhouette3_kernel<false>), dim3((n + 255) / 256), dim3(256), lds, h->stream, m->view, d_pts, d_rmax, n, d_out);
    }
    W3_TRY(hipGetLastError());
    W3_TRY(hipMemcpyAsync(out_dist, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    W3_TRY(hipStreamSynchronize(h->stream));
    return WOST_OK;
}

int wost3_render_sdf(wost3_handle h, int which_mesh, float *out_dist)
{
    if (!h || !out_dist) return set_error(WOST_ERR_INVALID, "null argument");
    DeviceMesh3 *m = pick3(h, which_mesh);
    if (!m) return set_error(WOST_ERR_INVALID, "unknown mesh selector");
    W3_TRY(hipSetDevice(h->device));
    const int n = (int)h->n_pixels;
    Scratch3 s;
    float *d_out;
    W3_TRY(s.alloc(&d_out, n));
    const size_t lds = (size_t)(3 * (m->view.n_tris > 0 ? m->view.levels : 1) + 1) * 256 * sizeof(uint32_t);
    hipLaunchKernelGGL(render3_sdf_kernel, dim3((n + 255) / 256), dim3(256), lds, h->stream, m->view, h->probe, h->settings.width, h->settings.height,
                       which_mesh == WOST_MESH_NEUMANN ? 1 : 0, d_out);
    W3_TRY(hipGetLastError());
    W3_TRY(hipMemcpyAsync(out_dist, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    W3_TRY(hipStreamSynchronize(h->stream));
    return WOST_OK;
}

int wost3_render_source(wost3_handle h, float *out_rgb)
{
    if (!h || !out_rgb) return set_error(WOST_ERR_INVALID, "null argument");
    W3_TRY(hipSetDevice(h->device));
    const int n = (int)h->n_pixels;
    Scratch3 s;
    float *d_out;
    W3_TRY(s.alloc(&d_out, (size_t)n * 3));
    hipLaunchKernelGGL(render3_source_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->src, h->probe, h->settings.width, h->settings.height, d_out);
    W3_TRY(hipGetLastError());
    W3_TRY(hipMemcpyAsync(out_rgb, d_out, (size_t)n * 12, hipMemcpyDeviceToHost, h->stream));
    W3_TRY(hipStreamSynchronize(h->stream));
    return WOST_OK;
}

int wost3_ray_intersect(wost3_handle h, int which_mesh, const float *origins, const float *dirs, const float *tmax, int32_t n,
                        int32_t *out_hit, float *out_t, int32_t *out_idx)
{
    if (!h || !origins || !dirs || !tmax || !out_hit || !out_t || !out_idx || n < 0) return set_error(WOST_ERR_INVALID, "null argument");
    DeviceMesh3 *m = pick3(h, which_mesh);
    if (!m) return set_error(WOST_ERR_INVALID, "unknown mesh selector");
    if (n == 0) return WOST_OK;
    W3_TRY(hipSetDevice(h->device));
    Scratch3 s;
    float *d_o, *d_d, *d_tm, *d_t;
    int32_t *d_hit, *d_idx;
    W3_TRY(s.alloc(&d_o, (size_t)n * 3)); W3_TRY(s.alloc(&d_d, (size_t)n * 3)); W3_TRY(s.alloc(&d_tm, n)); W3_TRY(s.alloc(&d_t, n));
    W3_TRY(s.alloc(&d_hit, n)); W3_TRY(s.alloc(&d_idx, n));
    W3_TRY(hipMemcpyAsync(d_o, origins, (size_t)n * 12, hipMemcpyHostToDevice, h->stream));
    W3_TRY(hipMemcpyAsync(d_d, dirs, (size_t)n * 12, hipMemcpyHostToDevice, h->stream));
    W3_TRY(hipMemcpyAsync(d_tm, tmax, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    {
        const size_t lds = (size_t)(3 * (m->view.n_tris > 0 ? m->view.levels : 1) + 1) * 256 * sizeof(uint32_t);
        if (m->view.n_tris > WOST3_FLAT_MAX) hipLaunchKernelGGL((ray3_kernel<true>), dim3((n + 255) / 256), dim3(256), lds, h->stream, m->view, d_o, d_d, d_tm, n, d_hit, d_t, d_idx);
        else hipLaunchKernelGGL((ray3_kernel<false>), dim3((n + 255) / 256), dim3(256), lds, h->stream, m->view, d_o, d_d, d_tm, n, d_hit, d_t, d_idx);
    }
    W3_TRY(hipGetLastError());
    W3_TRY(hipMemcpyAsync(out_hit, d_hit, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    W3_TRY(hipMemcpyAsync(out_t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    W3_TRY(hipMemcpyAsync(out_idx, d_idx, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    W3_TRY(hipStreamSynchronize(h->stream));
    return WOST_OK;
}

int wost3_vmf_eval(int device, const float *kappa, const float *cos_theta, int32_t n, float *pdf)
{
    if (!kappa || !cos_theta || !pdf || n < 0) return set_error(WOST_ERR_INVALID, "null argument");
    if (n == 0) return WOST_OK;
    W3_TRY(hipSetDevice(device));
    Scratch3 s;
    float *d_k, *d_c, *d_p;
    W3_TRY(s.alloc(&d_k, n)); W3_TRY(s.alloc(&d_c, n)); W3_TRY(s.alloc(&d_p, n));
    W3_TRY(hipMemcpy(d_k, kappa, (size_t)n * 4, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_c, cos_theta, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(vmf_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_k, d_c, n, d_p);
    W3_TRY(hipGetLastError());
    W3_TRY(hipMemcpy(pdf, d_p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return WOST_OK;
}

int wost3_vmf_sample(int device, const float *kappa, const float *mu, const uint64_t *seed, int32_t n, int32_t per_point, float *dirs)
{
    if (!kappa || !mu || !seed || !dirs || n < 0 || per_point < 1) return set_error(WOST_ERR_INVALID, "null argument");
    if (n == 0) return WOST_OK;
    W3_TRY(hipSetDevice(device));
    Scratch3 s;
    float *d_k, *d_m, *d_o;
    uint64_t *d_s;
    W3_TRY(s.alloc(&d_k, n)); W3_TRY(s.alloc(&d_m, (size_t)n * 3)); W3_TRY(s.alloc(&d_s, n)); W3_TRY(s.alloc(&d_o, (size_t)n * per_point * 3));
    W3_TRY(hipMemcpy(d_k, kappa, (size_t)n * 4, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_m, mu, (size_t)n * 12, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_s, seed, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(vmf_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_k, d_m, d_s, n, per_point, d_o);
    W3_TRY(hipGetLastError());
    W3_TRY(hipMemcpy(dirs, d_o, (size_t)n * per_point * 12, hipMemcpyDeviceToHost));
    return WOST_OK;
}

int wost3_vmm_pdf_sample(int device, const float *raw, const float *wi, const uint64_t *seed, int32_t n, float *pdf, float *sample_dir)
{
    if (!raw || !wi || (sample_dir && !seed) || n < 0) return set_error(WOST_ERR_INVALID, "null argument");
    if (n == 0) return WOST_OK;
    W3_TRY(hipSetDevice(device));
    Scratch3 s;
    float *d_r, *d_w, *d_p = nullptr, *d_d = nullptr;
    uint64_t *d_s = nullptr;
    W3_TRY(s.alloc(&d_r, (size_t)n * 40)); W3_TRY(s.alloc(&d_w, (size_t)n * 3));
    W3_TRY(hipMemcpy(d_r, raw, (size_t)n * 160, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_w, wi, (size_t)n * 12, hipMemcpyHostToDevice));
    if (pdf) W3_TRY(s.alloc(&d_p, n));
    if (sample_dir) {
        W3_TRY(s.alloc(&d_d, (size_t)n * 3)); W3_TRY(s.alloc(&d_s, n));
        W3_TRY(hipMemcpy(d_s, seed, (size_t)n * 8, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(vmm3_pdf_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_r, d_w, d_s, n, d_p, d_d);
    W3_TRY(hipGetLastError());
    if (pdf) W3_TRY(hipMemcpy(pdf, d_p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (sample_dir) W3_TRY(hipMemcpy(sample_dir, d_d, (size_t)n * 12, hipMemcpyDeviceToHost));
    return WOST_OK;
}

int wost3_vmm_loss_gradients(int device, const float *raw, const float *dir, const float *li, const float *dir_pdf, const uint8_t *on_neumann,
                             const float *normal, int32_t n, float loss_scale, float *dl_draw, float *likelihood)
{
    if (!raw || !dir || !li || !dir_pdf || !dl_draw || (on_neumann && !normal) || n < 0) return set_error(WOST_ERR_INVALID, "null argument");
    if (n == 0) return WOST_OK;
    W3_TRY(hipSetDevice(device));
    Scratch3 s;
    float *d_r, *d_d, *d_l, *d_p, *d_n = nullptr, *d_g, *d_k = nullptr;
    unsigned char *d_o = nullptr;
    W3_TRY(s.alloc(&d_r, (size_t)n * 41)); W3_TRY(s.alloc(&d_d, (size_t)n * 3)); W3_TRY(s.alloc(&d_l, n)); W3_TRY(s.alloc(&d_p, n));
    W3_TRY(s.alloc(&d_g, (size_t)n * 41));
    W3_TRY(hipMemcpy(d_r, raw, (size_t)n * 164, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_d, dir, (size_t)n * 12, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_l, li, (size_t)n * 4, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_p, dir_pdf, (size_t)n * 4, hipMemcpyHostToDevice));
    if (on_neumann) {
        W3_TRY(s.alloc(&d_o, n)); W3_TRY(s.alloc(&d_n, (size_t)n * 3));
        W3_TRY(hipMemcpy(d_o, on_neumann, (size_t)n, hipMemcpyHostToDevice));
        W3_TRY(hipMemcpy(d_n, normal, (size_t)n * 12, hipMemcpyHostToDevice));
    }
    if (likelihood) W3_TRY(s.alloc(&d_k, n));
    hipLaunchKernelGGL(vmm3_loss_gradients_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_r, d_d, d_l, d_p, d_o, d_n, n, loss_scale, d_g, d_k);
    W3_TRY(hipGetLastError());
    W3_TRY(hipMemcpy(dl_draw, d_g, (size_t)n * 164, hipMemcpyDeviceToHost));
    if (likelihood) W3_TRY(hipMemcpy(likelihood, d_k, (size_t)n * 4, hipMemcpyDeviceToHost));
    return WOST_OK;
}

}  // extern "C"


// =====================================================================================================================
// GuidedIntegrator<3> (SURVEY.md 8a rows a21-a27 with DIM == 3; reference integrator/guided/integrator.cu with common3d,
// guided/parameters.h:26-33: 3 network inputs, 8 x (lambda, kappa, mean vector) + selection logit = 41 outputs)
// =====================================================================================================================
// A depth-synchronous wavefront like the 2-D guided path before its fusion (wost_guided.hip): per sample and depth
//   g3_separate_kernel  closest triangle, epsilon-shell -> colour into the pixel and its training records; else closest
//                       silhouette edge, star radius (no 0.99 here, :238-239), Neumann sample; the out-of-shell walkers are
//                       compacted into a queue together with their normalised network inputs
//   network inference   on the queue (the three-input network, wost3_net_create; scalar kernels)
//   g3_sample_kernel    routing by the selection probability, direction from the vMF mixture or uniform with one-sample
//                       MIS (reflection about the Neumann normal), the walker's ray, throughput, training record
// and after every trained sample the ordered training set, the loss gradients (vmm3_loss_gradients_kernel) and the Adam
// steps.  One thread per pixel / queue entry; per-pixel arithmetic and draw order are those of the CPU restatement the tests
// compare with (tests/test_guided_3d.py): bit-exact, the source term (sampleSource, :277-364 with DIM == 3) included.  From the first
// depth that needs no network on, g3_tail_kernel takes every walker that is left to its end in one launch.
namespace wost {

constexpr int kRec3Fields = 15;      // sol rgb, pos xyz, dir xyz, pdf, thp, normal xyz, onNeumann
constexpr int kMaxTrainDepth3 = 4;   // parameters.h:7

struct alignas(256) GStats3Dev {
    unsigned long long steps, started, absorbed, truncated, nhits, guided, net_points;
};

struct G3Box {
    float min[3], max[3];     // scene.aabb: contains()
    float c[3], e[3];         // centre and extent of the box inflated by 0.5 % of its diagonal (train.h:149-155)
};

struct G3Params {
    DevMesh3 dm, nm;
    DevSettings st;
    DevProbe3 probe;
    DevSource3 src;
    const uint8_t *mask;
    G3Box box;
    int32_t n_pixels, shard_index, shard_count;
    // per pixel
    uint64_t *rng;
    float *sol;               // 3 per pixel
    uint32_t *cur_depth;
    float *rec;               // [slot][field][pixel]
    int32_t *state;           // 0 none, 1 evaluation point queued, 2 out of shell
    float *wx, *wn;           // 3 per pixel: position, Neumann normal
    float *wthp, *wrb;
    uint8_t *won;
    int32_t *whint, *hint0;
    // the queue of a depth
    uint32_t *q_pid, *q_count;
    float *net_in, *net_out;
    GStats3Dev *stats;
    int32_t training, train_offset, train_stride, max_train_depth;
    int32_t depth, guiding, first_sample, stack_stride;
    float uniform_fraction;
    // the tree queries of a wave's walkers through its task pools (closest_triangle_pool & co.): pool_cap tasks per pool and wave,
    // pool_offset words into the block's LDS (behind the stack columns); pool_cap = 0: one descent per thread
    int32_t pool_cap, pool_offset;
};

// the task pools of this wave (8-byte LDS atomics: from an 8-byte boundary, whatever static words precede the dynamic segment)
__device__ __forceinline__ WavePool3 g3_pools(const G3Params &P, uint32_t *lds)
{
    uint32_t *pw = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(lds + P.pool_offset) + 7u) & ~(uintptr_t)7u) +
                   (threadIdx.x >> 6) * (2 * P.pool_cap + kPool3OwnerWords);
    return WavePool3{pw + kPool3OwnerWords, pw + kPool3OwnerWords + P.pool_cap, pw, P.pool_cap};
}

__device__ __forceinline__ GStats3Dev *g3_stats(GStats3Dev *s) { return s + (blockIdx.x & (kStat3Copies - 1)); }

__device__ __forceinline__ void g3_count(bool pred, unsigned long long *counter)
{
    const unsigned long long bal = __ballot(pred);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(counter, (unsigned long long)__popcll(bal));
}

__device__ __forceinline__ bool g3_training_pixel(const G3Params &P, uint32_t pid)
{
    return P.training && ((pid - (uint32_t)P.train_offset) % (uint32_t)P.train_stride == 0u);
}

__device__ __forceinline__ float &rec3_at(const G3Params &P, int slot, int field, uint32_t pid)
{
    return P.rec[((size_t)slot * kRec3Fields + field) * (size_t)P.n_pixels + pid];
}

// recordSolution / recordSourceContribution (guided.h:48-68): add to every record this walk has created
__device__ __forceinline__ void g3_record_solution(const G3Params &P, uint32_t pid, const float (&c)[3])
{
    const uint32_t n = min(P.cur_depth[pid], (uint32_t)kMaxTrainDepth3);
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) rec3_at(P, i, k, pid) = rec3_at(P, i, k, pid) + c[k];
}

__device__ __forceinline__ bool g3_box_contains(const G3Box &b, V3 q)
{
    return b.min[0] <= q.x && q.x <= b.max[0] && b.min[1] <= q.y && q.y <= b.max[1] && b.min[2] <= q.z && q.z <= b.max[2];
}

__device__ __forceinline__ void g3_normalize(const G3Box &b, V3 q, float (&o)[3])
{
    o[0] = 0.5f + (q.x - b.c[0]) / b.e[0];
    o[1] = 0.5f + (q.y - b.c[1]) / b.e[1];
    o[2] = 0.5f + (q.z - b.c[2]) / b.e[2];
}

// start of a sample (prepareSolve :112-128 on the first one, reset + generateEvaluationPoints :131-150 on every one)
__global__ __launch_bounds__(256) void g3_begin_kernel(G3Params P)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = false;
    if (p < P.n_pixels) {
        if (P.first_sample) {
            Pcg rng;
            pcg_seed_pixel(rng, p, P.st.width);
            P.rng[p] = rng.state;
            P.sol[3 * (size_t)p] = 0.0f; P.sol[3 * (size_t)p + 1] = 0.0f; P.sol[3 * (size_t)p + 2] = 0.0f;
            P.hint0[p] = -1;
        }
        P.cur_depth[p] = 0;
        const int px = p % P.st.width, py = p / P.st.width;
        const int tile = (py >> 3) * ((P.st.width + 7) >> 3) + (px >> 3);
        active = (tile % P.shard_count) == P.shard_index && (P.mask == nullptr || P.mask[p] != 0);
        P.state[p] = active ? 1 : 0;
        if (active) {
            const V3 x = eval_point3(P.probe, px, py, P.st.width, P.st.height);
            P.wx[3 * (size_t)p] = x.x; P.wx[3 * (size_t)p + 1] = x.y; P.wx[3 * (size_t)p + 2] = x.z;
            P.wn[3 * (size_t)p] = 0.0f; P.wn[3 * (size_t)p + 1] = 0.0f; P.wn[3 * (size_t)p + 2] = 0.0f;
            P.wthp[p] = 1.0f;
            P.won[p] = 0;
            P.whint[p] = P.hint0[p];
        }
    }
    g3_count(active, &g3_stats(P.stats)->started);
}

// separateEvaluationPoint + handleBoundary + sampleNeumann (guided/integrator.cu:153-249, 252-274, 367-494 with DIM == 3)
// for the walker of pixel p at `depth` (live: it has an evaluation point queued); every lane of the wave takes part (the tree
// queries are answered by the wave).  Returns whether the walker stays (out of the shell, R_B stored) and its position.
template <bool EMISSIVE, bool NTREE, bool SOURCE>
__device__ __forceinline__ bool g3_separate_body(const G3Params &P, int depth, int p, bool live, const WavePool3 &W, const LdsColumn &stk, V3 &x_out)
{
    const bool pooled = P.pool_cap > 0;
    g3_count(live, &g3_stats(P.stats)->steps);
    bool keep = false, absorbed = false;
    V3 x = v3(0.0f, 0.0f, 0.0f);
    // ---- the closest Dirichlet triangle: by the wave for all its walkers (closest_triangle_pool), or one descent per thread ----
    Closest cp{WOST_INF, -1};
    const bool has_d = P.dm.n_tris > 0;
    if (live) {
        x = v3(P.wx[3 * (size_t)p], P.wx[3 * (size_t)p + 1], P.wx[3 * (size_t)p + 2]);
        if (has_d && pooled) {
            const int32_t hint = P.whint[p];      // closest_triangle's seed
            if (hint >= 0 && P.dm.triOrig[hint] != WOST_FAR_INDEX) {
                const float4 a = P.dm.tri[3 * (size_t)hint], b = P.dm.tri[3 * (size_t)hint + 1], c = P.dm.tri[3 * (size_t)hint + 2];
                cp = Closest{tri_d2(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), x), hint};
            }
        }
    }
    if (has_d) {
        if (pooled) cp = closest_triangle_pool(P.dm, x, cp, live, W, stk, 64);
        else if (live) cp = closest_triangle(P.dm, x, P.whint[p], stk);
    }
    V3 nn = v3(0.0f, 0.0f, 0.0f);
    float thp = 0.0f, R_D = WOST_INF;
    bool on_n = false, train_px = false;
    Pcg rng{0, 1};
    const float eps = P.st.eps;
    const uint32_t pid = (uint32_t)p;
    if (live) {
        nn = v3(P.wn[3 * (size_t)p], P.wn[3 * (size_t)p + 1], P.wn[3 * (size_t)p + 2]);
        thp = P.wthp[p];
        on_n = P.won[p] != 0;
        train_px = g3_training_pixel(P, pid);
        rng = Pcg{P.rng[p], 1};
        if (has_d) {
            P.whint[p] = cp.slot;
            if (depth == 0) P.hint0[p] = cp.slot;
            const float4 a = P.dm.tri[3 * (size_t)cp.slot], b = P.dm.tri[3 * (size_t)cp.slot + 1], c = P.dm.tri[3 * (size_t)cp.slot + 2];
            const V3 p0 = v3(a.x, a.y, a.z), e0 = v3(b.x, b.y, b.z) - p0, e1 = v3(c.x, c.y, c.z) - p0;
            const int side = tri_side(p0, cross3(e0, e1), x);
            float u, v;
            tri_uv(p0, e0, e1, x, u, v);
            R_D = sqrtf(cp.d2);
            if (R_D < eps && u > 0.0f && v > 0.0f && u + v < 1.0f) {
                float col[3];
                const int32_t *tv = P.dm.triVerts + 3 * (size_t)cp.slot;
                surface_color3(P.dm.colors, tv[0], tv[1], tv[2], side, u, v, col);
                float *s = P.sol + 3 * (size_t)p;
                for (int k = 0; k < 3; ++k) {
                    col[k] *= P.st.dirichlet_intensity;
                    col[k] *= thp;
                    s[k] = col[k] + s[k];
                }
                if (train_px) g3_record_solution(P, pid, col);
                absorbed = true;
            }
        }
    }
    // ---- the closest silhouette edge: the same choice ----
    const bool mid = live && !absorbed;
    float R_N = WOST_INF;
    if (P.nm.n_tris > 0) {
        if (NTREE && pooled) R_N = closest_silhouette3_wave(P.nm, x, R_D, mid, W, stk);
        else if (mid) R_N = closest_silhouette3<NTREE>(P.nm, x, R_D, stk);
    }
    if (live) {
        if (!absorbed) {

            const float R_B = fmaxf(WOST_R_B_FLOOR, fminf(R_D, R_N));     // no 0.99 in the guided integrator (:238-239)
            if (!isinf(R_B)) {
                keep = true;
                P.wrb[p] = R_B;
                if (SOURCE) {
                    // sampleSource (guided/integrator.cu:277-364, templated on DIM): the uniform 3-D step's restatement (step3_b),
                    // the contribution recorded like a Neumann one (recordSourceContribution)
                    V3 sdir;
                    float dir_pdf, salpha = 1.0f;
                    {
                        const float u1 = pcg_next_float(rng), u2 = pcg_next_float(rng);
                        float c, s;
                        sincos_2pi(u2, c, s);
                        if (on_n) {
                            const float z = u1, r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
                            sdir = frame_to_world(nn, r * c, r * s, z);
                            dir_pdf = 1.0f / WOST_2PI;
                            salpha = 0.5f;
                        } else {
                            const float z = 1 - 2 * u1, r = sqrtf(1 - z * z);
                            sdir = v3(r * c, r * s, z);
                            dir_pdf = 1.0f / WOST_4PI;
                        }
                    }
                    float dist = R_B;
                    if (P.nm.n_tris > 0) {
                        float t;
                        int hi;
                        if (ray_closest3<NTREE>(P.nm, v3(x.x + eps * sdir.x, x.y + eps * sdir.y, x.z + eps * sdir.z), sdir, dist, t, hi, stk)) dist = fminf(t, dist);
                    }
                    const float g1 = pcg_next_float(rng), g2 = pcg_next_float(rng);
                    float gc, gs;
                    sincos_2pi(g2, gc, gs);
                    float r = (1.0f + sqrtf(1.0f - cbrt01(g1 * g1)) * gc) * R_B / 2.0f;
                    r = fmaxf(1e-4f, r);
                    if (r > R_B) r = R_B / 2.0f;
                    if (r <= dist) {
                        float f[3], col[3];
                        source3_eval(P.src, v3(x.x + r * sdir.x, x.y + r * sdir.y, x.z + r * sdir.z), f);
                        const float norm = R_B * R_B / 6.0f;
                        const float c1 = (1.0f / WOST_4PI) / (r * r), c2 = dir_pdf / (r * r);
                        float *sl = P.sol + 3 * (size_t)p;
                        for (int k = 0; k < 3; ++k) {
                            col[k] = thp * f[k] * norm * c1 / c2 / salpha;
                            sl[k] = col[k] + sl[k];
                        }
                        if (train_px) g3_record_solution(P, pid, col);
                    }
                }
                if (P.nm.n_tris > 0) {      // sampleNeumann: three draws whether or not the boundary emits
                    const float u0 = pcg_next_float(rng), u1 = pcg_next_float(rng), u2 = pcg_next_float(rng);
                    if (EMISSIVE) {
                        float pdf;
                        const int oi = (NTREE && P.nm.obox_levels > 0) ? sample_in_sphere3_tree(P.nm, x, R_B, u0, pdf) : sample_in_sphere3_flat(P.nm, x, R_B, u0, pdf);
                        if (oi != -1 && pdf > 0) {
                            const DevTri S = P.nm.flat[oi];
                            const V3 s0 = ld3(S.p0), s1 = ld3(S.p1), s2 = ld3(S.p2);
                            const float su = sqrtf(u1), b1 = u2 * su, b0 = 1.0f - su, b2 = 1.0f - b0 - b1;
                            const V3 sp = v3((s0.x * b0 + s1.x * b1) + s2.x * b2, (s0.y * b0 + s1.y * b1) + s2.y * b2, (s0.z * b0 + s1.z * b1) + s2.z * b2);
                            const V3 rv = sp - x;
                            const float r = sqrtf(dot3(rv, rv));
                            if (r < R_B && r > 0) {
                                V3 o = x;
                                if (on_n) o = v3(x.x + eps * nn.x, x.y + eps * nn.y, x.z + eps * nn.z);
                                V3 rd = sp - o;
                                const float cd = sqrtf(dot3(rd, rd));
                                if (cd > 0) { rd.x /= cd; rd.y /= cd; rd.z /= cd; }
                                if (!ray_any3<NTREE>(P.nm, o, rd, cd - eps, stk)) {
                                    int side = tri_side(s0, ld3(S.nraw), x);
                                    float uu, vv;
                                    tri_uv(s0, s1 - s0, s2 - s0, sp, uu, vv);
                                    if (on_n) {
                                        const float dn = dot3(ld3(S.n), nn);
                                        side = (0.0f < dn) - (dn < 0.0f);
                                    }
                                    if (side != 0) {
                                        float col[3];
                                        const int32_t *tv = P.nm.flatVerts + 3 * (size_t)oi;
                                        surface_color3(P.nm.colors, tv[0], tv[1], tv[2], side, uu, vv, col);
                                        const float alpha = on_n ? 0.5f : 1.0f;
                                        const float G = (1.0f / r - 1.0f / R_B) / WOST_4PI;
                                        float *s = P.sol + 3 * (size_t)p;
                                        for (int k = 0; k < 3; ++k) {
                                            col[k] *= P.st.neumann_intensity;
                                            col[k] *= thp * G / alpha / pdf;
                                            col[k] = -col[k];
                                            s[k] = col[k] + s[k];
                                        }
                                        if (train_px) g3_record_solution(P, pid, col);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        P.rng[p] = rng.state;
        P.state[p] = keep ? 2 : 0;
    }
    g3_count(absorbed, &g3_stats(P.stats)->absorbed);
    x_out = x;
    return keep;
}

template <bool EMISSIVE, bool NTREE, bool SOURCE>
__global__ __launch_bounds__(256) void g3_separate_kernel(G3Params P)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk{lds_stack + threadIdx.x, (uint32_t)P.stack_stride};
    const WavePool3 W = g3_pools(P, lds_stack);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < P.n_pixels && P.state[p] == 1;
    V3 x;
    const bool keep = g3_separate_body<EMISSIVE, NTREE, SOURCE>(P, P.depth, p, live, W, stk, x);
    const uint32_t s = block_push(keep, P.q_count);
    if (keep) {
        P.q_pid[s] = (uint32_t)p;
        float in3[3];
        g3_normalize(P.box, x, in3);
        P.net_in[3 * (size_t)s] = in3[0]; P.net_in[3 * (size_t)s + 1] = in3[1]; P.net_in[3 * (size_t)s + 2] = in3[2];
    }
}


// handleOutShellPoint + handleGuidedSampling / handleUniformSampling, or oneStepWalk beyond the guided depths
// (guided/integrator.cu:497-526, 671-880, 883-965 with DIM == 3)
// for the walker of pixel `pid` (live: out of the shell, R_B stored) at `depth`; raw = its 41 network outputs when `guiding`
template <bool NTREE>
__device__ __forceinline__ void g3_sample_body(const G3Params &P, int depth, bool guiding, uint32_t pid, bool live, const float *raw, const WavePool3 &W,
                                               const LdsColumn &stk)
{
    const bool pooled = NTREE && P.pool_cap > 0;
    bool guided = false, hit = false, moved = false;
    // (the walker's ray is answered by the wave for all its walkers, ray_closest3_wave: the step is cut in two around it)
    size_t p = 0;
    V3 x = v3(0.0f, 0.0f, 0.0f), nn = x, dir = x, cur = x;
    float thp = 0.0f, R_B = 0.0f, pdf = 0.0f, alpha = 1.0f;
    bool on_n = false, record = false, dropped = false;
    const float eps = P.st.eps;
    Pcg rng{0, 1};
    if (live) {
        p = pid;
        x = v3(P.wx[3 * p], P.wx[3 * p + 1], P.wx[3 * p + 2]);
        nn = v3(P.wn[3 * p], P.wn[3 * p + 1], P.wn[3 * p + 2]);
        thp = P.wthp[p]; R_B = P.wrb[p];
        on_n = P.won[p] != 0;
        record = g3_training_pixel(P, pid) && depth < P.max_train_depth;
        rng = Pcg{P.rng[p], 1};
        auto uniform_dir = [&]() {
            const float u1 = pcg_next_float(rng), u2 = pcg_next_float(rng);
            float c, s;
            sincos_2pi(u2, c, s);
            if (on_n) {
                const float z = u1, r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
                dir = frame_to_world(nn, r * c, r * s, z);
                pdf = 1.0f / WOST_2PI;
                alpha = 0.5f;
            } else {
                const float z = 1 - 2 * u1, r = sqrtf(1 - z * z);
                dir = v3(r * c, r * s, z);
                pdf = 1.0f / WOST_4PI;
                alpha = 1.0f;
            }
        };
        if (!guiding) {
            uniform_dir();
        } else {
            const float sel = 1 / (1.f + det_expf(-raw[40]));
            const bool inside = g3_box_contains(P.box, x);
            bool to_guided = (P.uniform_fraction == 0) || (pcg_next_float(rng) < sel);
            to_guided = to_guided && inside;
            if (to_guided) {
                if (!(P.uniform_fraction < 1.0f)) {
                    dropped = true;                      // the guided kernel is never launched (:1031): the walk ends here
                } else {
                    Vmm3 m;
                    vmm3_build(m, raw);
                    V3 w = vmm3_sample(m, rng);
                    float guided_pdf = vmm3_pdf(m, w);
                    float uniform_pdf = 1.0f / WOST_4PI;
                    alpha = 1.0f;
                    if (on_n) {
                        uniform_pdf = 1.0f / WOST_2PI;
                        alpha = 0.5f;
                        const float dn = (w.x * nn.x + w.y * nn.y) + w.z * nn.z;
                        const V3 r = v3(w.x - 2 * dn * nn.x, w.y - 2 * dn * nn.y, w.z - 2 * dn * nn.z);
                        if ((nn.x * w.x + nn.y * w.y) + nn.z * w.z <= 0) w = r;
                        guided_pdf += vmm3_pdf(m, r);
                    }
                    dir = w;
                    pdf = sel * guided_pdf + (1.0f - sel) * uniform_pdf;
                    guided = true;
                }
            } else {
                uniform_dir();
                if (inside) {
                    Vmm3 m;
                    vmm3_build(m, raw);
                    float guided_pdf = vmm3_pdf(m, dir);
                    if (on_n) {
                        const float dn = (dir.x * nn.x + dir.y * nn.y) + dir.z * nn.z;
                        guided_pdf += vmm3_pdf(m, v3(dir.x - 2 * dn * nn.x, dir.y - 2 * dn * nn.y, dir.z - 2 * dn * nn.z));
                    }
                    pdf = sel * guided_pdf + (1.0f - sel) * pdf;
                }
            }
        }
        cur = x;
        if (on_n) cur = v3(x.x + eps * nn.x, x.y + eps * nn.y, x.z + eps * nn.z);
    }
    const bool go = live && !dropped;
    float t = 0.0f;
    int hi = -1;
    if (P.nm.n_tris > 0) {
        if (pooled) hit = ray_closest3_wave(P.nm, cur, dir, R_B, go, t, hi, W, stk, 32);
        else if (go) hit = ray_closest3<NTREE>(P.nm, cur, dir, R_B, t, hi, stk);
    }
    if (live) {
        if (dropped) {
            P.state[p] = 0;
        } else {
            V3 nxt = v3(x.x + R_B * dir.x, x.y + R_B * dir.y, x.z + R_B * dir.z);
            V3 hn = v3(0.0f, 0.0f, 0.0f);
            if (P.nm.n_tris > 0) {
                if (hit) {
                    hn = ld3(P.nm.flat[hi].n);
                    if (dot3(hn, dir) > 0) hn = v3(-hn.x, -hn.y, -hn.z);
                    nxt = v3(cur.x + t * dir.x, cur.y + t * dir.y, cur.z + t * dir.z);
                }
            }
            if (record) {       // incrementDepth (guided.h:21-46): the vertex BEFORE the step
                const uint32_t d = P.cur_depth[p];
                if (d < (uint32_t)kMaxTrainDepth3) {
                    rec3_at(P, d, 0, pid) = 0.0f; rec3_at(P, d, 1, pid) = 0.0f; rec3_at(P, d, 2, pid) = 0.0f;
                    rec3_at(P, d, 3, pid) = x.x; rec3_at(P, d, 4, pid) = x.y; rec3_at(P, d, 5, pid) = x.z;
                    rec3_at(P, d, 6, pid) = dir.x; rec3_at(P, d, 7, pid) = dir.y; rec3_at(P, d, 8, pid) = dir.z;
                    rec3_at(P, d, 9, pid) = pdf;
                    rec3_at(P, d, 10, pid) = thp;
                    rec3_at(P, d, 11, pid) = nn.x; rec3_at(P, d, 12, pid) = nn.y; rec3_at(P, d, 13, pid) = nn.z;
                    rec3_at(P, d, 14, pid) = on_n ? 1.0f : 0.0f;
                    P.cur_depth[p] = d + 1;
                }
            }
            P.wthp[p] = thp / pdf / alpha / WOST_4PI;
            P.wx[3 * p] = nxt.x; P.wx[3 * p + 1] = nxt.y; P.wx[3 * p + 2] = nxt.z;
            P.wn[3 * p] = hn.x; P.wn[3 * p + 1] = hn.y; P.wn[3 * p + 2] = hn.z;
            P.won[p] = hit ? 1 : 0;
            P.state[p] = 1;
            moved = true;
        }
        P.rng[p] = rng.state;
    }
    GStats3Dev *st = g3_stats(P.stats);
    g3_count(guided, &st->guided);
    g3_count(hit, &st->nhits);
    g3_count(moved && depth == P.st.max_depth - 1, &st->truncated);
    g3_count(live && guiding, &st->net_points);
}

template <bool NTREE>
__global__ __launch_bounds__(256) void g3_sample_kernel(G3Params P)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk{lds_stack + threadIdx.x, (uint32_t)P.stack_stride};
    const WavePool3 W = g3_pools(P, lds_stack);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < *P.q_count;
    g3_sample_body<NTREE>(P, P.depth, P.guiding != 0, live ? P.q_pid[i] : 0u, live, P.net_out + 41 * (size_t)i, W, stk);
}

// The unguided tail of a sample: from depth >= maxGuidedDepth on nothing needs the network, yet a launch pair per depth over a
// frame that holds a handful of walkers cost what its slowest tree query costs (most of the ~2000 launches of a 16-sample
// solve).  Here every walker that is left runs to its end in ONE launch -- the same bodies, depth after depth, the tree queries
// still answered by the wave -- with its state where the bodies keep it (a thread reads back its own stores).
template <bool EMISSIVE, bool NTREE, bool SOURCE>
__global__ __launch_bounds__(256) void g3_tail_kernel(G3Params P)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk{lds_stack + threadIdx.x, (uint32_t)P.stack_stride};
    const WavePool3 W = g3_pools(P, lds_stack);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = p < P.n_pixels;
    for (int depth = P.depth; depth < P.st.max_depth; ++depth) {
        const bool live = mine && P.state[p] == 1;
        if (!__ballot(live)) break;       // (wave-uniform: the queries are the wave's)
        V3 x;
        const bool keep = g3_separate_body<EMISSIVE, NTREE, SOURCE>(P, depth, p, live, W, stk, x);
        g3_sample_body<NTREE>(P, depth, false, (uint32_t)(mine ? p : 0), keep, nullptr, W, stk);
    }
}


// ---- the training set of a pass, in (pixel, record) order (train.h:423-471) ------------------------------------------------
struct T3Params {
    G3Params G;
    uint32_t *block_sums;      // records per block of 256 training pixels; after the scan: first output index of the block
    int32_t n_train_pixels;
    float *t_x, *t_dir, *t_sol, *t_li, *t_pdf, *t_nrm;
    uint8_t *t_onn;
};

template <bool SCATTER>
__global__ __launch_bounds__(256) void g3_train_set_kernel(T3Params T)
{
    __shared__ uint32_t s_scan[256];
    const G3Params &P = T.G;
    const int t = blockIdx.x * 256 + threadIdx.x;
    uint32_t n_valid = 0;
    uint32_t valid_mask = 0;
    uint32_t pid = 0;
    if (t < T.n_train_pixels) {
        pid = (uint32_t)P.train_offset + (uint32_t)t * (uint32_t)P.train_stride;
        const uint32_t depth = P.cur_depth[pid];
        for (uint32_t k = 0; k < depth; ++k) {
            const V3 rp = v3(rec3_at(P, k, 3, pid), rec3_at(P, k, 4, pid), rec3_at(P, k, 5, pid));
            if (!g3_box_contains(P.box, rp)) continue;
            const float thp = rec3_at(P, k, 10, pid), pdf = rec3_at(P, k, 9, pid);
            float s3[3];
            bool bad = false;
            for (int ch = 0; ch < 3; ++ch) {
                float v = 0.0f;
                if (fabsf(thp) > 1e-5f) v = rec3_at(P, k, ch, pid) / thp;
                s3[ch] = fabsf(v);
                bad = bad || isnan(s3[ch]);
            }
            float in3[3];
            g3_normalize(P.box, rp, in3);
            bad = bad || isnan(in3[0]) || isnan(in3[1]) || isnan(in3[2]) || isnan(rec3_at(P, k, 6, pid)) || isnan(rec3_at(P, k, 7, pid)) ||
                  isnan(rec3_at(P, k, 8, pid)) || isnan(pdf) || pdf == 0;
            if (bad) continue;
            valid_mask |= 1u << k;
            ++n_valid;
        }
    }
    // exclusive prefix of n_valid over the block
    s_scan[threadIdx.x] = n_valid;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)off ? s_scan[threadIdx.x - off] : 0u;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    if (!SCATTER) {
        if (threadIdx.x == 255) T.block_sums[blockIdx.x] = s_scan[255];
        return;
    }
    size_t o = (size_t)T.block_sums[blockIdx.x] + (s_scan[threadIdx.x] - n_valid);
    for (uint32_t k = 0; k < (uint32_t)kMaxTrainDepth3; ++k) {
        if (!(valid_mask & (1u << k))) continue;
        const V3 rp = v3(rec3_at(P, k, 3, pid), rec3_at(P, k, 4, pid), rec3_at(P, k, 5, pid));
        const float thp = rec3_at(P, k, 10, pid);
        float s3[3], in3[3];
        for (int ch = 0; ch < 3; ++ch) {
            float v = 0.0f;
            if (fabsf(thp) > 1e-5f) v = rec3_at(P, k, ch, pid) / thp;
            s3[ch] = fabsf(v);
        }
        g3_normalize(P.box, rp, in3);
        for (int c = 0; c < 3; ++c) {
            T.t_x[3 * o + c] = in3[c];
            T.t_dir[3 * o + c] = rec3_at(P, k, 6 + c, pid);
            T.t_sol[3 * o + c] = s3[c];
            T.t_nrm[3 * o + c] = rec3_at(P, k, 11 + c, pid);
        }
        T.t_li[o] = (s3[0] + s3[1] + s3[2]) / 3.0f;
        T.t_pdf[o] = rec3_at(P, k, 9, pid);
        T.t_onn[o] = rec3_at(P, k, 14, pid) != 0.0f ? 1 : 0;
        ++o;
    }
}

// exclusive scan of the block sums in place (one block; the total goes to sums[n])
__global__ __launch_bounds__(1024) void g3_scan_kernel(uint32_t *sums, int n)
{
    __shared__ uint32_t s[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < n ? sums[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t a = threadIdx.x >= (unsigned)off ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < n) sums[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[n] = carry;
}

__global__ void g3_resolve_kernel(const float *sol, const int32_t *owned_state, int n, float spp, float *field)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * n) field[i] = sol[i] / spp;
}

}  // namespace wost

struct wost3_guided {
    int device = 0;
    wost3_handle scene = nullptr;
    wost3_guided_settings s{};
    wost_net_handle net = nullptr;
    G3Box box{};
    std::vector<void *> allocs;
    uint64_t *rng = nullptr;
    float *sol = nullptr, *rec = nullptr, *wx = nullptr, *wn = nullptr, *wthp = nullptr, *wrb = nullptr, *net_in = nullptr, *net_out = nullptr, *field = nullptr;
    uint32_t *cur_depth = nullptr, *q_pid = nullptr, *q_count = nullptr, *block_sums = nullptr;
    int32_t *state = nullptr, *whint = nullptr, *hint0 = nullptr;
    uint8_t *won = nullptr, *t_onn = nullptr;
    float *t_x = nullptr, *t_dir = nullptr, *t_sol = nullptr, *t_li = nullptr, *t_pdf = nullptr, *t_nrm = nullptr;
    GStats3Dev *stats = nullptr;
    uint32_t *host_word = nullptr;      // pinned
    uint32_t last_train_n = 0;
    uint64_t host_rng = 0;
};

template <class T>
static hipError_t g3_alloc(wost3_guided *g, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) g->allocs.push_back(q);
    *p = reinterpret_cast<T *>(q);
    return e;
}

static void g3_free(wost3_guided *g)
{
    if (!g) return;
    (void)hipSetDevice(g->device);
    for (void *p : g->allocs) (void)hipFree(p);
    if (g->host_word) (void)hipHostFree(g->host_word);
    if (g->net) (void)wost_net_destroy(g->net);
    if (g->scene) (void)wost3_destroy(g->scene);
    delete g;
}

static int run_guided3(wost3_guided *g, int shard_index, int shard_count, float *field_host, float *field_dev, wost_guided_stats *stats)
{
    const auto t_start = std::chrono::high_resolution_clock::now();
    W3_TRY(hipSetDevice(g->device));
    wost3_context *c = g->scene;
    const wost3_guided_settings &s = g->s;
    const int N = s.width * s.height;
    hipStream_t stream = c->stream;
    const int d_levels = c->dm.view.n_tris > 0 ? c->dm.view.levels : 1, n_levels = c->nm.view.n_tris > 0 ? c->nm.view.levels : 1;
    const int stack_words = 3 * std::max(d_levels, n_levels) + 4;
    size_t lds = (size_t)stack_words * 256 * sizeof(uint32_t);
    const bool ntree = c->nm.view.n_tris > WOST_FLAT_MAX, emissive = c->nm.view.n_tris > 0 && c->nm.view.emissive;
    G3Params P{};
    // the tree queries of a wave's walkers through its task pools, as in walk3_kernel (WOST3_WAVE=0: one descent per thread)
    P.pool_cap = (d_levels <= 11 && n_levels <= 11) ? 512 : 0;
    if (const char *w = std::getenv("WOST3_WAVE")) P.pool_cap = std::atoi(w) != 0 ? P.pool_cap : 0;
    if (const char *w = std::getenv("WOST3_POOL_CAP")) P.pool_cap = P.pool_cap ? std::min(4096, std::max(96, std::atoi(w))) : 0;
    P.pool_offset = stack_words * 256;
    if (P.pool_cap && lds + (size_t)4 * (2 * (size_t)P.pool_cap + kPool3OwnerWords) * sizeof(uint32_t) + 8 > 64 * 1024) P.pool_cap = 0;
    if (P.pool_cap) lds += (size_t)4 * (2 * (size_t)P.pool_cap + kPool3OwnerWords) * sizeof(uint32_t) + 8;
    P.dm = c->dm.view; P.nm = c->nm.view; P.st = c->dst; P.probe = c->probe; P.mask = c->mask; P.box = g->box; P.src = c->src;
    const bool has_src = c->src.rgb != nullptr;
    P.n_pixels = N; P.shard_index = shard_index; P.shard_count = shard_count;
    P.rng = g->rng; P.sol = g->sol; P.cur_depth = g->cur_depth; P.rec = g->rec; P.state = g->state; P.wx = g->wx; P.wn = g->wn;
    P.wthp = g->wthp; P.wrb = g->wrb; P.won = g->won; P.whint = g->whint; P.hint0 = g->hint0;
    P.q_pid = g->q_pid; P.q_count = g->q_count; P.net_in = g->net_in; P.net_out = g->net_out; P.stats = g->stats;
    P.max_train_depth = s.max_train_depth; P.stack_stride = 256;
    uint32_t train_offset = 0;
    if (s.train_pixel_stride > 1) {
        if (s.train_pixel_offset >= 0) train_offset = (uint32_t)s.train_pixel_offset;
        else {
            // prepareSolve (integrator.cu:126): one draw of the integrator's host sampler per solve (pcg32, seed of the handle)
            const uint64_t old = g->host_rng;
            g->host_rng = old * 0x5851f42d4c957f2dULL + 1u;
            const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
            union { uint32_t u; float f; } x;
            x.u = (((xs >> rot) | (xs << ((~rot + 1u) & 31))) >> 9) | 0x3f800000u;
            train_offset = (uint32_t)((x.f - 1.0f) * (float)s.train_pixel_stride);
        }
    }
    P.train_offset = (int32_t)train_offset; P.train_stride = s.train_pixel_stride;
    const int n_train_pixels = (int)(((size_t)N - train_offset + (size_t)s.train_pixel_stride - 1) / (size_t)s.train_pixel_stride);
    const int n_train_blocks = (n_train_pixels + 255) / 256;
    W3_TRY(hipMemsetAsync(g->stats, 0, kStat3Copies * sizeof(GStats3Dev), stream));
    const unsigned grid_px = (unsigned)((N + 255) / 256);
    uint32_t launches = 0;
    uint64_t train_samples = 0;
    double train_ms = 0.0;
    const int opt_before = net_optimizer_steps(g->net);
    const uint64_t net_launches_before = net_launch_count(g->net);
    bool training = true;
    float uniform_fraction = s.uniform_fraction_training;
    int max_guided_depth = s.max_guided_depth_training;
    for (int sample = 0; sample < s.spp; ++sample) {
        if (sample == s.train_spp_count) {      // :991-996
            training = false;
            uniform_fraction = s.uniform_fraction_guiding;
            max_guided_depth = s.max_guided_depth_guiding;
        }
        P.training = training ? 1 : 0; P.uniform_fraction = uniform_fraction; P.first_sample = sample == 0 ? 1 : 0;
        hipLaunchKernelGGL(g3_begin_kernel, dim3(grid_px), dim3(256), 0, stream, P);
        ++launches;
        // No host round trip inside a sample: the launches of a depth are sized for the frame (their kernels read the true length of
        // the queue on the device; a block beyond it ends at once), and from the first depth that needs no network on, ONE launch
        // takes every walker that is left to its end (g3_tail_kernel).  A round trip per depth -- later one every fourth depth --
        // and the launch pairs of the late depths, whose few walkers cost a launch what its slowest tree query costs, were most
        // of the solve's wall time (about 2000 launches per 16-sample solve).
        const uint32_t n_upper = (uint32_t)N;
        for (int depth = 0; depth < s.max_depth; ++depth) {
            P.depth = depth; P.guiding = depth < max_guided_depth ? 1 : 0;
            if (!P.guiding) {
#define G3_LAUNCH(K, E, T)                                                                                              \
    do {                                                                                                                \
        if (has_src) hipLaunchKernelGGL((K<E, T, true>), dim3(grid_px), dim3(256), lds, stream, P);                       \
        else hipLaunchKernelGGL((K<E, T, false>), dim3(grid_px), dim3(256), lds, stream, P);                              \
    } while (0)
                if (ntree) { if (emissive) G3_LAUNCH(g3_tail_kernel, true, true); else G3_LAUNCH(g3_tail_kernel, false, true); }
                else       { if (emissive) G3_LAUNCH(g3_tail_kernel, true, false); else G3_LAUNCH(g3_tail_kernel, false, false); }
                ++launches;
                break;
            }
            W3_TRY(hipMemsetAsync(g->q_count, 0, sizeof(uint32_t), stream));
            if (ntree) { if (emissive) G3_LAUNCH(g3_separate_kernel, true, true); else G3_LAUNCH(g3_separate_kernel, false, true); }
            else       { if (emissive) G3_LAUNCH(g3_separate_kernel, true, false); else G3_LAUNCH(g3_separate_kernel, false, false); }
            ++launches;
            {
                const int rc = net_inference_dev(g->net, g->net_in, g->q_count, (int)n_upper, g->net_out, true, stream, 0);
                if (rc != WOST_OK) return rc;
            }
            const unsigned grid_q = (n_upper + 255u) / 256u;
            if (ntree) hipLaunchKernelGGL((g3_sample_kernel<true>), dim3(grid_q), dim3(256), lds, stream, P);
            else hipLaunchKernelGGL((g3_sample_kernel<false>), dim3(grid_q), dim3(256), lds, stream, P);
            ++launches;
        }
        W3_TRY(hipGetLastError());
        if (training) {
            const auto t0 = std::chrono::high_resolution_clock::now();
            T3Params T{};
            T.G = P; T.block_sums = g->block_sums; T.n_train_pixels = n_train_pixels;
            T.t_x = g->t_x; T.t_dir = g->t_dir; T.t_sol = g->t_sol; T.t_li = g->t_li; T.t_pdf = g->t_pdf; T.t_nrm = g->t_nrm; T.t_onn = g->t_onn;
            hipLaunchKernelGGL((g3_train_set_kernel<false>), dim3(n_train_blocks), dim3(256), 0, stream, T);
            hipLaunchKernelGGL(g3_scan_kernel, dim3(1), dim3(1024), 0, stream, g->block_sums, n_train_blocks);
            hipLaunchKernelGGL((g3_train_set_kernel<true>), dim3(n_train_blocks), dim3(256), 0, stream, T);
            launches += 3;
            W3_TRY(hipMemcpyAsync(g->host_word, g->block_sums + n_train_blocks, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            W3_TRY(hipStreamSynchronize(stream));
            const size_t n = g->host_word[0];
            g->last_train_n = (uint32_t)n;
            train_samples += n;
            const size_t bs = (size_t)s.batch_size;
            size_t n_batches = std::min(n / bs + 1, (size_t)s.batches_per_spp);
            for (size_t it = 0; it < n_batches; ++it) {
                if (it * bs > n) break;
                size_t local = std::min(n - it * bs, bs);
                local -= local % 128;
                if (local < (size_t)s.min_batch_size) break;
                const size_t o = it * bs;
                float *out = nullptr, *dl = nullptr;
                int rc = net_forward_train_dev(g->net, g->t_x + 3 * o, (int)local, stream, &out, &dl);
                if (rc != WOST_OK) return rc;
                hipLaunchKernelGGL(vmm3_loss_gradients_kernel, dim3((unsigned)((local + 255) / 256)), dim3(256), 0, stream, out, g->t_dir + 3 * o,
                                   g->t_li + o, g->t_pdf + o, g->t_onn + o, g->t_nrm + 3 * o, (int)local, s.loss_scale, dl, (float *)nullptr);
                ++launches;
                rc = net_backward_update_dev(g->net, g->t_x + 3 * o, (int)local, s.loss_scale, 1, stream);
                if (rc != WOST_OK) return rc;
            }
            W3_TRY(hipStreamSynchronize(stream));
            train_ms += std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
        }
    }
    hipLaunchKernelGGL(g3_resolve_kernel, dim3((unsigned)((3 * N + 255) / 256)), dim3(256), 0, stream, g->sol, g->state, N, (float)s.spp, g->field);
    W3_TRY(hipGetLastError());
    if (field_host) W3_TRY(hipMemcpyAsync(field_host, g->field, (size_t)N * 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
    if (field_dev) W3_TRY(hipMemcpyAsync(field_dev, g->field, (size_t)N * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream));
    std::vector<GStats3Dev> copies(kStat3Copies);
    W3_TRY(hipMemcpyAsync(copies.data(), g->stats, kStat3Copies * sizeof(GStats3Dev), hipMemcpyDeviceToHost, stream));
    W3_TRY(hipStreamSynchronize(stream));
    if (stats) {
        *stats = wost_guided_stats{};
        for (const GStats3Dev &k : copies) {
            stats->walk_steps += k.steps; stats->walks_started += k.started; stats->walks_absorbed += k.absorbed;
            stats->walks_truncated += k.truncated; stats->neumann_hits += k.nhits; stats->guided_steps += k.guided; stats->net_points += k.net_points;
        }
        stats->train_samples = train_samples;
        stats->optimizer_steps = (uint64_t)(net_optimizer_steps(g->net) - opt_before);
        stats->train_ms = train_ms;
        stats->kernel_launches = launches + (uint32_t)(net_launch_count(g->net) - net_launches_before);
        stats->reserved = train_offset;
        stats->solve_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_start).count();
    }
    return WOST_OK;
}

extern "C" {

int wost3_guided_create(const wost3_scene_desc *scene, const wost3_guided_settings *s, const wost_net_config *net, uint64_t net_seed,
                        int device, wost3_guided_handle *out)
{
    if (!scene || !s || !net || !out) return set_error(WOST_ERR_INVALID, "null argument");
    *out = nullptr;
    if (s->max_train_depth < 0 || s->max_train_depth > kMaxTrainDepth3 || s->train_pixel_stride < 1 || s->batch_size < 128 ||
        s->batches_per_spp < 0 || s->train_spp_count < 0)
        return set_error(WOST_ERR_INVALID, "bad guided settings");
    if (net->n_output != 41) return set_error(WOST_ERR_INVALID, "the 3-D guiding network has 41 outputs (8 x (lambda, kappa, mean vector) + selection logit)");
    wost_settings us{s->width, s->height, s->spp, s->max_depth, s->eps_shell};
    wost3_handle sc = nullptr;
    int rc = wost3_create(scene, &us, device, &sc);
    if (rc != WOST_OK) return rc;
    wost3_guided *g = new (std::nothrow) wost3_guided();
    if (!g) { (void)wost3_destroy(sc); return set_error(WOST_ERR_NOMEM, "out of host memory"); }
    g->device = device; g->scene = sc; g->s = *s;
    g->host_rng = 0x853c49e6748fea9bULL ^ net_seed;
    rc = wost3_net_create(device, net, net_seed, &g->net);
    if (rc != WOST_OK) { g3_free(g); return rc; }
    {
        // normalizeSpatialCoord (train.h:149-155): the box inflated by 0.5 % of its diagonal; Eigen norm() adds the squares in order
        const float e[3] = {s->aabb_max[0] - s->aabb_min[0], s->aabb_max[1] - s->aabb_min[1], s->aabb_max[2] - s->aabb_min[2]};
        const float infl = std::sqrt((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) * 0.005f;
        for (int a = 0; a < 3; ++a) {
            const float lo = s->aabb_min[a] - infl, hi = s->aabb_max[a] + infl;
            g->box.min[a] = s->aabb_min[a]; g->box.max[a] = s->aabb_max[a];
            g->box.c[a] = (lo + hi) / 2.0f; g->box.e[a] = hi - lo;
        }
    }
    const size_t N = (size_t)s->width * s->height, cap = N * kMaxTrainDepth3;
    hipError_t e = hipSuccess;
#define G3A(p, n) if (e == hipSuccess) e = g3_alloc(g, &g->p, (n))
    G3A(rng, N); G3A(sol, 3 * N); G3A(cur_depth, N); G3A(rec, (size_t)kMaxTrainDepth3 * kRec3Fields * N); G3A(state, N); G3A(wx, 3 * N); G3A(wn, 3 * N);
    G3A(wthp, N); G3A(wrb, N); G3A(won, N); G3A(whint, N); G3A(hint0, N); G3A(q_pid, N); G3A(q_count, 4); G3A(net_in, 3 * N); G3A(net_out, 41 * N);
    G3A(field, 3 * N); G3A(stats, kStat3Copies); G3A(block_sums, N / 256 + 4);
    G3A(t_x, 3 * cap); G3A(t_dir, 3 * cap); G3A(t_sol, 3 * cap); G3A(t_li, cap); G3A(t_pdf, cap); G3A(t_nrm, 3 * cap); G3A(t_onn, cap);
#undef G3A
    if (e == hipSuccess) e = hipHostMalloc((void **)&g->host_word, 4 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(g->cur_depth, 0, N * sizeof(uint32_t));
    if (e != hipSuccess) {
        g3_free(g);
        return set_error(WOST_ERR_DEVICE, std::string("guided 3-D allocation: ") + hipGetErrorString(e));
    }
    *out = g;
    return WOST_OK;
}

int wost3_guided_destroy(wost3_guided_handle h)
{
    g3_free(h);
    return WOST_OK;
}

int wost3_guided_network(wost3_guided_handle h, wost_net_handle *net)
{
    if (!h || !net) return set_error(WOST_ERR_INVALID, "null argument");
    *net = h->net;
    return WOST_OK;
}

int wost3_guided_scene(wost3_guided_handle h, wost3_handle *scene)
{
    if (!h || !scene) return set_error(WOST_ERR_INVALID, "null argument");
    *scene = h->scene;
    return WOST_OK;
}

int wost3_guided_solve(wost3_guided_handle h, float *field_rgb, wost_guided_stats *stats)
{
    if (!h || !field_rgb) return set_error(WOST_ERR_INVALID, "null argument");
    return run_guided3(h, 0, 1, field_rgb, nullptr, stats);
}

int wost3_guided_solve_sharded(wost3_guided_handle h, int32_t shard_index, int32_t shard_count, float *field_rgb_dev, wost_guided_stats *stats)
{
    if (!h || !field_rgb_dev) return set_error(WOST_ERR_INVALID, "null argument");
    if (shard_count <= 0 || shard_index < 0 || shard_index >= shard_count) return set_error(WOST_ERR_INVALID, "bad shard");
    return run_guided3(h, shard_index, shard_count, nullptr, field_rgb_dev, stats);
}

// queryNetwork(Vector3f) (exec.cu:175-186, guided/integrator.cu:566-615): the raw mixture parameters (41 per point) of the
// inference weights at world positions
int wost3_guided_query_network(wost3_guided_handle h, const float *pts, int32_t n, float *raw)
{
    if (!h || !pts || !raw || n < 0) return set_error(WOST_ERR_INVALID, "bad argument");
    std::vector<float> in((size_t)n * 3);
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) in[3 * (size_t)i + a] = 0.5f + (pts[3 * (size_t)i + a] - h->box.c[a]) / h->box.e[a];
    return wost_net_inference(h->net, in.data(), n, raw, 1);
}

// the training set of the most recent training pass, (pixel, record) order; arrays may be NULL; *n = its size
int wost3_guided_train_set(wost3_guided_handle h, int32_t capacity, int32_t *n, float *xyz, float *dir, float *solution, float *dir_pdf,
                           float *normal, uint8_t *on_neumann)
{
    if (!h || !n || capacity < 0) return set_error(WOST_ERR_INVALID, "bad argument");
    W3_TRY(hipSetDevice(h->device));
    *n = (int32_t)h->last_train_n;
    const size_t m = std::min((size_t)capacity, (size_t)h->last_train_n);
    if (m == 0) return WOST_OK;
    if (xyz) W3_TRY(hipMemcpy(xyz, h->t_x, m * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (dir) W3_TRY(hipMemcpy(dir, h->t_dir, m * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (solution) W3_TRY(hipMemcpy(solution, h->t_sol, m * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (dir_pdf) W3_TRY(hipMemcpy(dir_pdf, h->t_pdf, m * sizeof(float), hipMemcpyDeviceToHost));
    if (normal) W3_TRY(hipMemcpy(normal, h->t_nrm, m * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (on_neumann) W3_TRY(hipMemcpy(on_neumann, h->t_onn, m, hipMemcpyDeviceToHost));
    return WOST_OK;
}

}  // extern "C"
