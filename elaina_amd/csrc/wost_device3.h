// wost_device3.h -- device side of the 3-D Walk-on-Stars path shared by its translation units (wost_hip3d.hip: the uniform
// integrator and the C-ABI of the scene; wost_guided3.hip: GuidedIntegrator<3>): vectors, the triangle / edge records and
// the mesh view, closest triangle, closest silhouette edge and ray queries -- one descent per lane or answered by the wave
// through its LDS task pools (wost_pool.h) --, the index-ordered sampling sweeps, the dense-grid source term.  gfx950 only.
// Arithmetic contract: DESIGN.md 2.3.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "wost_device.h"
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

}  // namespace wost
