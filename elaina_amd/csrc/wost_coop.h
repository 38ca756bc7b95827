// wost_coop.h -- the Neumann-side tree queries of a 2-D walk step (closest silhouette vertex, the walker's ray) answered by a
// whole WAVE for all its walkers together, on the loop of wost_pool.h.  Device code for gfx950, included by wost_hip.hip.
// Both queries are minima -- over silhouette vertices within rmax, over (t, original index) of the hits -- so the answers are
// those of the flat loops, bit for bit.  The 3-D twins (wost_hip3d.hip) are where the design was measured first.
#pragma once

#include "wost_pool.h"

namespace wost {

constexpr int kPoolOwnerWords = 8 * 64;      // per-owner words of the larger of the two queries (the ray)

// closest silhouette vertex within rmax of (qx, qy), for every lane with `active` (all 64 lanes must call)
template <class STK>
__device__ __forceinline__ float closest_silhouette_wave(const DevMesh &m, float qx, float qy, float rmax, bool active, const WavePool &W, const STK &stk)
{
    constexpr float kSlack = 1.000030517578125f;       // closest_silhouette_tree's
    const int lane = threadIdx.x & 63;
    float *oq = reinterpret_cast<float *>(W.own);                 // x [0, 64), y [64, 128)
    uint32_t *obest = W.own + 128, *ofound = W.own + 192;         // the flat loop's best2 (bits) and `found`
    if (active) {
        oq[lane] = qx; oq[64 + lane] = qy;
        obest[lane] = __float_as_uint(rmax * rmax);
        ofound[lane] = 0u;
    }
    const bool done = pool_run(
        W, m.levels, active, 64,
        [&](uint32_t g, int owner, bool leaf, bool (&v)[4], uint32_t (&key)[4]) {
            const float x = oq[owner], y = oq[64 + owner];
            const float bd = __uint_as_float(obest[owner]) * kSlack;
            const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 4u * WOST_NODE_FLOATS));
            const float4 CX = nd[0], CY = nd[1], UX = nd[2], UY = nd[3], HL = nd[4], HW = nd[5];
            const float d0 = obb_d2(CX.x, CY.x, UX.x, UY.x, HL.x, HW.x, x, y), d1 = obb_d2(CX.y, CY.y, UX.y, UY.y, HL.y, HW.y, x, y);
            const float d2 = obb_d2(CX.z, CY.z, UX.z, UY.z, HL.z, HW.z, x, y), d3 = obb_d2(CX.w, CY.w, UX.w, UY.w, HL.w, HW.w, x, y);
            if (leaf) {
                // (a vertex is an end of its segment: no closer than the segment, whose distance the leaf record gives)
                v[0] = !(d0 > bd); v[1] = !(d1 > bd); v[2] = !(d2 > bd); v[3] = !(d3 > bd);
            } else {
                const float4 *cn = m.cones + 5 * (size_t)g;
                const float4 AX = cn[0], AY = cn[1], CH = cn[2], SH = cn[3], RD = cn[4];
                const bool c0 = d0 <= bd && cone_may_hold_silhouette(AX.x, AY.x, CH.x, SH.x, RD.x, CX.x, CY.x, x, y);
                const bool c1 = d1 <= bd && cone_may_hold_silhouette(AX.y, AY.y, CH.y, SH.y, RD.y, CX.y, CY.y, x, y);
                const bool c2 = d2 <= bd && cone_may_hold_silhouette(AX.z, AY.z, CH.z, SH.z, RD.z, CX.z, CY.z, x, y);
                const bool c3 = d3 <= bd && cone_may_hold_silhouette(AX.w, AY.w, CH.w, SH.w, RD.w, CX.w, CY.w, x, y);
                key[0] = c0 ? ((__float_as_uint(d0) & ~0x3u) | 0u) : 0xffffffffu;
                key[1] = c1 ? ((__float_as_uint(d1) & ~0x3u) | 1u) : 0xffffffffu;
                key[2] = c2 ? ((__float_as_uint(d2) & ~0x3u) | 2u) : 0xffffffffu;
                key[3] = c3 ? ((__float_as_uint(d3) & ~0x3u) | 3u) : 0xffffffffu;
            }
        },
        [&](uint32_t slot, int owner) {
            const int2 vv = m.segVerts[slot];
            if (vv.x < 0) return;
            const float x = oq[owner], y = oq[64 + owner];
            const float b0 = __uint_as_float(obest[owner]);
            const bool f0 = ofound[owner] != 0u;
            float b = b0;
            bool f = f0;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const DevSilVertex sv = m.sil[c ? vv.y : vv.x];
                const float vx = x - sv.x, vy = y - sv.y;
                const float d2 = dot2(vx, vy, vx, vy);
                if (d2 > b) continue;
                if (vertex_is_silhouette(m, sv, vx, vy, d2) && (d2 < b || !f)) {
                    b = d2;
                    f = true;
                }
            }
            if (f && (b < b0 || !f0)) {
                atomicMin(&obest[owner], __float_as_uint(b));
                ofound[owner] = 1u;
            }
        });
    float r = WOST_INF;
    if (!done) {
        if (active) r = closest_silhouette_tree(m, qx, qy, rmax, stk);
    } else if (active && ofound[lane] != 0u) {
        r = sqrtf(__uint_as_float(obest[lane]));
    }
    wave_lds_fence();
    return r;
}

// the walker's ray: closest hit (smallest t, lowest original index among equal ones) for every lane with `active`
template <class STK>
__device__ __forceinline__ bool ray_closest_wave(const DevMesh &m, float ox, float oy, float dx, float dy, float tmax, bool active, float &t_out,
                                                 int &idx_out, const WavePool &W, const STK &stk, int slot_trigger)
{
    const int lane = threadIdx.x & 63;
    unsigned long long *okey = reinterpret_cast<unsigned long long *>(W.own);     // [64]: bits(|t|) << 32 | original index
    float *of = reinterpret_cast<float *>(W.own) + 128;                            // ox, oy, dx, dy, tmax: 5 x [64]
    uint32_t *obound = W.own + 128 + 5 * 64;                                       // the pruning limit (bits)
    if (active) {
        okey[lane] = ~0ull;
        of[lane] = ox; of[64 + lane] = oy; of[128 + lane] = dx; of[192 + lane] = dy; of[256 + lane] = tmax;
        obound[lane] = __float_as_uint(tmax);
    }
    const bool done = pool_run(
        W, m.levels, active, slot_trigger,
        [&](uint32_t g, int owner, bool leaf, bool (&v)[4], uint32_t (&key)[4]) {
            const float rx = of[owner], ry = of[64 + owner], rdx = of[128 + owner], rdy = of[192 + owner];
            const float lim = __uint_as_float(obound[owner]);
            const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 4u * WOST_NODE_FLOATS));
            const float4 CX = nd[0], CY = nd[1], UX = nd[2], UY = nd[3], HL = nd[4], HW = nd[5];
            const float e0 = CX.x >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.x, CY.x, UX.x, UY.x, HL.x, HW.x, rx, ry, rdx, rdy, lim);
            const float e1 = CX.y >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.y, CY.y, UX.y, UY.y, HL.y, HW.y, rx, ry, rdx, rdy, lim);
            const float e2 = CX.z >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.z, CY.z, UX.z, UY.z, HL.z, HW.z, rx, ry, rdx, rdy, lim);
            const float e3 = CX.w >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.w, CY.w, UX.w, UY.w, HL.w, HW.w, rx, ry, rdx, rdy, lim);
            if (leaf) {
                v[0] = e0 <= lim; v[1] = e1 <= lim; v[2] = e2 <= lim; v[3] = e3 <= lim;
            } else {
                key[0] = (e0 <= lim) ? ((__float_as_uint(e0) & ~0x3u) | 0u) : 0xffffffffu;
                key[1] = (e1 <= lim) ? ((__float_as_uint(e1) & ~0x3u) | 1u) : 0xffffffffu;
                key[2] = (e2 <= lim) ? ((__float_as_uint(e2) & ~0x3u) | 2u) : 0xffffffffu;
                key[3] = (e3 <= lim) ? ((__float_as_uint(e3) & ~0x3u) | 3u) : 0xffffffffu;
            }
        },
        [&](uint32_t slot, int owner) {
            const int o = m.segOrig[slot];
            if (o == WOST_FAR_INDEX) return;
            const DevFlatSeg s = m.flat[o];
            const float lim = of[256 + owner];       // (the exact test runs against tmax, like the flat loop's)
            float t;
            if (seg_ray(s, of[owner], of[64 + owner], of[128 + owner], of[192 + owner], lim, t)) {
                const float at = fabsf(t);           // (t may be -0)
                atomicMin(&okey[owner], ((unsigned long long)__float_as_uint(at) << 32) | (unsigned long long)(uint32_t)o);
                atomicMin(&obound[owner], __float_as_uint(fminf(at, lim)));
            }
        });
    bool hit = false;
    t_out = WOST_INF;
    idx_out = -1;
    if (!done) {
        if (active) hit = ray_tree<false>(m, ox, oy, dx, dy, tmax, t_out, idx_out, stk);
    } else if (active) {
        const unsigned long long key = okey[lane];
        if (key != ~0ull) {
            // the winner's parameter from its own test (the key holds |t|)
            idx_out = (int)(uint32_t)key;
            hit = seg_ray(m.flat[idx_out], ox, oy, dx, dy, tmax, t_out);
        }
    }
    wave_lds_fence();
    return hit;
}

}  // namespace wost
