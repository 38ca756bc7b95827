// wost_coop.h -- the Neumann-side tree queries of a walk step (closest silhouette vertex, the walker's ray) answered by a whole
// WAVE for all its walkers together, 2-D.  Device code for gfx950, included by wost_hip.hip; not part of the C-ABI.
//
// Inside step_finish every lane runs its own query to completion (closest_silhouette_tree, ray_tree): a wave lasts as long as
// its longest query, and a last-level visit -- eight vertex tests or four exact ray / segment tests behind per-lane skips -- is
// executed for the whole wave whenever one lane needs it.  Here the work of all walkers of the wave goes through two pools of
// tasks in LDS -- (owner lane, tree node) and (owner lane, leaf slot) -- and every trip runs ONE body on up to 64 tasks, whoever
// owns them: a node task measures the four children (and their normal cones) against its owner's bound and pushes those that
// survive, the farthest first, so that the next trip takes every task's nearest child off the top; a slot task tests the two
// vertices (or the segment) of one leaf slot and folds the result into its owner's words with LDS atomics.  Both queries are
// minima -- over silhouette vertices within rmax, over (t, original index) of the hits -- so the order in which candidates are
// met does not matter and the answer is that of the flat loop, bit for bit; a bound that is tightened later than the private
// descent would have only costs visits.  The 3-D twin (wost_hip3d.hip) is where the design was measured first.
// The pools are bounded: a trip takes only as many node tasks as leave room for all their children, and a wave that cannot take
// any (pool full of inner nodes) answers its queries the old way.
#pragma once

#include "wost_device.h"

namespace wost {

struct WavePool {
    uint32_t *node, *slot;      // [cap] owner lane << 26 | node index / leaf slot
    uint32_t *own;              // [8][64] per-owner operands and results
    int cap;
};
constexpr uint32_t kPoolIndex = (1u << 26) - 1u;
constexpr int kPoolOwnerWords = 8 * 64;

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS instructions of one wave execute in order: only the compiler has to keep the order
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void pool_push(uint32_t *pool, int &n, bool valid, uint32_t value, int lane)
{
    const unsigned long long mask = __ballot(valid);
    if (valid) pool[n + __popcll(mask & ((1ull << lane) - 1ull))] = value;
    n += __popcll(mask);
}

__device__ __forceinline__ void node_level_pos(uint32_t g, int &level, uint32_t &pos)
{
    level = (31 - __clz((int)(3u * g + 1u))) >> 1;
    pos = g - level_first(level);
}

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
    int n_node = 0, n_slot = 0;
    pool_push(W.node, n_node, active, (uint32_t)lane << 26, lane);        // the roots
    bool overflow = false;
    wave_lds_fence();
    while (n_node > 0 || n_slot > 0) {
        if (n_slot >= 64 || n_node == 0) {
            const int k = min(64, n_slot);
            n_slot -= k;
            if (lane < k) {
                const uint32_t e = W.slot[n_slot + lane];
                const int owner = (int)(e >> 26);
                const int2 vv = m.segVerts[e & kPoolIndex];
                if (vv.x >= 0) {
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
                }
            }
        } else {
            const int k = min(min(64, n_node), (W.cap - n_node) / 3);
            if (k <= 0 || n_slot + 4 * k > W.cap) {
                overflow = true;
                break;
            }
            n_node -= k;
            const bool t = lane < k;
            const uint32_t e = t ? W.node[n_node + lane] : 0u;
            wave_lds_fence();        // the tasks are read before the pushes below overwrite them
            const uint32_t own_bits = e & ~kPoolIndex;
            bool leaf = false, v0 = false, v1 = false, v2 = false, v3 = false;
            uint32_t k0 = 0xffffffffu, k1 = 0xffffffffu, k2 = 0xffffffffu, k3 = 0xffffffffu;
            uint32_t child0 = 0;     // the first child: node index (inner) or leaf slot
            if (t) {
                const uint32_t g = e & kPoolIndex;
                const int owner = (int)(e >> 26);
                int level;
                uint32_t pos;
                node_level_pos(g, level, pos);
                const float x = oq[owner], y = oq[64 + owner];
                const float bd = __uint_as_float(obest[owner]) * kSlack;
                const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 96u));
                const float4 CX = nd[0], CY = nd[1], UX = nd[2], UY = nd[3], HL = nd[4], HW = nd[5];
                const float d0 = obb_d2(CX.x, CY.x, UX.x, UY.x, HL.x, HW.x, x, y), d1 = obb_d2(CX.y, CY.y, UX.y, UY.y, HL.y, HW.y, x, y);
                const float d2 = obb_d2(CX.z, CY.z, UX.z, UY.z, HL.z, HW.z, x, y), d3 = obb_d2(CX.w, CY.w, UX.w, UY.w, HL.w, HW.w, x, y);
                leaf = level == m.levels;
                if (leaf) {
                    // (a vertex is an end of its segment: no closer than the segment, whose distance the leaf record gives)
                    child0 = 4u * pos;
                    v0 = !(d0 > bd); v1 = !(d1 > bd); v2 = !(d2 > bd); v3 = !(d3 > bd);
                } else {
                    child0 = level_first(level + 1) + 4u * pos;
                    const float4 *cn = m.cones + 5 * (size_t)g;
                    const float4 AX = cn[0], AY = cn[1], CH = cn[2], SH = cn[3], RD = cn[4];
                    const bool c0 = d0 <= bd && cone_may_hold_silhouette(AX.x, AY.x, CH.x, SH.x, RD.x, CX.x, CY.x, x, y);
                    const bool c1 = d1 <= bd && cone_may_hold_silhouette(AX.y, AY.y, CH.y, SH.y, RD.y, CX.y, CY.y, x, y);
                    const bool c2 = d2 <= bd && cone_may_hold_silhouette(AX.z, AY.z, CH.z, SH.z, RD.z, CX.z, CY.z, x, y);
                    const bool c3 = d3 <= bd && cone_may_hold_silhouette(AX.w, AY.w, CH.w, SH.w, RD.w, CX.w, CY.w, x, y);
                    k0 = c0 ? ((__float_as_uint(d0) & ~0x3u) | 0u) : 0xffffffffu;
                    k1 = c1 ? ((__float_as_uint(d1) & ~0x3u) | 1u) : 0xffffffffu;
                    k2 = c2 ? ((__float_as_uint(d2) & ~0x3u) | 2u) : 0xffffffffu;
                    k3 = c3 ? ((__float_as_uint(d3) & ~0x3u) | 3u) : 0xffffffffu;
                    cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2);
                }
            }
            // a leaf's segments within the bound become slot tasks ...
            pool_push(W.slot, n_slot, leaf && v0, own_bits | (child0 + 0u), lane);
            pool_push(W.slot, n_slot, leaf && v1, own_bits | (child0 + 1u), lane);
            pool_push(W.slot, n_slot, leaf && v2, own_bits | (child0 + 2u), lane);
            pool_push(W.slot, n_slot, leaf && v3, own_bits | (child0 + 3u), lane);
            // ... an inner node's children node tasks, the farthest first: every task's nearest child ends up in the top 64
            pool_push(W.node, n_node, k3 != 0xffffffffu, own_bits | (child0 + (k3 & 3u)), lane);
            pool_push(W.node, n_node, k2 != 0xffffffffu, own_bits | (child0 + (k2 & 3u)), lane);
            pool_push(W.node, n_node, k1 != 0xffffffffu, own_bits | (child0 + (k1 & 3u)), lane);
            pool_push(W.node, n_node, k0 != 0xffffffffu, own_bits | (child0 + (k0 & 3u)), lane);
        }
        wave_lds_fence();
    }
    float r = WOST_INF;
    if (overflow) {
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
    int n_node = 0, n_slot = 0;
    pool_push(W.node, n_node, active, (uint32_t)lane << 26, lane);
    bool overflow = false;
    wave_lds_fence();
    while (n_node > 0 || n_slot > 0) {
        if (n_slot >= slot_trigger || n_node == 0) {
            const int k = min(64, n_slot);
            n_slot -= k;
            if (lane < k) {
                const uint32_t e = W.slot[n_slot + lane];
                const int owner = (int)(e >> 26);
                const int o = m.segOrig[e & kPoolIndex];
                if (o != WOST_FAR_INDEX) {
                    const DevFlatSeg s = m.flat[o];
                    const float lim = of[256 + owner];
                    float t;
                    if (seg_ray(s, of[owner], of[64 + owner], of[128 + owner], of[192 + owner], lim, t)) {
                        const float at = fabsf(t);       // (t may be -0)
                        atomicMin(&okey[owner], ((unsigned long long)__float_as_uint(at) << 32) | (unsigned long long)(uint32_t)o);
                        atomicMin(&obound[owner], __float_as_uint(fminf(at, lim)));
                    }
                }
            }
        } else {
            const int k = min(min(64, n_node), (W.cap - n_node) / 3);
            if (k <= 0 || n_slot + 4 * k > W.cap) {
                overflow = true;
                break;
            }
            n_node -= k;
            const bool t = lane < k;
            const uint32_t e = t ? W.node[n_node + lane] : 0u;
            wave_lds_fence();
            const uint32_t own_bits = e & ~kPoolIndex;
            bool leaf = false, v0 = false, v1 = false, v2 = false, v3 = false;
            uint32_t k0 = 0xffffffffu, k1 = 0xffffffffu, k2 = 0xffffffffu, k3 = 0xffffffffu;
            uint32_t child0 = 0;
            if (t) {
                const uint32_t g = e & kPoolIndex;
                const int owner = (int)(e >> 26);
                int level;
                uint32_t pos;
                node_level_pos(g, level, pos);
                const float rx = of[owner], ry = of[64 + owner], rdx = of[128 + owner], rdy = of[192 + owner];
                const float lim = __uint_as_float(obound[owner]);
                const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 96u));
                const float4 CX = nd[0], CY = nd[1], UX = nd[2], UY = nd[3], HL = nd[4], HW = nd[5];
                const float e0 = CX.x >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.x, CY.x, UX.x, UY.x, HL.x, HW.x, rx, ry, rdx, rdy, lim);
                const float e1 = CX.y >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.y, CY.y, UX.y, UY.y, HL.y, HW.y, rx, ry, rdx, rdy, lim);
                const float e2 = CX.z >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.z, CY.z, UX.z, UY.z, HL.z, HW.z, rx, ry, rdx, rdy, lim);
                const float e3 = CX.w >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.w, CY.w, UX.w, UY.w, HL.w, HW.w, rx, ry, rdx, rdy, lim);
                leaf = level == m.levels;
                if (leaf) {
                    child0 = 4u * pos;
                    v0 = e0 <= lim; v1 = e1 <= lim; v2 = e2 <= lim; v3 = e3 <= lim;
                } else {
                    child0 = level_first(level + 1) + 4u * pos;
                    k0 = (e0 <= lim) ? ((__float_as_uint(e0) & ~0x3u) | 0u) : 0xffffffffu;
                    k1 = (e1 <= lim) ? ((__float_as_uint(e1) & ~0x3u) | 1u) : 0xffffffffu;
                    k2 = (e2 <= lim) ? ((__float_as_uint(e2) & ~0x3u) | 2u) : 0xffffffffu;
                    k3 = (e3 <= lim) ? ((__float_as_uint(e3) & ~0x3u) | 3u) : 0xffffffffu;
                    cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2);
                }
            }
            pool_push(W.slot, n_slot, leaf && v0, own_bits | (child0 + 0u), lane);
            pool_push(W.slot, n_slot, leaf && v1, own_bits | (child0 + 1u), lane);
            pool_push(W.slot, n_slot, leaf && v2, own_bits | (child0 + 2u), lane);
            pool_push(W.slot, n_slot, leaf && v3, own_bits | (child0 + 3u), lane);
            pool_push(W.node, n_node, k3 != 0xffffffffu, own_bits | (child0 + (k3 & 3u)), lane);
            pool_push(W.node, n_node, k2 != 0xffffffffu, own_bits | (child0 + (k2 & 3u)), lane);
            pool_push(W.node, n_node, k1 != 0xffffffffu, own_bits | (child0 + (k1 & 3u)), lane);
            pool_push(W.node, n_node, k0 != 0xffffffffu, own_bits | (child0 + (k0 & 3u)), lane);
        }
        wave_lds_fence();
    }
    bool hit = false;
    t_out = WOST_INF;
    idx_out = -1;
    if (overflow) {
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
