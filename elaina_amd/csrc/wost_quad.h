// wost_quad.h -- one closest-point query shared by the four lanes of a quad (gfx950).
//
// The round kernel gives every lane its own walker and its own descent: a node visit is a chain of
// about 150 dependent vector instructions (four oriented-box distances one after the other, a
// five-exchange sort, three pushes, the pop), and a launch that does not fill the chip -- the tail of a
// solve, one rank's shard of a multi-GPU job, the drain of a one-sample pass -- lasts as long as the
// longest such chain.  Here the four lanes of a quad (lanes 4k .. 4k+3, the unit of the DPP
// quad_perm cross-lane operand) hold ONE walker: lane j measures child j, the sort is a rank
// computed from the three other lanes' keys, the three pushes are one LDS store, and the pop
// examines the four top entries at once.  The arithmetic per child, the keys, the stack contents
// and hence the visiting order are those of trav_visit (wost_device.h), so the answer and every
// counter are bit-identical; a visit is about 70 instructions instead of 150 and its node fetch is
// one 96-byte record per quad instead of six 16-byte gathers per lane.
// Everything else of a walk step runs replicated in the four lanes (same values, same instructions).
#pragma once

#include "wost_device.h"

namespace wost {

// quad_perm selectors of the DPP operand: lane j of a quad reads lane sel[j]
#define WOST_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))

template <int CTRL>
__device__ __forceinline__ uint32_t quad_read(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float quad_readf(float v)
{
    return __uint_as_float(quad_read<CTRL>(__float_as_uint(v)));
}

// the same tree of minima as fminf(fminf(d0, d1), fminf(d2, d3)), in every lane of the quad
__device__ __forceinline__ float quad_min_f(float v)
{
    v = fminf(v, quad_readf<WOST_QP(1, 0, 3, 2)>(v));
    return fminf(v, quad_readf<WOST_QP(2, 3, 0, 1)>(v));
}
__device__ __forceinline__ uint32_t quad_min_u(uint32_t v)
{
    v = min(v, quad_read<WOST_QP(1, 0, 3, 2)>(v));
    return min(v, quad_read<WOST_QP(2, 3, 0, 1)>(v));
}
__device__ __forceinline__ uint32_t quad_sum_u(uint32_t v)
{
    v += quad_read<WOST_QP(1, 0, 3, 2)>(v);
    return v + quad_read<WOST_QP(2, 3, 0, 1)>(v);
}

// The quad's column of the LDS traversal stack: entries are contiguous (entry i at col[i]), so the
// lanes of a quad that push or examine neighbouring entries touch neighbouring banks.
struct QuadColumn {
    uint32_t *col;
    __device__ __forceinline__ void put(int i, uint32_t key) const { col[i] = key; }
    __device__ __forceinline__ uint32_t get(int i) const { return col[i]; }
};

// trav_pop for a quad: lane j examines entry sp-1-j; the first live one (lowest j) is taken, everything
// above it is dropped -- what the one-entry-at-a-time loop of trav_pop does, four entries per trip.
__device__ __forceinline__ bool quad_pop(Trav &T, const QuadColumn &stk, int j, float bound)
{
    for (;;) {
        if (T.sp <= 0) return false;
        const int idx = T.sp - 1 - j;
        const uint32_t key = stk.get(max(idx, 0));
        const bool live = idx >= 0 && __uint_as_float(key & ~0x3Fu) <= bound;
        const uint32_t cand = live ? key : 0xffffffffu;        // a stack key is below 2^31: never all ones
        const uint32_t b0 = quad_read<WOST_QP(0, 0, 0, 0)>(cand), b1 = quad_read<WOST_QP(1, 1, 1, 1)>(cand);
        const uint32_t b2 = quad_read<WOST_QP(2, 2, 2, 2)>(cand), b3 = quad_read<WOST_QP(3, 3, 3, 3)>(cand);
        const int jl = (b0 != 0xffffffffu) ? 0 : (b1 != 0xffffffffu) ? 1 : (b2 != 0xffffffffu) ? 2 : (b3 != 0xffffffffu) ? 3 : 4;
        if (jl == 4) {
            T.sp = max(T.sp - 4, 0);
            continue;
        }
        const uint32_t k = (jl == 0) ? b0 : (jl == 1) ? b1 : (jl == 2) ? b2 : b3;
        T.sp = T.sp - 1 - jl;
        const int el = (int)((k >> 2) & 15u);
        const int parent = T.pos >> (2 * (T.level - el + 1));
        T.pos = 4 * parent + (int)(k & 3u);
        T.level = el;
        return true;
    }
}

// trav_visit for a quad: ONE node, lane j = child j.  Returns false when the query is complete.  T is
// replicated in the four lanes and stays so.
template <bool SLACK>
__device__ __forceinline__ bool quad_visit(const DevMesh &m, float qx, float qy, Trav &T, const QuadColumn &stk, int j)
{
    const uint32_t g = level_first(T.level) + (uint32_t)T.pos;
    const float *nd = reinterpret_cast<const float *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 4u * WOST_NODE_FLOATS)) + j;
    const float cx = nd[0], cy = nd[4], ux = nd[8], uy = nd[12], hl = nd[16], hw = nd[20];
    const float bd = T.best.d2;
    const bool at_leaf = T.level == m.levels;
    float d = obb_d2(cx, cy, ux, uy, hl, hw, qx, qy);
    if (SLACK) d = d * (at_leaf ? 1.0f : kBoxShrink);
    // ---- last level: exact segment distances; one strict winner is the common case ----
    {
        // (squared distances are non-negative and never NaN: their bit patterns order like the values, and the integer
        // minimum folds into the DPP operand -- the same result as fminf(fminf(d0, d1), fminf(d2, d3)))
        const float mn = __uint_as_float(quad_min_u(__float_as_uint(d)));
        const bool eq = d == mn;
        const int n_eq = (int)quad_sum_u(eq ? 1u : 0u);
        const int jmin = (int)quad_min_u(eq ? (uint32_t)j : 4u);
        const int slot = 4 * T.pos + jmin;
        const bool clean = n_eq == 1;
        const bool win = at_leaf && clean && mn < bd;
        const bool tie = at_leaf && mn <= bd && !win && !(clean && slot == T.best.slot);
        T.best.d2 = win ? mn : T.best.d2;
        T.best.slot = win ? slot : T.best.slot;
        T.best_orig = win ? -1 : T.best_orig;
        if (tie)      // (quad-uniform) the rare path on all four distances, replicated
            trav_leaf_ties(m, T, 4 * T.pos, quad_readf<WOST_QP(0, 0, 0, 0)>(d), quad_readf<WOST_QP(1, 1, 1, 1)>(d),
                           quad_readf<WOST_QP(2, 2, 2, 2)>(d), quad_readf<WOST_QP(3, 3, 3, 3)>(d));
    }
    // ---- inner level: the key of trav_visit; an invalid child gets a key above every valid one that still differs
    // between the lanes, so the rank among the four keys is a total order: valid children by (distance, index), then the rest
    {
        const uint32_t tag = (uint32_t)(T.level + 1) << 2;
        const bool valid = !at_leaf && d <= bd;
        const uint32_t key = valid ? ((__float_as_uint(d) & ~0x3Fu) | tag | (uint32_t)j) : (0xfffffffcu | (uint32_t)j);
        const uint32_t k1 = quad_read<WOST_QP(1, 2, 3, 0)>(key), k2 = quad_read<WOST_QP(2, 3, 0, 1)>(key), k3 = quad_read<WOST_QP(3, 0, 1, 2)>(key);
        const int rank = (int)(k1 < key) + (int)(k2 < key) + (int)(k3 < key);
        const uint32_t kmin = min(min(key, k1), min(k2, k3));
        const int nv = (int)(key < 0x80000000u) + (int)(k1 < 0x80000000u) + (int)(k2 < 0x80000000u) + (int)(k3 < 0x80000000u);
        // the sorted valid keys but the nearest go on the stack far to near: rank nv-1 at sp, ..., rank 1 at sp+nv-2
        if (valid && rank >= 1) stk.put(T.sp + nv - 1 - rank, key);
        // the entries one lane of the quad has written are read by the other three (quad_pop, the next visit): an LDS fence
        // inside the wave, so that the compiler cannot move a later load above these stores (the hardware keeps a wave's LDS
        // operations in order: no instruction is spent)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        T.sp += max(nv - 1, 0);
        if (kmin < 0x80000000u) {
            T.pos = 4 * T.pos + (int)(kmin & 3u);
            T.level = T.level + 1;
            return true;
        }
    }
    return quad_pop(T, stk, j, T.best.d2);
}

}  // namespace wost
