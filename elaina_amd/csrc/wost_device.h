// wost_device.h -- device-side geometry queries and the per-step walk logic (gfx950).
//
// Everything the wavefront loop of the reference's uniform integrator does per walk item
// (integrator/uniform/integrator.cu:128-211 separate, :224-231 handleBoundary, :336-444
// sampleNeumann, :465-525 oneStepWalk) is one inlined device function here, so that a walk
// item never leaves registers between stages.  The lbvh queries it needs are written
// against the implicit 4-ary LBVH of lbvh.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wost_math.h"

namespace wost {

#define WOST_R_B_FLOOR 1e-4f
#define WOST_R_B_SHRINK 0.99f
#define WOST_SIL_PRECISION 1e-3f
#define WOST_FAR_INDEX 0x7fffffff
#define WOST_INF __builtin_inff()
// floats between the child records of consecutive nodes: 24 (the six operand vectors, 96 bytes) or, as an experiment, 32 -- one node per
// 128-byte line, never two lines per fetch, a third more memory (EXPERIMENTS 22)
#ifndef WOST_NODE_FLOATS
#define WOST_NODE_FLOATS 24
#endif

// ---- device views ----------------------------------------------------------------------
struct DevFlatSeg {
    float ax, ay, ex, ey;
    float inv_len2, len, nx, ny;
    float cx, cy, ux, uy;
    float hl, pad0, pad1, pad2;
};
struct DevSilVertex {
    float x, y;
    int32_t prev, next;
};

struct DevMesh {
    const float4 *nodes;     // [n_nodes * 6]: cx[4] cy[4] ux[4] uy[4] hl[4] hw[4] of the children
    const float4 *segA;      // [slots] ax, ay, ex, ey
    const float *segInv;     // [slots] 1/|e|^2
    const int32_t *segOrig;  // [slots]
    const float *segCol;     // [slots*12]
    const DevFlatSeg *flat;  // [n_segs] original order
    const float *flatCol;    // [n_segs*12]
    const DevSilVertex *sil; // [n_sil] one per mesh vertex
    const float4 *silN;      // [n_sil] the unit normals of the vertex's two segments (prev.nx, prev.ny, next.nx, next.ny; zeros where there is none)
    const float4 *cones;     // [n_nodes * 5]: SNCH cones of the children: ax[4] ay[4] cos[4] sin[4] rad[4]
    const int2 *segVerts;    // [slots] vertex ids of the slot's segment
    int32_t n_segs;
    int32_t n_sil;
    int32_t levels;
    int32_t first_leaf;
    int32_t emissive;        // any non-zero colour
    float far2;              // squared distance from the mesh beyond which box pruning needs the relative slack (trav_visit<true>)
    float huge2;             // squared distance beyond which a closest-point query is answered by a scan of the whole wave (closest_point_wave)
    // the segments once more, compact and in slot order, for that scan: (cx, cy, ux, uy) + half length = the operands of a leaf
    // visit's distance, the slot and the original index; padded to a multiple of 256 with records that cannot win
    const float4 *scanBox;
    const float *scanHl;
    const int2 *scanId;      // (slot, original index)
    int32_t n_scan;
    // boxes over runs of consecutive ORIGINAL indices (sample_in_sphere_tree): level l holds one box (lo.x, lo.y,
    // hi.x, hi.y) per run of 4^(l+1) segments, obox + obox_off[l]; obox_levels = 0: not built
    const float4 *obox;
    int32_t obox_off[12];
    int32_t obox_levels;
    // compact copies for those sweeps, original order, padded to a multiple of four segments (length 0): a run of four is
    // fetched with six loads issued together instead of four dependent 64-byte records
    const float *lens;       // lengths
    const float4 *sampBox;   // cx cy ux uy of seg_d2
    const float *sampHl;     // hl of seg_d2
};

struct DevProbe {
    float scale, posx, posy, upx, upy;
};

// source term: dense RGB grid, index = world * scale + offset (see wost_source_desc)
struct DevSource {
    const float *rgb;   // nullptr = no source term
    int32_t nx, ny;
    float sx, sy, ox, oy;
    float intensity;
};

struct DevSettings {
    int32_t width, height, spp, max_depth;
    float eps;
    float dirichlet_intensity, neumann_intensity;
};

// ---- block-level stream compaction (shared by the uniform and the guided kernels) --------------
// Up to 1024 threads, every thread of the block must call it: ballot + popcount inside a wave, one
// atomic per block on the queue counter; returns the output slot of this lane (valid when `keep`).
__device__ __forceinline__ uint32_t block_push(bool keep, uint32_t *counter)
{
    __shared__ uint32_t s_cnt[16], s_base;
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63) >> 6;
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int w = 0; w < n_waves; ++w) total += s_cnt[w];
        s_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t base = s_base;
    for (int w = 0; w < wave; ++w) base += s_cnt[w];
    return base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
}

// ---- evaluation grid (reference core/evaluation_grid.h:27-33) --------------------------
__device__ __forceinline__ void eval_point(const DevProbe &p, int px, int py, int width, int height,
                                           float &x, float &y)
{
    float ndcx = 2.0f * (float)px / (float)width + -1.0f;
    float ndcy = 2.0f * (float)py / (float)height + -1.0f;
    float ux = p.upy, uy = -p.upx;
    float vx = p.upx, vy = p.upy;
    x = p.scale * (ndcx * ux + ndcy * vx) + p.posx;
    y = p.scale * (ndcx * uy + ndcy * vy) + p.posy;
}

// ---- squared distance to an oriented box / a segment (DESIGN.md "segment distance") ------
// Box = centre c, unit axis u, half length hl along u, half width hw across it.  A segment
// is the box with hw = 0; for it this IS the distance of the arithmetic contract.
__device__ __forceinline__ float obb_d2(float cx, float cy, float ux, float uy, float hl, float hw, float qx, float qy)
{
    const float wx = qx - cx, wy = qy - cy;
    const float u = dot2(wx, wy, ux, uy);
    const float v = cross2(ux, uy, wx, wy);
    const float du = fmaxf(fabsf(u) - hl, 0.0f);
    const float dv = fmaxf(fabsf(v) - hw, 0.0f);
    return dot2(du, dv, du, dv);
}

__device__ __forceinline__ float seg_d2(const DevFlatSeg &s, float qx, float qy)
{
    return obb_d2(s.cx, s.cy, s.ux, s.uy, s.hl, 0.0f, qx, qy);
}

// result of a closest-point query: slot in sorted order + squared distance
struct Closest {
    float d2;
    int32_t slot;
};

__device__ __forceinline__ void cswap(uint32_t &a, uint32_t &b)
{
    uint32_t lo = min(a, b), hi = max(a, b);
    a = lo;
    b = hi;
}

// Traversal state of one closest-point query on the wide LBVH.
//
// A node is addressed as (level, pos): heap index = first(level) + pos with
// first(L) = (4^L - 1) / 3 = 0x55555555 >> (32 - 2L); its children are (level+1, 4*pos+j);
// the children of a node of the last level (level == mesh.levels) are the segment slots
// 4*pos+j.  `stack` is the lane's column of the LDS traversal stack: entry i lives at
// stack[i*stride].  A stack entry is ONE 32-bit word: the child's box distance with its 6
// low mantissa bits replaced by (level << 2 | j).  The truncated distance is a lower bound
// of the true one, so stale entries are dropped at pop time without touching memory, and
// the node position is recovered from the position of the last visited node (which always
// lies below the entry's parent in a depth-first traversal):
// parent pos = pos >> 2*(level - entry_level + 1).
// relative slack of the box-against-best comparisons of trav_visit<true>: 1 - 2^-17.  The box distance and the exact segment
// distance each carry a few 10^-7 of relative rounding; a looser slack (10^-4 was tried) is as exact but opens every box of
// the mesh for a query thousands of scene sizes away -- 73 ms for the 64 steps of ONE leaked walker on ladybug
constexpr float kBoxShrink = 0.99999237060546875f;

struct Trav {
    int32_t level;      // level of the node to visit next
    int32_t pos;        // position of that node inside its level
    int32_t sp;         // number of entries on the stack
    Closest best;       // may start from a valid candidate (temporal hint)
    int32_t best_orig;  // original index of best.slot, loaded lazily when an exact tie shows up
};

__device__ __forceinline__ Trav trav_begin(Closest seed)
{
    return Trav{0, 0, 0, seed, -1};
}

__device__ __forceinline__ uint32_t level_first(int level)
{
    return level == 0 ? 0u : (0x55555555u >> (32 - 2 * level));
}

// A lane's column of the LDS traversal stack: entry i at col[i * stride] (bank = lane).  Block sizes
// are powers of two, so the entry address is ONE v_lshl_add_u32 (a 24-bit multiply plus a shift-add
// cost two half-rate instructions per access: tools/micro/op_rate.hip).
struct LdsColumn {
    uint32_t *col;
    uint32_t shift;   // log2 of the words between consecutive entries
    __device__ __forceinline__ LdsColumn(uint32_t *c, uint32_t stride) : col(c), shift(31u - (uint32_t)__builtin_clz(stride)) {}
    __device__ __forceinline__ void put(int i, uint32_t key) const { col[(uint32_t)i << shift] = key; }
    __device__ __forceinline__ uint32_t get(int i) const { return col[(uint32_t)i << shift]; }
};

// Pop the next entry that can still tie or beat the current best.  Returns false when the
// stack is exhausted (query complete).
// `bound` = what an entry's (truncated) box distance may not exceed: the best distance so far, or a slightly larger
// number where box and primitive distances come from different formulas (3-D)
template <class STK>
__device__ __forceinline__ bool trav_pop(Trav &T, const STK &stk, float bound)
{
    // Divergent branches are what this kernel pays most for (scalar exec-mask traffic), so the
    // common case -- the first or second entry is live -- runs as straight-line predicated
    // code: the LDS read is unconditional (index clamped), everything else is a select.
    bool need = true;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = max(T.sp - 1, 0);
        const uint32_t key = stk.get(idx);
        const bool can = need && T.sp > 0;
        const float dlb = __uint_as_float(key & ~0x3Fu);
        const bool take = can && dlb <= bound;
        const int el = (int)((key >> 2) & 15u);
        const int parent = T.pos >> (2 * (T.level - el + 1));
        T.pos = take ? 4 * parent + (int)(key & 3u) : T.pos;
        T.level = take ? el : T.level;
        T.sp = can ? T.sp - 1 : T.sp;
        need = need && !take;
    }
    while (need && T.sp > 0) {
        --T.sp;
        const uint32_t key = stk.get(T.sp);
        const float dlb = __uint_as_float(key & ~0x3Fu);
        if (dlb <= bound) {
            const int el = (int)((key >> 2) & 15u);
            const int parent = T.pos >> (2 * (T.level - el + 1));
            T.pos = 4 * parent + (int)(key & 3u);
            T.level = el;
            need = false;
        }
    }
    return !need;
}

template <class STK>
__device__ __forceinline__ bool trav_pop(Trav &T, const STK &stk)
{
    return trav_pop(T, stk, T.best.d2);
}

// rare path of a leaf visit: an exact tie between candidates; lowest ORIGINAL index wins
__device__ __forceinline__ void trav_leaf_ties(const DevMesh &m, Trav &T, int slot0, float e0, float e1, float e2, float e3)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ek = (k == 0) ? e0 : (k == 1) ? e1 : (k == 2) ? e2 : e3;
        const int slot = slot0 + k;
        if (ek < T.best.d2) {
            T.best.d2 = ek;
            T.best.slot = slot;
            T.best_orig = -1;
        } else if (ek == T.best.d2 && slot != T.best.slot) {
            if (T.best_orig < 0) T.best_orig = (T.best.slot >= 0) ? m.segOrig[T.best.slot] : WOST_FAR_INDEX;
            const int o = m.segOrig[slot];
            if (o < T.best_orig) {
                T.best.slot = slot;
                T.best_orig = o;
            }
        }
    }
}

// Visit ONE node: measure the four children (oriented boxes, or the segments themselves on
// the last level -- same arithmetic), then either update the best candidate (last level) or
// push the children that can still tie or win in far-to-near order and step into the
// nearest.  Returns false when the query is complete.  Ties between segments are broken by
// the lowest ORIGINAL index, so the answer does not depend on the tree or the visiting order.
// (An LDS mirror of the top levels of the tree was measured at +-0 % in round 1 and cost a flat
// load path with its own branch per visit; the top of the tree is L1-resident anyway.)
// SLACK = true prunes boxes with a relative slack as well (their distance shrunk by 10^-4): box and segment distances come
// from different formulas, and for a query far outside the mesh one ulp of the squared distance exceeds the absolute
// padding of the boxes -- a closer or tying segment would be skipped (found by tools/fuzz/fuzz_parity.py: probes and
// escaped walkers 50 scene sizes away).  Within DevMesh::far2 of the mesh the padding covers the rounding and the plain
// form is exact; it is the one the walk kernels run in their lane machines, where a single extra live register costs a
// quarter of the throughput; a query that starts beyond far2 is answered by closest_point_far instead.
template <bool SLACK = false, class STK = LdsColumn>
__device__ __forceinline__ bool trav_visit(const DevMesh &m, float qx, float qy, Trav &T, const STK &stk)
{
    const uint32_t g = level_first(T.level) + (uint32_t)T.pos;
    // 96-byte nodes; the byte offset stays below 4 GiB
    const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 4u * WOST_NODE_FLOATS));
    const float4 CX = nd[0], CY = nd[1], UX = nd[2], UY = nd[3], HL = nd[4], HW = nd[5];
    // plain scalar fp32: on gfx950 a packed v_pk_fma_f32 costs two v_fma_f32 (tools/micro/op_rate.hip,
    // profiles/r02_micro_*), has no |x| source modifier and needs hazard nops; the scalar form
    // spends 12 instructions per child with the absolute values folded into the subtractions
    const float bd = T.best.d2;
    const bool at_leaf = T.level == m.levels;
    const float shrink = (SLACK && !at_leaf) ? kBoxShrink : 1.0f;
    const float d0 = SLACK ? obb_d2(CX.x, CY.x, UX.x, UY.x, HL.x, HW.x, qx, qy) * shrink : obb_d2(CX.x, CY.x, UX.x, UY.x, HL.x, HW.x, qx, qy);
    const float d1 = SLACK ? obb_d2(CX.y, CY.y, UX.y, UY.y, HL.y, HW.y, qx, qy) * shrink : obb_d2(CX.y, CY.y, UX.y, UY.y, HL.y, HW.y, qx, qy);
    const float d2 = SLACK ? obb_d2(CX.z, CY.z, UX.z, UY.z, HL.z, HW.z, qx, qy) * shrink : obb_d2(CX.z, CY.z, UX.z, UY.z, HL.z, HW.z, qx, qy);
    const float d3 = SLACK ? obb_d2(CX.w, CY.w, UX.w, UY.w, HL.w, HW.w, qx, qy) * shrink : obb_d2(CX.w, CY.w, UX.w, UY.w, HL.w, HW.w, qx, qy);
#ifdef WOST_EXP_VALU
    {   // sensitivity experiment (developer builds only): extra dependent vector instructions per visit
        float acc = qx;
#pragma unroll
        for (int k = 0; k < WOST_EXP_VALU; ++k) acc = __builtin_fmaf(acc, 1.0000001f, d0);
        if (acc == 12345.678f) T.best_orig = 0;
    }
#endif
#ifdef WOST_EXP_LOADS
    {   // sensitivity experiment (developer builds only): a second, unrelated node fetched per visit
        const uint32_t nn = level_first(m.levels + 1);
        uint32_t g2 = g + (nn >> 1);
        g2 = g2 >= nn ? g2 - nn : g2;
        const float4 *n2 = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g2, 4u * WOST_NODE_FLOATS));
        const float acc = n2[0].x + n2[1].x + n2[2].x + n2[3].x + n2[4].x + n2[5].x;
        if (acc == 12345.678f) T.best_orig = 0;
    }
#endif
    // ---- last level: the children are segments, the distances are exact.  Branch-free in
    // the common case (one strict winner); exact ties take the rare path.
    {
        const float mn = fminf(fminf(d0, d1), fminf(d2, d3));
        const int n_eq = (d0 == mn) + (d1 == mn) + (d2 == mn) + (d3 == mn);
        const int slot = 4 * T.pos + ((d0 == mn) ? 0 : (d1 == mn) ? 1 : (d2 == mn) ? 2 : 3);
        const bool clean = (n_eq == 1);
        const bool win = at_leaf && clean && mn < bd;
        // same segment as the current best (the temporal hint met again) needs nothing
        const bool tie = at_leaf && mn <= bd && !win && !(clean && slot == T.best.slot);
        T.best.d2 = win ? mn : T.best.d2;
        T.best.slot = win ? slot : T.best.slot;
        T.best_orig = win ? -1 : T.best_orig;
        if (tie) trav_leaf_ties(m, T, 4 * T.pos, d0, d1, d2, d3);
    }
    // ---- inner level: near-first order.  The (non-negative) distance orders as an integer;
    // level and child index ride in the low mantissa bits, so five integer compare-exchanges
    // sort the candidates.  Lanes at the last level simply have no valid key.
    {
        const uint32_t tag = (uint32_t)(T.level + 1) << 2;
        const bool inner = !at_leaf;
        uint32_t k0 = (inner && d0 <= bd) ? ((__float_as_uint(d0) & ~0x3Fu) | tag | 0u) : 0xffffffffu;
        uint32_t k1 = (inner && d1 <= bd) ? ((__float_as_uint(d1) & ~0x3Fu) | tag | 1u) : 0xffffffffu;
        uint32_t k2 = (inner && d2 <= bd) ? ((__float_as_uint(d2) & ~0x3Fu) | tag | 2u) : 0xffffffffu;
        uint32_t k3 = (inner && d3 <= bd) ? ((__float_as_uint(d3) & ~0x3Fu) | tag | 3u) : 0xffffffffu;
        cswap(k0, k1);
        cswap(k2, k3);
        cswap(k0, k2);
        cswap(k1, k3);
        cswap(k1, k2);
        // branch-free pushes: sorted keys keep the invalid ones (0xffffffff) last, i.e. first
        // in far-to-near push order; an invalid key is written but the pointer does not move,
        // so the next write lands on top of it (the column has three words of slack)
        int sp = T.sp;
        stk.put(sp, k3);
        sp += (k3 != 0xffffffffu) ? 1 : 0;
        stk.put(sp, k2);
        sp += (k2 != 0xffffffffu) ? 1 : 0;
        stk.put(sp, k1);
        sp += (k1 != 0xffffffffu) ? 1 : 0;
        T.sp = sp;
        const bool descend = k0 != 0xffffffffu;
        T.pos = descend ? 4 * T.pos + (int)(k0 & 3u) : T.pos;
        T.level = descend ? T.level + 1 : T.level;
        if (descend) return true;
    }
    return trav_pop(T, stk);
}

__device__ __forceinline__ Closest closest_point(const DevMesh &m, float qx, float qy, Closest seed,
                                                 uint32_t *stack, int stride)
{
    Trav T = trav_begin(seed);
    const LdsColumn stk{stack, (uint32_t)stride};
    while (trav_visit<true>(m, qx, qy, T, stk)) {
    }
    return T.best;
}

// distance of q to the segment stored in `slot` (seed of a query: temporal hint)
__device__ __forceinline__ Closest slot_candidate(const DevMesh &m, int32_t slot, float qx, float qy)
{
    const float *nd = reinterpret_cast<const float *>(m.nodes + (WOST_NODE_FLOATS / 4) * (size_t)(m.first_leaf + (slot >> 2))) + (slot & 3);
    return Closest{obb_d2(nd[0], nd[4], nd[8], nd[12], nd[16], 0.0f, qx, qy), slot};
}

// The closest segment to (qx, qy) -- the same point in all 64 lanes -- by a scan of every segment, the lanes sharing the leaf
// level of the tree: the same distances as a leaf visit, the lowest ORIGINAL index among equal ones.  For the walkers that
// strayed so far from the mesh (DevMesh::huge2) that all its segments lie within the rounding of one another: the descent with
// its relative slack would open every box for them, one lane and one node at a time (two seconds for the 128 steps of one
// leaked walker on the 61 000 segments of fille).  Returns the same answer in every lane.
__device__ __forceinline__ Closest closest_point_wave(const DevMesh &m, float qx, float qy)
{
    const int lane = threadIdx.x & 63;
    float bd = WOST_INF;
    int32_t bs = -1, bo = WOST_FAR_INDEX;
    for (int i = lane; i < m.n_scan; i += 256) {
        // four records per lane and trip, their loads in flight together
        float4 b[4];
        float h[4];
        int2 id[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b[j] = m.scanBox[i + 64 * j];
            h[j] = m.scanHl[i + 64 * j];
            id[j] = m.scanId[i + 64 * j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = obb_d2(b[j].x, b[j].y, b[j].z, b[j].w, h[j], 0.0f, qx, qy);
            if (id[j].y != WOST_FAR_INDEX && (d < bd || (d == bd && id[j].y < bo))) {       // (padding records never win)
                bd = d; bs = id[j].x; bo = id[j].y;
            }
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

// Brute-force variant for tiny meshes (wave-uniform loop over the flat records, which the
// compiler turns into scalar loads).  Returns the ORIGINAL index in .slot.
__device__ __forceinline__ Closest closest_point_flat(const DevMesh &m, float qx, float qy)
{
    Closest best{WOST_INF, -1};
    for (int i = 0; i < m.n_segs; ++i) {
        const DevFlatSeg s = m.flat[i];
        float d = seg_d2(s, qx, qy);
        if (d < best.d2) {  // ascending i: strict < keeps the lowest index on ties
            best.d2 = d;
            best.slot = i;
        }
    }
    return best;
}

// ---- closest silhouette vertex (reference call site integrator.cu:189) ------------------
// Distance to the nearest vertex of the mesh that is a silhouette as seen from q, limited
// to vertices within rmax; +inf when there is none.  Test = FCPW isSilhouetteVertex.
__device__ __forceinline__ float closest_silhouette_flat(const DevMesh &m, float qx, float qy, float rmax)
{
#ifdef WOST_EXP_NO_SIL
    return WOST_INF;      // developer experiment: what the silhouette loop costs (a convex box seen from inside has none)
#endif
    float best2 = rmax * rmax;
    bool found = false;
    if (m.n_sil == 4) {
        // The Neumann boundary of every shipped scene is a four-vertex box.  The loop below fetches one vertex per trip, then its two
        // segments' records, with scalar loads each trip waits for; it costs 3-4 % of config 2 (a build without it: EXPERIMENTS 16)
        // although nearly every test ends with "farther than R_D" or "not a silhouette".  Here the four vertices and the normals of
        // their segments come with TWO loads that wait for nothing, and the loop's body runs on them unrolled: the same tests in the
        // same order on the same numbers, so the same result.
#ifdef WOST_EXP_SIL_FAST_ALWAYS
        return WOST_INF;      // developer experiment: the cost of everything below
#endif
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const DevSilVertex sv = m.sil[v];
            const float4 nn = m.silN[v];
            if (sv.prev < 0 && sv.next < 0) continue;
            const float vx = qx - sv.x, vy = qy - sv.y;
            const float d2 = dot2(vx, vy, vx, vy);
            if (d2 > best2) continue;
            bool is_sil = (sv.prev < 0 || sv.next < 0);
            if (!is_sil) {
                const float d = sqrtf(d2);
                if (d <= WOST_SIL_PRECISION) {
                    const float det = cross2(nn.x, nn.y, nn.z, nn.w);
                    is_sil = (-det > WOST_SIL_PRECISION);
                } else {
                    const float ux = vx / d, uy = vy / d;
                    const float dot0 = dot2(ux, uy, nn.x, nn.y);
                    const float dot1 = dot2(ux, uy, nn.z, nn.w);
                    if (fabsf(dot0) <= WOST_SIL_PRECISION || fabsf(dot1) <= WOST_SIL_PRECISION) is_sil = false;
                    else is_sil = (dot0 * dot1 < 0.0f);
                }
            }
            if (is_sil && (d2 < best2 || !found)) {
                best2 = d2;
                found = true;
            }
        }
        return found ? sqrtf(best2) : WOST_INF;
    }
    for (int v = 0; v < m.n_sil; ++v) {
        const DevSilVertex sv = m.sil[v];
        if (sv.prev < 0 && sv.next < 0) continue;  // vertex without segments
        float vx = qx - sv.x, vy = qy - sv.y;
        float d2 = dot2(vx, vy, vx, vy);
        if (d2 > best2) continue;
        bool is_sil = (sv.prev < 0 || sv.next < 0);
        if (!is_sil) {
            const DevFlatSeg s0 = m.flat[sv.prev], s1 = m.flat[sv.next];
            float d = sqrtf(d2);
            if (d <= WOST_SIL_PRECISION) {
                float det = cross2(s0.nx, s0.ny, s1.nx, s1.ny);
                is_sil = (-det > WOST_SIL_PRECISION);
            } else {
                float ux = vx / d, uy = vy / d;
                float dot0 = dot2(ux, uy, s0.nx, s0.ny);
                float dot1 = dot2(ux, uy, s1.nx, s1.ny);
                if (fabsf(dot0) <= WOST_SIL_PRECISION || fabsf(dot1) <= WOST_SIL_PRECISION) is_sil = false;
                else is_sil = (dot0 * dot1 < 0.0f);
            }
        }
        if (is_sil && (d2 < best2 || !found)) {
            best2 = d2;
            found = true;
        }
    }
    return found ? sqrtf(best2) : WOST_INF;
}

// ---- ray / segment (reference call sites integrator.cu:385-390,500) ----------------------
// the comparisons of seg_ray without its division: does the ray o + t d, t in [0, tmax], cross the segment a + s e, s in [0, 1]?
__device__ __forceinline__ bool seg_ray_hits(float ax, float ay, float ex, float ey, float ox, float oy, float dx, float dy, float tmax, float &uv,
                                             float &dv)
{
    float ux = ax - ox, uy = ay - oy;
    dv = cross2(dx, dy, ex, ey);
    if (dv == 0.0f) return false;
    float ud = cross2(ux, uy, dx, dy);
    uv = cross2(ux, uy, ex, ey);
    float adv = fabsf(dv);
    float sgn = (dv < 0.0f) ? -1.0f : 1.0f;
    float ud_s = ud * sgn, uv_s = uv * sgn;
    if (ud_s < 0.0f || ud_s > adv) return false;
    if (uv_s < 0.0f || uv_s > tmax * adv) return false;
    return true;
}

__device__ __forceinline__ bool seg_ray(const DevFlatSeg &s, float ox, float oy, float dx, float dy, float tmax,
                                        float &t)
{
    float uv, dv;
    if (!seg_ray_hits(s.ax, s.ay, s.ex, s.ey, ox, oy, dx, dy, tmax, uv, dv)) return false;
    t = uv / dv;
    return true;
}

// The four-segment boundary of every shipped scene (a box): the loops below unrolled -- the four records fetched with loads that
// do not wait for one another, the same tests in the same order, the division only for a segment that is hit.
template <bool ANY_HIT>
__device__ __forceinline__ bool ray_box_flat(const DevMesh &m, float ox, float oy, float dx, float dy, float tmax, float &t_out, int &idx_out)
{
    float4 a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float4 *>(&m.flat[i]);      // ax ay ex ey
    bool hit = false;
    float bt = WOST_INF;
    int bi = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float uv, dv;
        if (seg_ray_hits(a[i].x, a[i].y, a[i].z, a[i].w, ox, oy, dx, dy, tmax, uv, dv)) {
            if (ANY_HIT) {
                hit = true;
            } else {
                const float t = uv / dv;
                if (!hit || t < bt) {
                    bt = t;
                    bi = i;
                    hit = true;
                }
            }
        }
    }
    t_out = bt;
    idx_out = bi;
    return hit;
}

__device__ __forceinline__ bool ray_closest_flat(const DevMesh &m, float ox, float oy, float dx, float dy, float tmax,
                                                 float &t_out, int &idx_out)
{
    if (m.n_segs == 4) return ray_box_flat<false>(m, ox, oy, dx, dy, tmax, t_out, idx_out);
    bool hit = false;
    float bt = WOST_INF;
    int bi = -1;
    for (int i = 0; i < m.n_segs; ++i) {
        const DevFlatSeg s = m.flat[i];
        float t;
        if (seg_ray(s, ox, oy, dx, dy, tmax, t)) {
            if (!hit || t < bt) {
                bt = t;
                bi = i;
                hit = true;
            }
        }
    }
    t_out = bt;
    idx_out = bi;
    return hit;
}

__device__ __forceinline__ bool ray_any_flat(const DevMesh &m, float ox, float oy, float dx, float dy, float tmax)
{
    bool hit = false;
    if (m.n_segs == 4) {
        float t;
        int i;
        return ray_box_flat<true>(m, ox, oy, dx, dy, tmax, t, i);
    }
    for (int i = 0; i < m.n_segs; ++i) {
        const DevFlatSeg s = m.flat[i];
        float t;
        hit = hit || seg_ray(s, ox, oy, dx, dy, tmax, t);
    }
    return hit;
}

// ---- primitive sampling in a ball (reference call site integrator.cu:349-354) ------------
__device__ __forceinline__ int sample_in_sphere_flat(const DevMesh &m, float qx, float qy, float R, float u,
                                                     float &pdf)
{
    float R2 = R * R;
    float total = 0.0f;
    for (int i = 0; i < m.n_segs; ++i) {
        const DevFlatSeg s = m.flat[i];
        float d2 = seg_d2(s, qx, qy);
        if (d2 <= R2 && s.len > 0.0f) total += s.len;
    }
    pdf = 0.0f;
    if (!(total > 0.0f)) return -1;
    float target = u * total;
    float cum = 0.0f;
    int last = -1;
    bool done = false;
    for (int i = 0; i < m.n_segs; ++i) {
        const DevFlatSeg s = m.flat[i];
        float d2 = seg_d2(s, qx, qy);
        if (!done && d2 <= R2 && s.len > 0.0f) {
            cum += s.len;
            last = i;
            if (target < cum) done = true;
        }
    }
    float len = m.flat[last].len;
    pdf = (len / total) / len;
    return last;
}

// The same selection for boundary meshes too large to walk twice per step.  The probabilities are defined over
// the segments in ORIGINAL index order (inverse CDF), so the LBVH's Morton order is of no use; instead runs of
// consecutive indices carry boxes (level l: runs of 4^(l+1)), and an index-ordered sweep skips every aligned run
// whose box lies beyond the ball -- coarsest first.  The segments that are tested, their order and the float sums
// are those of the flat loop: identical result.
__device__ __forceinline__ float obox_d2(float4 b, float qx, float qy)
{
    const float dx = fmaxf(fmaxf(b.x - qx, qx - b.z), 0.0f), dy = fmaxf(fmaxf(b.y - qy, qy - b.w), 0.0f);
    return __builtin_fmaf(dx, dx, dy * dy);
}

// largest squared distance from q to a point of the box
__device__ __forceinline__ float obox_far_d2(float4 b, float qx, float qy)
{
    const float dx = fmaxf(fabsf(qx - b.x), fabsf(qx - b.z)), dy = fmaxf(fabsf(qy - b.y), fabsf(qy - b.w));
    return __builtin_fmaf(dx, dx, dy * dy);
}

// Calls, in ascending order of i (a multiple of four), f(i) for every group of four segments whose run boxes all touch
// the ball and g(i) for the groups of runs that lie INSIDE it (every point of the padded box closer than R by a
// relative margin, so the flat loop's test d2 <= R2 holds for each of them without being evaluated); both return
// false to stop.
template <class F, class G>
__device__ __forceinline__ void sweep_in_sphere(const DevMesh &m, float qx, float qy, float R2, F f, G g)
{
    // the boxes are padded by a fraction of the mesh extent, but a walker can be far outside the mesh (open boundaries:
    // |q| and R of 10^7 mesh units occur), where the rounding of both distances grows with |q|: the skip test is slack
    // by a relative 10^-4 on top (a run that is tested needlessly costs four exact tests; one that is skipped wrongly
    // changes the result)
    const float R2s = R2 * 1.000030517578125f, R2i = R2 * 0.9999f;     // skip: 1 + 2^-15 (a loose slack tests whole far meshes)
    int i = 0;
    while (i < m.n_segs) {
        // the longest aligned run starting at i that is decided as a whole, coarsest first
        int skip = 0, inside = 0;
        for (int l = m.obox_levels - 1; l >= 0 && (skip | inside) == 0; --l) {
            const int run = 4 << (2 * l);
            if ((i & (run - 1)) == 0) {
                const float4 b = m.obox[m.obox_off[l] + i / run];
                if (obox_d2(b, qx, qy) > R2s) skip = run;
                else if (obox_far_d2(b, qx, qy) <= R2i) inside = run;
            }
        }
        if (skip) {
            i += skip;
            continue;
        }
        if (inside) {
            const int end = min(i + inside, m.n_segs);
            for (; i < end; i += 4)
                if (!g(i)) return;
            continue;
        }
        if (!f(i)) return;
        i += 4;
    }
}

// the four segments from i on, in order: take(index, length) for those the flat loop accepts (padding has length 0)
template <bool TEST, class T>
__device__ __forceinline__ bool sample_group(const DevMesh &m, int i, float qx, float qy, float R2, T take)
{
    const float4 l = *reinterpret_cast<const float4 *>(m.lens + i);
    if (TEST) {
        const float4 b0 = m.sampBox[i], b1 = m.sampBox[i + 1], b2 = m.sampBox[i + 2], b3 = m.sampBox[i + 3];
        const float4 h = *reinterpret_cast<const float4 *>(m.sampHl + i);
        if (obb_d2(b0.x, b0.y, b0.z, b0.w, h.x, 0.0f, qx, qy) <= R2 && l.x > 0.0f && !take(i, l.x)) return false;
        if (obb_d2(b1.x, b1.y, b1.z, b1.w, h.y, 0.0f, qx, qy) <= R2 && l.y > 0.0f && !take(i + 1, l.y)) return false;
        if (obb_d2(b2.x, b2.y, b2.z, b2.w, h.z, 0.0f, qx, qy) <= R2 && l.z > 0.0f && !take(i + 2, l.z)) return false;
        if (obb_d2(b3.x, b3.y, b3.z, b3.w, h.w, 0.0f, qx, qy) <= R2 && l.w > 0.0f && !take(i + 3, l.w)) return false;
    } else {
        if (l.x > 0.0f && !take(i, l.x)) return false;
        if (l.y > 0.0f && !take(i + 1, l.y)) return false;
        if (l.z > 0.0f && !take(i + 2, l.z)) return false;
        if (l.w > 0.0f && !take(i + 3, l.w)) return false;
    }
    return true;
}

__device__ __forceinline__ int sample_in_sphere_tree(const DevMesh &m, float qx, float qy, float R, float u, float &pdf)
{
    const float R2 = R * R;
    float total = 0.0f;
    auto add = [&](int, float len) {
        total += len;
        return true;
    };
    sweep_in_sphere(
        m, qx, qy, R2, [&](int i) { return sample_group<true>(m, i, qx, qy, R2, add); },
        [&](int i) { return sample_group<false>(m, i, qx, qy, R2, add); });
    pdf = 0.0f;
    if (!(total > 0.0f)) return -1;
    const float target = u * total;
    float cum = 0.0f;
    int last = -1;
    auto pick = [&](int i, float len) {
        cum += len;
        last = i;
        return !(target < cum);
    };
    sweep_in_sphere(
        m, qx, qy, R2, [&](int i) { return sample_group<true>(m, i, qx, qy, R2, pick); },
        [&](int i) { return sample_group<false>(m, i, qx, qy, R2, pick); });
    const float len = m.lens[last];
    pdf = (len / total) / len;
    return last;
}

// ---- the same queries on the wide LBVH, for boundary meshes too large for flat loops -------
// Results are identical to the flat loops (layout-independent definitions); the tree only
// prunes.  `stk` is any free traversal-stack column of the lane / walker.

// silhouette predicate of one vertex (FCPW isSilhouetteVertex, flipNormalOrientation = false)
__device__ __forceinline__ bool vertex_is_silhouette(const DevMesh &m, const DevSilVertex &sv, float vx, float vy, float d2)
{
    if (sv.prev < 0 || sv.next < 0) return true;
    const DevFlatSeg s0 = m.flat[sv.prev], s1 = m.flat[sv.next];
    const float d = sqrtf(d2);
    if (d <= WOST_SIL_PRECISION) {
        const float det = cross2(s0.nx, s0.ny, s1.nx, s1.ny);
        return -det > WOST_SIL_PRECISION;
    }
    const float ux = vx / d, uy = vy / d;
    const float dot0 = dot2(ux, uy, s0.nx, s0.ny);
    const float dot1 = dot2(ux, uy, s1.nx, s1.ny);
    if (fabsf(dot0) <= WOST_SIL_PRECISION || fabsf(dot1) <= WOST_SIL_PRECISION) return false;
    return dot0 * dot1 < 0.0f;
}

// SNCH test (Sawhney et al. 2023, "Walk on Stars", spatialized normal cone hierarchy): can the
// subtree whose normals lie in the cone (axis, half angle) and whose vertices lie in the disc
// (c, rad) contain a silhouette seen from q?  A silhouette needs a normal perpendicular to a
// view direction, i.e. pi/2 inside [angle(axis, view) -/+ (half + view half angle)].
// Trig-free and conservative: |cos(angle)| <= sin(half + viewhalf), slack added.
__device__ __forceinline__ bool cone_may_hold_silhouette(float ax, float ay, float ch, float sh, float rad, float cx, float cy,
                                                        float qx, float qy)
{
    if (ch <= 0.0f) return true;                 // marked "cannot prune"
    const float wx = cx - qx, wy = cy - qy;
    const float l2 = dot2(wx, wy, wx, wy);
    if (l2 <= rad * rad * 1.0001f) return true;  // q inside the bounding disc: no view cone
    const float inv_l = 1.0f / sqrtf(l2);
    const float sv = fminf(rad * inv_l, 1.0f);                 // sin(view half angle)
    const float cv = sqrtf(fmaxf(1.0f - sv * sv, 0.0f));
    const float cos_sum = ch * cv - sh * sv;                   // cos(half + viewhalf)
    if (cos_sum <= 1e-3f) return true;                         // sum >= ~90 degrees
    const float sin_sum = sh * cv + ch * sv;
    const float c = (ax * wx + ay * wy) * inv_l;               // cos(angle(axis, view axis))
    return fabsf(c) <= sin_sum + 1e-3f;
}

// Near-first like the closest-point descent (Trav, trav_pop: the keys carry the box distance, so entries that a
// silhouette found meanwhile has overtaken are culled at the pop without a visit).  T.best.d2 is the pruning bound --
// the best squared distance with a relative slack, because box and vertex distances come from different formulas
// and a walker may stand far outside the scene -- the exact minimum is kept apart: the flat loop's answer.
template <class STK>
__device__ __forceinline__ float closest_silhouette_tree(const DevMesh &m, float qx, float qy, float rmax, const STK &stk)
{
    constexpr float kSlack = 1.000030517578125f;   // 1 + 2^-15: the rounding is a few 10^-7; 10^-4 opens every box for a far walker
    float best2 = rmax * rmax;
    bool found = false;
    Trav T = trav_begin(Closest{best2 * kSlack, -1});
    for (;;) {
        const uint32_t g = level_first(T.level) + (uint32_t)T.pos;
        const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 4u * WOST_NODE_FLOATS));
        const float4 CX = nd[0], CY = nd[1], UX = nd[2], UY = nd[3], HL = nd[4], HW = nd[5];
        bool more;
        if (T.level == m.levels) {
            const int slot0 = 4 * T.pos;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int2 vv = m.segVerts[slot0 + j];
                if (vv.x < 0) continue;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const DevSilVertex sv = m.sil[e ? vv.y : vv.x];
                    const float vx = qx - sv.x, vy = qy - sv.y;
                    const float d2 = dot2(vx, vy, vx, vy);
                    if (d2 > best2) continue;
                    if (vertex_is_silhouette(m, sv, vx, vy, d2) && (d2 < best2 || !found)) {
                        best2 = d2;
                        found = true;
                    }
                }
            }
            T.best.d2 = best2 * kSlack;
            more = trav_pop(T, stk);
        } else {
            const float4 *cn = m.cones + 5 * (size_t)g;
            const float4 AX = cn[0], AY = cn[1], CH = cn[2], SH = cn[3], RD = cn[4];
            const float bd = T.best.d2;
            const float d0 = obb_d2(CX.x, CY.x, UX.x, UY.x, HL.x, HW.x, qx, qy), d1 = obb_d2(CX.y, CY.y, UX.y, UY.y, HL.y, HW.y, qx, qy);
            const float d2 = obb_d2(CX.z, CY.z, UX.z, UY.z, HL.z, HW.z, qx, qy), d3 = obb_d2(CX.w, CY.w, UX.w, UY.w, HL.w, HW.w, qx, qy);
            const bool c0 = d0 <= bd && cone_may_hold_silhouette(AX.x, AY.x, CH.x, SH.x, RD.x, CX.x, CY.x, qx, qy);
            const bool c1 = d1 <= bd && cone_may_hold_silhouette(AX.y, AY.y, CH.y, SH.y, RD.y, CX.y, CY.y, qx, qy);
            const bool c2 = d2 <= bd && cone_may_hold_silhouette(AX.z, AY.z, CH.z, SH.z, RD.z, CX.z, CY.z, qx, qy);
            const bool c3 = d3 <= bd && cone_may_hold_silhouette(AX.w, AY.w, CH.w, SH.w, RD.w, CX.w, CY.w, qx, qy);
            const uint32_t tag = (uint32_t)(T.level + 1) << 2;
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
        if (!more) break;
    }
    return found ? sqrtf(best2) : WOST_INF;
}

// entry distance of the ray o + t d, t in [0, tmax], into an oriented box; +inf when it misses.
// Pruning only: slabs are widened by a relative epsilon, so a box is never missed.
__device__ __forceinline__ float ray_obb_entry(float cx, float cy, float ux, float uy, float hl, float hw, float ox, float oy,
                                               float dx, float dy, float tmax)
{
    const float wx = ox - cx, wy = oy - cy;
    const float ou = wx * ux + wy * uy, ov = wy * ux - wx * uy;
    const float du = dx * ux + dy * uy, dv = dy * ux - dx * uy;
    // (the second term: the record's centre and axis are the ROUNDED image of a segment whose exact test runs on its end points;
    // on a fine mesh far from the origin -- segments of 0.02 at |x| = 100 -- that image lies several 10^-6 off the segment, more
    // than the relative widening: a ray that starts on the boundary was missed by the box and hit by the flat loop)
    const float slack = 1e-4f * (fabsf(ou) + fabsf(ov) + hl + hw) + 2.4e-7f * (fabsf(cx) + fabsf(cy)) + 1e-30f;
    const float hlp = hl + slack, hwp = hw + slack;
    float t0 = 0.0f, t1 = tmax;
    // slab along u
    if (fabsf(du) > 1e-20f) {
        const float inv = 1.0f / du;
        float a = (-hlp - ou) * inv, b = (hlp - ou) * inv;
        t0 = fmaxf(t0, fminf(a, b)); t1 = fminf(t1, fmaxf(a, b));
    } else if (fabsf(ou) > hlp) return WOST_INF;
    if (fabsf(dv) > 1e-20f) {
        const float inv = 1.0f / dv;
        float a = (-hwp - ov) * inv, b = (hwp - ov) * inv;
        t0 = fmaxf(t0, fminf(a, b)); t1 = fminf(t1, fmaxf(a, b));
    } else if (fabsf(ov) > hwp) return WOST_INF;
    const float tol = 1e-4f * (fabsf(t0) + fabsf(t1)) + 1e-6f;
    return (t0 <= t1 + tol) ? fmaxf(t0 - tol, 0.0f) : WOST_INF;
}

// closest hit (ANY_HIT = false) or any hit (true); ties in t go to the lowest ORIGINAL index.
// Near-first like the closest-point descent: the key of a child is where the ray enters its box (a lower bound, see
// ray_obb_entry), T.best.d2 carries the limit -- tmax, then the best hit -- so entries beyond a hit found meanwhile
// are culled at the pop.  A leaf measures its four segments as boxes of no width first and runs the exact test
// (the flat loop's seg_ray on the flat record) only for those the ray can reach within the limit.
template <bool ANY_HIT, class STK>
__device__ __forceinline__ bool ray_tree(const DevMesh &m, float ox, float oy, float dx, float dy, float tmax, float &t_out,
                                         int &idx_out, const STK &stk)
{
    bool hit = false;
    float bt = WOST_INF;
    int bi = -1;
    Trav T = trav_begin(Closest{tmax, -1});
    for (;;) {
        const uint32_t g = level_first(T.level) + (uint32_t)T.pos;
        const float4 *nd = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(m.nodes) + __umul24(g, 4u * WOST_NODE_FLOATS));
        const float4 CX = nd[0], CY = nd[1], UX = nd[2], UY = nd[3], HL = nd[4], HW = nd[5];
        const float lim = T.best.d2;
        const float e0 = CX.x >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.x, CY.x, UX.x, UY.x, HL.x, HW.x, ox, oy, dx, dy, lim);
        const float e1 = CX.y >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.y, CY.y, UX.y, UY.y, HL.y, HW.y, ox, oy, dx, dy, lim);
        const float e2 = CX.z >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.z, CY.z, UX.z, UY.z, HL.z, HW.z, ox, oy, dx, dy, lim);
        const float e3 = CX.w >= 1.0e17f ? WOST_INF : ray_obb_entry(CX.w, CY.w, UX.w, UY.w, HL.w, HW.w, ox, oy, dx, dy, lim);
        bool more;
        if (T.level == m.levels) {
            const int slot0 = 4 * T.pos;
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                const float ej = j == 0 ? e0 : j == 1 ? e1 : j == 2 ? e2 : e3;
                if (!(ej <= T.best.d2)) continue;
                const int o = m.segOrig[slot0 + j];
                if (o == WOST_FAR_INDEX) continue;
                const DevFlatSeg s = m.flat[o];
                float t;
                if (seg_ray(s, ox, oy, dx, dy, tmax, t)) {
                    if (!hit || t < bt || (t == bt && o < bi)) {
                        bt = t;
                        bi = o;
                        hit = true;
                        T.best.d2 = fminf(bt, tmax);
                    }
                }
            }
            if (ANY_HIT && hit) break;
            more = trav_pop(T, stk);
        } else {
            const uint32_t tag = (uint32_t)(T.level + 1) << 2;
            uint32_t k0 = (e0 <= lim) ? ((__float_as_uint(e0) & ~0x3Fu) | tag | 0u) : 0xffffffffu;
            uint32_t k1 = (e1 <= lim) ? ((__float_as_uint(e1) & ~0x3Fu) | tag | 1u) : 0xffffffffu;
            uint32_t k2 = (e2 <= lim) ? ((__float_as_uint(e2) & ~0x3Fu) | tag | 2u) : 0xffffffffu;
            uint32_t k3 = (e3 <= lim) ? ((__float_as_uint(e3) & ~0x3Fu) | tag | 3u) : 0xffffffffu;
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
        if (!more) break;
    }
    t_out = bt;
    idx_out = bi;
    return hit;
}

// a boundary mesh is walked with the flat wave-uniform loops up to this many segments
#define WOST_FLAT_MAX 64

// TREE is a compile-time choice (the host picks the kernel instantiation from the mesh size):
// kernels for small boundary meshes carry no traversal code and keep their register budget.
template <bool TREE, class STK>
__device__ __forceinline__ float closest_silhouette(const DevMesh &m, float qx, float qy, float rmax, const STK &stk)
{
    if (!TREE) return closest_silhouette_flat(m, qx, qy, rmax);
    return closest_silhouette_tree(m, qx, qy, rmax, stk);
}

template <bool TREE, class STK>
__device__ __forceinline__ bool ray_closest(const DevMesh &m, float ox, float oy, float dx, float dy, float tmax, float &t_out,
                                            int &idx_out, const STK &stk)
{
    if (!TREE) return ray_closest_flat(m, ox, oy, dx, dy, tmax, t_out, idx_out);
    return ray_tree<false>(m, ox, oy, dx, dy, tmax, t_out, idx_out, stk);
}

template <bool TREE, class STK>
__device__ __forceinline__ bool ray_any(const DevMesh &m, float ox, float oy, float dx, float dy, float tmax, const STK &stk)
{
    if (!TREE) return ray_any_flat(m, ox, oy, dx, dy, tmax);
    float t;
    int i;
    return ray_tree<true>(m, ox, oy, dx, dy, tmax, t, i, stk);
}

// ---- surface colour (reference integrator/common.h:242-260, functors.h:60-64) ------------
// col12 = left(i0) left(i1) right(i0) right(i1)
__device__ __forceinline__ void surface_color(const float *col12, int side, float uv, float &r, float &g, float &b)
{
    const float *c = col12 + ((side >= 0) ? 0 : 6);
    r = c[0] * (1 - uv) + c[3] * uv;
    g = c[1] * (1 - uv) + c[4] * uv;
    b = c[2] * (1 - uv) + c[5] * uv;
}

}  // namespace wost
