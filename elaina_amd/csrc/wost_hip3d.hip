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
#include "wost_device3.h"
#include "wost_internal3.h"

namespace wost {

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

// B: the star radius, the source and Neumann samples, the direction of the step -- in three parts around the two rays those
// samples ask for (the source sample's line to the boundary, the boundary sample's shadow ray), so that the kernel can answer
// them by the wave like the walker's own ray (step3_b answers them on the spot).  The draws of a lane keep their order.
struct Step3B {
    float R_B;
    V3 sdir;                 // source sample: direction, its density, the boundary factor
    float dir_pdf, salpha;
    int oi;                  // boundary sample: triangle, density, the point, its distance, the shadow ray (o, rd, cd)
    float pdf, r, cd;
    V3 sp, o, rd;
};

// B0: the radius; the direction of the source sample.  true = no boundary at all (the walk ends)
template <bool SOURCE>
__device__ __forceinline__ bool step3_b0(const Walk3Params &P, Lane3 &L, float R_D, float R_N, Step3B &B)
{
    float R_B = fmaxf(WOST_R_B_FLOOR, fminf(R_D, R_N));
    R_B *= WOST_R_B_SHRINK;
    B.R_B = R_B;
    if (isinf(R_B)) return true;
    // ---- sampleSource (reference integrator/uniform/integrator.cu:235-316, DIM == 3) ----
    if (SOURCE) {
        B.salpha = 1.0f;
        const float u1 = pcg_next_float(L.rng), u2 = pcg_next_float(L.rng);
        float c, s;
        sincos_2pi(u2, c, s);
        if (L.on_n) {
            const float z = u1, r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
            B.sdir = frame_to_world(L.nn, r * c, r * s, z);
            B.dir_pdf = 1.0f / WOST_2PI;
            B.salpha = 0.5f;
        } else {
            const float z = 1 - 2 * u1, r = sqrtf(1 - z * z);
            B.sdir = v3(r * c, r * s, z);
            B.dir_pdf = 1.0f / WOST_4PI;
        }
    }
    return false;
}
// the source sample's line: from p + eps sdir along sdir, at most R_B long (asked for when the scene has a Neumann mesh)
__device__ __forceinline__ V3 step3_source_origin(const Walk3Params &P, const Lane3 &L, const Step3B &B)
{
    const float eps = P.st.eps;
    return v3(L.p.x + eps * B.sdir.x, L.p.y + eps * B.sdir.y, L.p.z + eps * B.sdir.z);
}

// B1: the source sample with its line answered (hit, t); the boundary sample.  true = the shadow ray (B.o, B.rd, B.cd - eps) is asked for
template <bool EMISSIVE, bool SOURCE, bool NTREE>
__device__ __forceinline__ bool step3_b1(const Walk3Params &P, Lane3 &L, Step3B &B, bool src_hit, float src_t)
{
    const bool has_n = P.nm.n_tris > 0;
    const float eps = P.st.eps, R_B = B.R_B;
    const V3 p = L.p;
    const float thp = L.thp;
    float (&sol)[3] = L.sol;
    Pcg &rng = L.rng;
    PROF3_T0();
    if (SOURCE) {
        // how far the straight line stays inside the star-shaped region (:279-292)
        float dist = R_B;
        if (has_n && src_hit) dist = fminf(src_t, dist);
        const V3 sdir = B.sdir;
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
            const float c1 = (1.0f / WOST_4PI) / (r * r), c2 = B.dir_pdf / (r * r);     // conditionalSampleSpherePDF<3>
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float col = thp * f[k] * norm * c1 / c2 / B.salpha;
                sol[k] = col + sol[k];
            }
        }
    }
    PROF3(1);
    // ---- sampleNeumann: three draws whether or not the boundary emits ----
    bool shadow = false;
    if (has_n) {
        const float u0 = pcg_next_float(rng), u1 = pcg_next_float(rng), u2 = pcg_next_float(rng);
        if (EMISSIVE) {
            float pdf;
            const int oi = (NTREE && P.nm.obox_levels > 0) ? sample_in_sphere3_tree(P.nm, p, R_B, u0, pdf) : sample_in_sphere3_flat(P.nm, p, R_B, u0, pdf);
            if (oi != -1 && pdf > 0) {
                const DevTri &S = P.nm.flat[oi];
                const V3 s0 = ld3(S.p0), s1 = ld3(S.p1), s2 = ld3(S.p2);
                const float su = sqrtf(u1), b1 = u2 * su, b0 = 1.0f - su, b2 = 1.0f - b0 - b1;
                const V3 sp = v3((s0.x * b0 + s1.x * b1) + s2.x * b2, (s0.y * b0 + s1.y * b1) + s2.y * b2,
                                 (s0.z * b0 + s1.z * b1) + s2.z * b2);
                const V3 rv = sp - p;
                const float r = sqrtf(dot3(rv, rv));
                if (r < R_B && r > 0) {
                    V3 o = p;
                    if (L.on_n) o = v3(p.x + eps * L.nn.x, p.y + eps * L.nn.y, p.z + eps * L.nn.z);
                    V3 rd = sp - o;
                    const float cd = sqrtf(dot3(rd, rd));
                    if (cd > 0) { rd.x /= cd; rd.y /= cd; rd.z /= cd; }
                    B.oi = oi; B.pdf = pdf; B.r = r; B.cd = cd; B.sp = sp; B.o = o; B.rd = rd;
                    shadow = true;
                }
            }
        }
    }
    return shadow;
}

// B2: the boundary sample when its shadow ray found nothing (`lit`); the direction of the step.  The walker's ray starts at
// `cur` along `dir` and is at most B.R_B long.
template <bool EMISSIVE>
__device__ __forceinline__ void step3_b2(const Walk3Params &P, Lane3 &L, const Step3B &B, bool lit, V3 &dir_out, V3 &cur_out)
{
    const float eps = P.st.eps;
    const V3 p = L.p;
    const bool on_n = L.on_n;
    const V3 nn = L.nn;
    float (&sol)[3] = L.sol;
    PROF3_T0();
    if (EMISSIVE && lit) {
        const DevTri &S = P.nm.flat[B.oi];
        const V3 s0 = ld3(S.p0), s1 = ld3(S.p1), s2 = ld3(S.p2);
        int side = tri_side(s0, ld3(S.nraw), p);
        float uu, vv;
        tri_uv(s0, s1 - s0, s2 - s0, B.sp, uu, vv);
        if (on_n) {
            const float dn = dot3(ld3(S.n), nn);
            side = (0.0f < dn) - (dn < 0.0f);
        }
        if (side != 0) {
            float col[3];
            const int32_t *tv = P.nm.flatVerts + 3 * (size_t)B.oi;
            surface_color3(P.nm.colors, tv[0], tv[1], tv[2], side, uu, vv, col);
            const float alpha = on_n ? 0.5f : 1.0f;
            const float G = (1.0f / B.r - 1.0f / B.R_B) / WOST_4PI;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                col[k] *= P.st.neumann_intensity;
                col[k] *= L.thp * G / alpha / B.pdf;
                sol[k] = -col[k] + sol[k];
            }
        }
    }
    PROF3(2);
    // ---- oneStepWalk ----
    V3 dir, cur = p;
    {
        const float u1 = pcg_next_float(L.rng), u2 = pcg_next_float(L.rng);
        float c, s;
        sincos_2pi(u2, c, s);
        if (on_n) {
            const float z = u1, r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
            dir = frame_to_world(nn, r * c, r * s, z);
            cur = v3(p.x + eps * nn.x, p.y + eps * nn.y, p.z + eps * nn.z);
        } else {
            const float z = 1 - 2 * u1, r = sqrtf(1 - z * z);
            dir = v3(r * c, r * s, z);
        }
    }
    dir_out = dir; cur_out = cur;
}

// B with both rays answered by the lane itself.  true = no boundary at all (the walk ends)
template <bool EMISSIVE, bool SOURCE, bool NTREE>
__device__ __forceinline__ bool step3_b(const Walk3Params &P, Lane3 &L, float R_D, float R_N, const LdsColumn &stk, float &R_B_out, V3 &dir_out, V3 &cur_out)
{
    Step3B B;
    if (step3_b0<SOURCE>(P, L, R_D, R_N, B)) return true;
    bool src_hit = false;
    float src_t = 0.0f;
    if (SOURCE && P.nm.n_tris > 0) {
        int hi;
        src_hit = ray_closest3<NTREE>(P.nm, step3_source_origin(P, L, B), B.sdir, B.R_B, src_t, hi, stk);
    }
    bool lit = step3_b1<EMISSIVE, SOURCE, NTREE>(P, L, B, src_hit, src_t);
    if (EMISSIVE && lit) lit = !ray_any3<NTREE>(P.nm, B.o, B.rd, B.cd - P.st.eps, stk);
    step3_b2<EMISSIVE>(P, L, B, lit, dir_out, cur_out);
    R_B_out = B.R_B;
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
                // part B around the rays of its samples, those answered by the wave as well (a lane's shadow ray through a tree
                // of its own was the longest wait of an emissive step)
                if ((EMISSIVE || SOURCE) && P.coop > 1) {
                    Step3B B;
                    B.sdir = B.o = B.rd = v3(0.0f, 0.0f, 0.0f);
                    B.R_B = B.cd = 0.0f;
                    if (mid) {
                        ended = step3_b0<SOURCE>(P, L, R_D, R_N, B);
                        go = !ended;
                    }
                    bool src_hit = false, lit = false;
                    float src_t = 0.0f;
                    if (SOURCE) {
                        int shi;
                        src_hit = ray_closest3_wave(P.nm, step3_source_origin(P, L, B), B.sdir, B.R_B, go, src_t, shi, W, stk, P.ray_slot_trigger);
                    }
                    if (go) lit = step3_b1<EMISSIVE, SOURCE, NTREE>(P, L, B, src_hit, src_t);
                    if (EMISSIVE) {
                        float st;
                        int shi;
                        const bool occluded = ray_closest3_wave(P.nm, B.o, B.rd, B.cd - P.st.eps, lit, st, shi, W, stk, P.ray_slot_trigger);
                        lit = lit && !occluded;
                    }
                    if (go) step3_b2<EMISSIVE>(P, L, B, lit, dir, cur);
                    R_B = B.R_B;
                } else if (mid) {
                    // (WOST3_COOP=1: the rays of the samples by the lane itself, as in rounds 3)
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

}  // namespace wost

using namespace wost;

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
        P.coop = ntree ? 2 : 0;          // 2: the rays of the source and boundary samples through the pools as well (1: by the lane)
        P.pool_cap = 512;
        P.ray_slot_trigger = 32;
        P.cp_slot_trigger = 64;
        if (const char *w = std::getenv("WOST3_CP_TRIGGER")) P.cp_slot_trigger = std::min(64, std::max(1, std::atoi(w)));
        if (const char *w = std::getenv("WOST3_COOP")) P.coop = ntree ? std::max(0, std::min(2, std::atoi(w))) : 0;
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

}  // extern "C"
