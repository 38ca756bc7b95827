// wost_hip.hip -- walk kernels and the C-ABI of libwost_hip.so (gfx950 / MI355X only).
//
// Design (DESIGN.md has the long form):
//   * one SoA walk-state queue; a queue slot is one PIXEL's walker, which carries its
//     PCG32 state, position, partial solution and the temporal hint of the last closest
//     segment.  A pixel's samples are strictly sequential in the reference (one RNG stream
//     per pixel, integrator/uniform/integrator.cu:71-77, one walk in flight per pixel), so a
//     slot restarts its next sample the moment the previous walk ends instead of idling
//     until every other walk of that sample is done -- same per-pixel arithmetic, no
//     starved late-depth launches.
//   * a launch ("round") advances every slot by up to `steps_per_round` walk steps held in
//     registers, then compacts the still-unfinished slots into the other queue with
//     ballot/popcount + one atomic per wave, and resolves finished pixels into the field.
//   * the per-lane LBVH traversal stack lives in LDS (one column per lane, bank = lane).
//   * with few samples per pixel the REFILL instantiation drains the queue in ONE launch; with many, and more walkers
//     than resident lanes, the first launch is PERSISTENT: lanes take whole pixels, longest expected chain first
//     (wost_order.h), until none is unread, and hand what they hold to a few rounds -- the remainders expected to be
//     long run to their end beside those rounds, four lanes to a walker (run_solve);
//     problems with a source term use the SOURCE instantiations (wost_walk.h).
//   * no managed memory, no CPU fallback.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "../../include/wost.h"
#include "lbvh.h"
#include "wost_build2.h"
#include "wost_order.h"
#include "wost_device.h"
#include "wost_internal.h"
#include "wost_walk.h"
#include "wost_quad.h"
#include "wost_coop.h"

namespace wost {

// ------------------------------------------------------------------------------------------
// queue layout
// ------------------------------------------------------------------------------------------
struct WalkQueue {
    uint32_t *pix;     // global pixel id
    float *x0, *y0;    // evaluation point of the pixel (start of every sample)
    float *px, *py;    // current walk position
    uint64_t *rng;     // PCG32 state (inc == 1 for every pixel: setSeed(.., seq 0))
    uint32_t *meta;    // sample[0:20) | depth[20:30) | onNeumann[30]
    float *nx, *ny;    // Neumann normal at the current point (valid when onNeumann)
    int32_t *hint;     // slot of the closest Dirichlet segment of the previous step
    float *thp;        // throughput (scalar: all three channels are always equal)
    float *sr, *sg, *sb;  // running solution of the pixel
    float *d0_d2;      // cached closest-point result of the evaluation point (depth 0 of
    int32_t *d0_slot;  //   every sample starts at the same point)
    float *est;        // written when a persistent launch hands a pixel over mid-way: walk steps the pixel is expected to need still
                       //   (not part of a walker's state: no kernel loads it back; the host orders the next launch by it)
};

static const int kQueueWords = 17 + 1;  // rng counts twice

// Atomics on one cache line are served one by one by its L2 channel (about 10 ns each: the eight
// counters of the 16 384 waves of a round cost over a millisecond), so the counters exist in
// kStatCopies copies 256 bytes apart, a block adds to the copy blockIdx % kStatCopies and the host
// sums the copies.
constexpr int kStatCopies = 64;
struct alignas(256) StatsDev {
    unsigned long long steps, started, absorbed, truncated, nhits, inner_visits, leaf_visits, trav_trips, step_trips, max_stack;
    unsigned long long sp_ge6, sp_ge10, sp_ge14;     // WOST_TRACK developer builds: visits that left at least that many stack entries
};
__device__ __forceinline__ StatsDev *my_stats(StatsDev *s) { return s + (blockIdx.x & (kStatCopies - 1)); }

struct RoundParams {
    DevMesh dm, nm;
    DevSettings st;
    DevProbe probe;
    DevSource src;
    WalkQueue in, out;
    const uint32_t *count_in;
    uint32_t *count_out;
    float *field;        // solution/spp written at field[(pix - field_base) * 3]
    int32_t field_base;
    StatsDev *stats;
    int32_t steps_per_round;
    int32_t stack_stride;  // = blockDim.x
    int32_t wait_weight;   // step phase runs when n_wait * wait_weight >= 8 * n_trav
    int32_t trav_burst;    // node visits per scheduling decision in the traversal phase
    int32_t lane_shift;    // one walker per 2^lane_shift lanes (0 = every lane)
    uint32_t *cursor;      // REFILL launches: next unread slot of the input queue
    const uint32_t *order; // the k-th walker of the launch is input slot order[k] (wost_order.h); nullptr = slot k
    int32_t reserve;       // REFILL launches: input slots a wave reserves per atomic on the cursor; 0 = exactly those it needs
    int32_t leave_dry;     // REFILL launches: 1 = a wave that finds the input queue dry finishes the steps in flight and hands its
                           //   walkers to the output queue (the host repacks them: rounds); 0 = it stays until its last pixel is done
    uint32_t *count_long;  // ... and counts the pixels it hands over that are expected to need long_steps walk steps or more
    float long_steps;
    // walk_quad_kernel, the launch of the long remainders: the first thin_count walkers sit four to a wave (sixteen to a workgroup)
    // in the first thin_blocks workgroups, the others sixteen to a wave behind them; thin_blocks == 0: lane_shift for all
    uint32_t thin_blocks, thin_count;
    // walkers that the plain kernel cannot serve exactly (a closest-point query that starts beyond dm.far2) leave the launch
    // at that point and are queued from the TOP of the output queue downwards: slot out_capacity - 1 - k, k from *count_far
    uint32_t *count_far;
    uint32_t out_capacity;
    // NEUMANN_TREE launches of walk_round_kernel: the Neumann-side tree queries of a step by the wave as a whole (wost_coop.h):
    // pool_cap tasks per pool and wave, pool_offset words into the block's LDS (behind the stack columns); 0 = per lane
    int32_t coop, pool_cap, pool_offset, ray_slot_trigger;
};

struct InitParams {
    DevMesh dm;
    DevSettings st;
    DevProbe probe;
    WalkQueue out;
    uint32_t *count_out;
    const uint8_t *mask;
    float *field;
    int32_t field_base;
    int32_t pixel_begin, pixel_end;
    int32_t shard_index, shard_count;
    int32_t tiles_x, tiles_y;
    int32_t stack_stride;
};

#define META_SAMPLE(m) ((m) & 0xfffffu)
#define META_DEPTH(m) (((m) >> 20) & 0x3ffu)
#define META_ONN(m) (((m) >> 30) & 1u)
#define META_PACK(s, d, n) ((uint32_t)(s) | ((uint32_t)(d) << 20) | ((uint32_t)(n) << 30))

// ------------------------------------------------------------------------------------------
// init: seed every owned pixel, cache its depth-0 closest point, compact into the queue
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void init_kernel(InitParams P)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int tile = t >> 6, i = t & 63;
    const int tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const int x = tx * 8 + (i & 7), y = ty * 8 + (i >> 3);
    bool owned = (ty < P.tiles_y) && (x < P.st.width) && (y < P.st.height);
    int pid = y * P.st.width + x;
    owned = owned && pid >= P.pixel_begin && pid < P.pixel_end && (tile % P.shard_count) == P.shard_index;
    bool active = owned && (P.mask == nullptr || P.mask[pid] != 0) && P.st.spp > 0;
    if (owned && !active) {
        // masked pixel: solution stays 0 (reference integrator.cu:92-95, :616-620)
        float *f = P.field + 3 * (size_t)(pid - P.field_base);
        float z = 0.0f / (float)P.st.spp;
        f[0] = z; f[1] = z; f[2] = z;
    }
    float x0 = 0, y0 = 0;
    Pcg rng{0, 1};
    Closest c0{WOST_INF, -1};
    if (active) {
        eval_point(P.probe, x, y, P.st.width, P.st.height, x0, y0);
        pcg_seed_pixel(rng, pid, P.st.width);
        if (P.dm.n_segs > 0) c0 = closest_point(P.dm, x0, y0, slot_candidate(P.dm, 0, x0, y0), stack, P.stack_stride);
    }
    const uint32_t s = block_push(active, P.count_out);
    if (active) {
        WalkQueue &q = P.out;
        q.pix[s] = pid;
        q.x0[s] = x0; q.y0[s] = y0;
        q.px[s] = x0; q.py[s] = y0;
        q.rng[s] = rng.state;
        q.meta[s] = META_PACK(0, 0, 0);
        q.nx[s] = 0.0f; q.ny[s] = 0.0f;
        q.hint[s] = c0.slot;
        q.thp[s] = 1.0f;
        q.sr[s] = 0.0f; q.sg[s] = 0.0f; q.sb[s] = 0.0f;
        q.d0_d2[s] = c0.d2;
        q.d0_slot[s] = c0.slot;
    }
}

// ------------------------------------------------------------------------------------------
// the walk round
// ------------------------------------------------------------------------------------------
struct Lane {
    float x0, y0, px, py;
    Pcg rng;
    uint32_t sample, depth;
    bool on_n;
    float nx, ny;
    int32_t hint;
    float thp;
    float sr, sg, sb;
    float d0_d2;
    int32_t d0_slot;
};

// Per-lane event counters of one launch, packed two 16-bit fields per register (the kernels
// are register-bound; a lane makes at most steps_per_round <= 32767 steps per launch):
// a = steps | started << 16, b = absorbed | truncated << 16, c = Neumann hits, visits = LBVH
// nodes visited (32 bits).
struct LaneStats {
    uint32_t a, b, c, visits;
};

// The part of one walk step that follows the closest-point query.  One walk step = one item
// consumed from the reference's evaluation-point queue at one depth: separate ->
// handleBoundary -> sampleNeumann -> oneStepWalk (reference
// integrator/uniform/integrator.cu:128-211, 224-231, 336-444, 465-525).  `cp` is the result
// of lbvh nearest() for L.px,L.py (ignored when there is no Dirichlet boundary).
// Returns STEP_* status bits (STEP_ENDED when the walk ended in this step); the caller keeps
// the statistics, so no counter is ever touched inside a divergent branch.
// status bits returned by step_finish
enum : uint32_t { STEP_ENDED = 1u, STEP_ABSORBED = 2u, STEP_TRUNCATED = 4u, STEP_NEUMANN_HIT = 8u };

template <bool NEUMANN_EMISSIVE, bool NEUMANN_TREE, bool SOURCE = false, class STK = LdsColumn>
__device__ __forceinline__ uint32_t step_finish(const DevMesh &dm, const DevMesh &nm, const DevSettings &st, Lane &L,
                                                const Closest cp, const STK &stk, const DevSource &src = DevSource{})
{
    const bool has_d = dm.n_segs > 0, has_n = nm.n_segs > 0;
    const float eps = st.eps;
    const float px = L.px, py = L.py;

    // ---- separateEvaluationPoint -------------------------------------------------------
    float R_D = WOST_INF;
    if (has_d) {
        L.hint = cp.slot;
        const float4 a = dm.segA[cp.slot];
        const float inv = dm.segInv[cp.slot];
        const float wx = px - a.x, wy = py - a.y;
        const float uv = dot2(wx, wy, a.z, a.w) * inv;       // computeProjectionRatio
        const float cr = cross2(a.z, a.w, wx, wy);           // checkPointSide
        const int side = (0.0f < cr) - (cr < 0.0f);
        R_D = sqrtf(cp.d2);
        const bool in_shell = (R_D < eps) && (uv > 0.0f && uv < 1.0f);
        if (in_shell) {
            // ---- handleBoundary ----
            float r, g, b;
            surface_color(dm.segCol + 12 * (size_t)cp.slot, side, uv, r, g, b);
            r *= st.dirichlet_intensity; g *= st.dirichlet_intensity; b *= st.dirichlet_intensity;
            r *= L.thp; g *= L.thp; b *= L.thp;
            L.sr = r + L.sr; L.sg = g + L.sg; L.sb = b + L.sb;
            return STEP_ENDED | STEP_ABSORBED;
        }
    }
    float R_N = WOST_INF;
    if (has_n) R_N = closest_silhouette<NEUMANN_TREE>(nm, px, py, R_D, stk);
    float R_B = fmaxf(WOST_R_B_FLOOR, fminf(R_D, R_N));
    R_B *= WOST_R_B_SHRINK;
    if (isinf(R_B)) return STEP_ENDED;

    // ---- sampleSource (problems with a source term only) ------------------------------------
    if (SOURCE) {
        float cr_, cg_, cb_;
        if (source_sample<NEUMANN_TREE>(src, nm, eps, px, py, R_B, L.on_n, L.nx, L.ny, L.thp, L.rng, stk, cr_, cg_, cb_)) {
            L.sr = cr_ + L.sr; L.sg = cg_ + L.sg; L.sb = cb_ + L.sb;
        }
    }

    // ---- sampleNeumann -------------------------------------------------------------------
    if (has_n) {
        float cr_, cg_, cb_;
        if (neumann_sample<NEUMANN_EMISSIVE, NEUMANN_TREE>(nm, st.neumann_intensity, eps, px, py, R_B, L.on_n, L.nx, L.ny,
                                                           L.thp, L.rng, stk, cr_, cg_, cb_)) {
            L.sr = cr_ + L.sr; L.sg = cg_ + L.sg; L.sb = cb_ + L.sb;
        }
    }

    // ---- oneStepWalk ---------------------------------------------------------------------
    float dirx, diry, pdf, alpha;
    uniform_direction(L.on_n, L.nx, L.ny, L.rng, dirx, diry, pdf, alpha);
    float nxt_x, nxt_y, hnx, hny;
    const bool hit = walk_advance<NEUMANN_TREE>(nm, eps, px, py, R_B, L.on_n, L.nx, L.ny, dirx, diry, stk, nxt_x, nxt_y,
                                                hnx, hny);
    const uint32_t hit_count = hit ? 1u : 0u;
    // 1/pdf/alpha/2pi is exactly 1.0f in fp32 for both branches (tests/test_oracle_units.py),
    // so a unit throughput stays a unit throughput without three IEEE divisions
    if (L.thp != 1.0f) L.thp = L.thp / pdf / alpha / WOST_2PI;
    L.px = nxt_x; L.py = nxt_y;
    L.on_n = hit; L.nx = hnx; L.ny = hny;
    L.depth++;
    const bool truncated = L.depth == (uint32_t)st.max_depth;
    return (truncated ? (STEP_ENDED | STEP_TRUNCATED) : 0u) | (hit_count ? STEP_NEUMANN_HIT : 0u);
}

// step_finish for ALL lanes of a wave at once, `stepping` marking those that take the step: the same statements in the same
// order per walker, but the two tree queries on the Neumann side -- the closest silhouette vertex after the Dirichlet part, the
// walker's ray at the end -- are answered by the wave as a whole between the parts (wost_coop.h).  Neumann meshes on the tree only.
template <bool NEUMANN_EMISSIVE, bool SOURCE, class STK>
__device__ __forceinline__ uint32_t step_finish_wave(const DevMesh &dm, const DevMesh &nm, const DevSettings &st, Lane &L, const Closest cp,
                                                     bool stepping, const WavePool &W, const STK &stk, int ray_slot_trigger, const DevSource &src)
{
    const bool has_d = dm.n_segs > 0;
    const float eps = st.eps;
    const float px = L.px, py = L.py;
    uint32_t status = 0u;
    bool mid = stepping;
    // ---- separateEvaluationPoint / handleBoundary ----
    float R_D = WOST_INF;
    if (stepping && has_d) {
        L.hint = cp.slot;
        const float4 a = dm.segA[cp.slot];
        const float inv = dm.segInv[cp.slot];
        const float wx = px - a.x, wy = py - a.y;
        const float uv = dot2(wx, wy, a.z, a.w) * inv;
        const float cr = cross2(a.z, a.w, wx, wy);
        const int side = (0.0f < cr) - (cr < 0.0f);
        R_D = sqrtf(cp.d2);
        if ((R_D < eps) && (uv > 0.0f && uv < 1.0f)) {
            float r, g, b;
            surface_color(dm.segCol + 12 * (size_t)cp.slot, side, uv, r, g, b);
            r *= st.dirichlet_intensity; g *= st.dirichlet_intensity; b *= st.dirichlet_intensity;
            r *= L.thp; g *= L.thp; b *= L.thp;
            L.sr = r + L.sr; L.sg = g + L.sg; L.sb = b + L.sb;
            status = STEP_ENDED | STEP_ABSORBED;
            mid = false;
        }
    }
    const float R_N = closest_silhouette_wave(nm, px, py, R_D, mid, W, stk);
    float R_B = 0.0f, dirx = 0.0f, diry = 0.0f, pdf = 1.0f, alpha = 1.0f;
    bool go = false;
    if (mid) {
        R_B = fmaxf(WOST_R_B_FLOOR, fminf(R_D, R_N));
        R_B *= WOST_R_B_SHRINK;
        if (isinf(R_B)) {
            status = STEP_ENDED;
        } else {
            if (SOURCE) {
                float cr_, cg_, cb_;
                if (source_sample<true>(src, nm, eps, px, py, R_B, L.on_n, L.nx, L.ny, L.thp, L.rng, stk, cr_, cg_, cb_)) {
                    L.sr = cr_ + L.sr; L.sg = cg_ + L.sg; L.sb = cb_ + L.sb;
                }
            }
            {
                float cr_, cg_, cb_;
                if (neumann_sample<NEUMANN_EMISSIVE, true>(nm, st.neumann_intensity, eps, px, py, R_B, L.on_n, L.nx, L.ny, L.thp, L.rng, stk, cr_, cg_, cb_)) {
                    L.sr = cr_ + L.sr; L.sg = cg_ + L.sg; L.sb = cb_ + L.sb;
                }
            }
            uniform_direction(L.on_n, L.nx, L.ny, L.rng, dirx, diry, pdf, alpha);
            go = true;
        }
    }
    // ---- oneStepWalk: walk_advance with the ray answered by the wave ----
    float cxp = px, cyp = py;
    if (go && L.on_n) {
        cxp += eps * L.nx;
        cyp += eps * L.ny;
    }
    float t = 0.0f;
    int hi = -1;
    const bool hit = ray_closest_wave(nm, cxp, cyp, dirx, diry, R_B, go, t, hi, W, stk, ray_slot_trigger);
    if (go) {
        float nxt_x = px + R_B * dirx, nxt_y = py + R_B * diry, hnx = 0.0f, hny = 0.0f;
        if (hit) {
            hnx = nm.flat[hi].nx;
            hny = nm.flat[hi].ny;
            if (dot2(hnx, hny, dirx, diry) > 0) { hnx = -hnx; hny = -hny; }
            nxt_x = cxp + t * dirx;
            nxt_y = cyp + t * diry;
        }
        if (L.thp != 1.0f) L.thp = L.thp / pdf / alpha / WOST_2PI;
        L.px = nxt_x; L.py = nxt_y;
        L.on_n = hit; L.nx = hnx; L.ny = hny;
        L.depth++;
        const bool truncated = L.depth == (uint32_t)st.max_depth;
        status = (truncated ? (STEP_ENDED | STEP_TRUNCATED) : 0u) | (hit ? STEP_NEUMANN_HIT : 0u);
    }
    return status;
}

__device__ __forceinline__ void load_lane(const WalkQueue &q, uint32_t slot, Lane &L, uint32_t &pix)
{
    pix = q.pix[slot];
    L.x0 = q.x0[slot]; L.y0 = q.y0[slot];
    L.px = q.px[slot]; L.py = q.py[slot];
    L.rng.state = q.rng[slot]; L.rng.inc = 1;
    const uint32_t m = q.meta[slot];
    L.sample = META_SAMPLE(m); L.depth = META_DEPTH(m); L.on_n = META_ONN(m) != 0;
    L.nx = q.nx[slot]; L.ny = q.ny[slot];
    L.hint = q.hint[slot];
    L.thp = q.thp[slot];
    L.sr = q.sr[slot]; L.sg = q.sg[slot]; L.sb = q.sb[slot];
    L.d0_d2 = q.d0_d2[slot]; L.d0_slot = q.d0_slot[slot];
}

__device__ __forceinline__ void store_lane(const WalkQueue &q, uint32_t s, const Lane &L, uint32_t pix)
{
    q.pix[s] = pix;
    q.x0[s] = L.x0; q.y0[s] = L.y0;
    q.px[s] = L.px; q.py[s] = L.py;
    q.rng[s] = L.rng.state;
    q.meta[s] = META_PACK(L.sample, L.depth, L.on_n ? 1 : 0);
    q.nx[s] = L.nx; q.ny[s] = L.ny;
    q.hint[s] = L.hint;
    q.thp[s] = L.thp;
    q.sr[s] = L.sr; q.sg[s] = L.sg; q.sb[s] = L.sb;
    q.d0_d2[s] = L.d0_d2; q.d0_slot[s] = L.d0_slot;
}

// walkers order[0 .. n) of `in` (0 .. n without an order) copied to slots 0 .. n of `out`
__global__ __launch_bounds__(256) void gather_walkers_kernel(WalkQueue in, const uint32_t *order, uint32_t n, WalkQueue out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Lane L;
    uint32_t pix;
    load_lane(in, order ? order[i] : i, L, pix);
    store_lane(out, i, L, pix);
}

// REFILL = false: one thread per queue slot, the round ends after steps_per_round steps and the
// survivors are compacted (the throughput path: with many samples per pixel a slot keeps itself
// busy by regenerating).  REFILL = true: a fixed number of resident threads; a lane whose pixel
// is complete writes it out and takes the next unread slot of the input queue (one atomic per
// wave), so the whole solve is ONE launch and no lane idles while work is left -- the
// low-sample-count path (time-to-1spp), where regeneration cannot fill the lanes.
// SLACK = false, the kernel of every ordinary launch: node visits in the plain form, exact within dm.far2 of the Dirichlet mesh.
// A walker whose query starts beyond that (it leaked through a corner of the boundary or escaped through an open one, or the
// probe looks at the scene from afar) stops there, untouched, and goes to the far end of the output queue; the host runs those
// few walkers through the SLACK = true instantiation -- node visits exact at any distance (trav_visit<true>), which costs this
// kernel a quarter of its throughput: more visits and, above all, registers -- for the length of a walk and returns them.
// PERSIST (with REFILL): the persistent first launch of a many-sample solve.  The same code, minus the state that only the
// few-samples form needs: no wave-private reservation (a refill is rare: exactly the slots needed, one atomic per refill), no
// 32-bit totals of the lane counters (a lane adds its counters to the statistics when its pixel is complete: six atomics per
// pixel).  Eight registers fewer -- the instantiation wants 98 where six waves per SIMD allow 80, and what it spilled was not
// only cold: the PCG state went through scratch at every step (a build with 96 registers and five waves per SIMD was 7 % faster
// than the spilling one at the same five waves, profiles/r06_o_*).
template <bool NEUMANN_EMISSIVE, bool NEUMANN_TREE, bool REFILL = false, bool SOURCE = false, bool SLACK = false, bool PERSIST = false>
#ifndef WOST_ROUND_WAVES
#define WOST_ROUND_WAVES 6      // waves per SIMD the round kernel is compiled for (tuning builds override it)
#endif
__global__ __launch_bounds__(256, NEUMANN_TREE ? 4 : WOST_ROUND_WAVES) void walk_round_kernel(RoundParams P)
{
    extern __shared__ uint32_t lds_stack[];       // the traversal stack columns, one per lane
    uint32_t *stack = lds_stack + threadIdx.x;
    // thin waves (lane_shift > 0, the last launches of a solve): only every 2^shift-th lane holds a
    // walker, so a wave waits for the longest query of 64 >> shift walkers instead of 64
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t slot = tid >> P.lane_shift;
    const uint32_t n_in = *P.count_in;
    const bool valid = slot < n_in && (tid & ((1u << P.lane_shift) - 1u)) == 0u;
    Lane L;
    LaneStats S{0, 0, 0, 0};
    uint32_t pix = 0;
    bool alive = false;
    if (valid) {
        load_lane(P.in, P.order ? P.order[slot] : slot, L, pix);
        alive = L.sample < (uint32_t)P.st.spp;
    }
    bool open = valid;            // this lane holds a pixel whose result has not been written
    uint32_t wide[6] = {0, 0, 0, 0, 0, 0};   // REFILL: 32-bit totals of the packed 16-bit lane counters
    uint32_t pool_next = 0, pool_end = 0;    // REFILL: this wave's reserved input slots [next, end)
    // Per-lane state machine.  A lane either has an INNER node or a LEAF of the LBVH to visit
    // for its current walk position, WAITs with a finished query for the step logic, or is DONE
    // for this round.  Query lengths differ wildly between lanes, so instead of running every
    // lane's query to completion in lock step, each trip of the loop runs ONE of two bodies
    // -- "visit one node" or "finish a step and start the next query" -- whichever more lanes
    // of the wave are ready for (weighted), while the lanes of the other kind accumulate.
    enum { MODE_TRAV = 1, MODE_WAIT = 3, MODE_DONE = 4, MODE_REFILL = 5, MODE_FAR = 6, MODE_HUGE = 7 };
    const bool has_d = P.dm.n_segs > 0;
    int mode = alive ? MODE_WAIT : MODE_DONE;
    bool fresh = true;   // first trip: no finished step yet, only start the query
    int budget = P.steps_per_round;
    Trav T = trav_begin(Closest{WOST_INF, -1});
    uint32_t trav_trips = 0, step_trips = 0;   // wave-uniform: scheduler diagnostics
#ifdef WOST_TRACK
    uint32_t trk_leaf = 0, trk_ge6 = 0, trk_ge10 = 0, trk_ge14 = 0;
    int trk_sp = 0;
#define WOST_TRACK_VISIT_PRE() do { trk_leaf += (T.level == P.dm.levels) ? 1u : 0u; } while (0)
#define WOST_TRACK_VISIT_POST() do { trk_sp = max(trk_sp, T.sp); trk_ge6 += T.sp >= 6; trk_ge10 += T.sp >= 10; trk_ge14 += T.sp >= 14; } while (0)
#else
#define WOST_TRACK_VISIT_PRE() do {} while (0)
#define WOST_TRACK_VISIT_POST() do {} while (0)
#endif
    const LdsColumn stk{stack, (uint32_t)P.stack_stride};
    for (;;) {
        if (REFILL) {
            // lanes whose pixel is complete: write it, then take the next unread input slot
            const unsigned long long need = __ballot(mode == MODE_REFILL);
            if (need) {
                // slots come from a wave-private reservation of 64 (one atomic on the shared cursor
                // per 64 pixels, not per refill: same-address atomics serialise in L2)
                // (many samples per pixel: a refill is rare -- config 2: one per wave and 30 walk steps -- and unread slots
                // parked in 6 000 private reservations would be a fifth of the frame when the cursor runs dry: reserve = 0 takes
                // exactly the slots needed)
                const int lane_ = threadIdx.x & 63;
                const uint32_t needed = (uint32_t)__popcll(need);
                const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane_) - 1ull));
                uint32_t s2;
                if (PERSIST) {
                    uint32_t fresh_base = 0;
                    if (lane_ == 0) fresh_base = atomicAdd(P.cursor, needed);
                    s2 = __shfl(fresh_base, 0) + rank;
                } else {
                    const uint32_t avail = pool_end - pool_next;
                    uint32_t fresh_base = 0;
                    const uint32_t want = P.reserve > 0 ? (uint32_t)P.reserve : needed - min(needed, avail);
                    if (needed > avail) {
                        if (lane_ == 0) fresh_base = atomicAdd(P.cursor, want);
                        fresh_base = __shfl(fresh_base, 0);
                    }
                    s2 = rank < avail ? pool_next + rank : fresh_base + (rank - avail);
                    if (needed > avail) {
                        pool_next = fresh_base + (needed - avail);
                        pool_end = fresh_base + want;
                    } else {
                        pool_next += needed;
                    }
                }
                // the queue is dry: with leave_dry the wave stops here -- no new step starts (budget 0), queries in flight finish
                // their step, pixels still open go to the output queue at the end of the kernel like the survivors of a round
                if (P.leave_dry && __ballot(mode == MODE_REFILL && s2 >= n_in)) budget = 0;
                if (mode == MODE_REFILL) {
                    if (open) {       // (a walker that left for the slack launch is not resolved here: `open` is false)
                        float *f = P.field + 3 * (size_t)((int32_t)pix - P.field_base);
                        const float spp = (float)P.st.spp;
                        f[0] = L.sr / spp; f[1] = L.sg / spp; f[2] = L.sb / spp;
                    }
                    open = false;
                    if (PERSIST) {
                        StatsDev *sd = my_stats(P.stats);
                        if (S.a & 0xffffu) atomicAdd(&sd->steps, (unsigned long long)(S.a & 0xffffu));
                        if (S.a >> 16) atomicAdd(&sd->started, (unsigned long long)(S.a >> 16));
                        if (S.b & 0xffffu) atomicAdd(&sd->absorbed, (unsigned long long)(S.b & 0xffffu));
                        if (S.b >> 16) atomicAdd(&sd->truncated, (unsigned long long)(S.b >> 16));
                        if (S.c) atomicAdd(&sd->nhits, (unsigned long long)S.c);
                        if (S.visits) atomicAdd(&sd->inner_visits, (unsigned long long)S.visits);
                    } else {
                        wide[0] += S.a & 0xffffu; wide[1] += S.a >> 16; wide[2] += S.b & 0xffffu; wide[3] += S.b >> 16;
                        wide[4] += S.c; wide[5] += S.visits;
                    }
                    S = LaneStats{0, 0, 0, 0};
                    if (s2 < n_in) {
                        load_lane(P.in, P.order ? P.order[s2] : s2, L, pix);
                        open = true;
                        alive = L.sample < (uint32_t)P.st.spp;
                        fresh = true;
                        mode = alive ? MODE_WAIT : MODE_REFILL;
                    } else {
                        mode = MODE_DONE;
                    }
                }
            }
        }
        const int n_trav = __popcll(__ballot(mode == MODE_TRAV));
        const int n_wait = __popcll(__ballot(mode == MODE_WAIT));
        if (n_trav + n_wait == 0) {
            if (REFILL && __ballot(mode == MODE_REFILL)) continue;
            break;
        }
        if (n_wait * P.wait_weight >= n_trav * 8) {
            ++step_trips;
            // ---- step phase ----
            uint32_t wave_status = 0u;
            if (NEUMANN_TREE && P.coop) {
                // (8-byte LDS atomics: the pools start at an 8-byte boundary whatever lies before the dynamic segment)
                uint32_t *pw = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(lds_stack + P.pool_offset) + 7u) & ~(uintptr_t)7u) +
                               (threadIdx.x >> 6) * (2 * P.pool_cap + kPoolOwnerWords);
                const WavePool W{pw + kPoolOwnerWords, pw + kPoolOwnerWords + P.pool_cap, pw, P.pool_cap};
                wave_status = step_finish_wave<NEUMANN_EMISSIVE, SOURCE>(P.dm, P.nm, P.st, L, T.best, mode == MODE_WAIT && !fresh, W, stk, P.ray_slot_trigger, P.src);
            }
            if (mode == MODE_WAIT) {
                if (!fresh) {
                    const uint32_t status = (NEUMANN_TREE && P.coop) ? wave_status
                                                                     : step_finish<NEUMANN_EMISSIVE, NEUMANN_TREE, SOURCE>(P.dm, P.nm, P.st, L, T.best, stk, P.src);
                    const bool ended = (status & STEP_ENDED) != 0u;
                    S.b += ((status >> 1) & 1u) | (((status >> 2) & 1u) << 16);
                    S.c += (status >> 3) & 1u;
                    if (ended) {
                        // next sample of this pixel starts right away (generateEvaluationPoints,
                        // reference integrator.cu:90-99 + workqueue.h:99-110)
                        L.sample++;
                        L.px = L.x0; L.py = L.y0;
                        L.depth = 0; L.on_n = false; L.nx = 0.0f; L.ny = 0.0f;
                        L.thp = 1.0f;
                        L.hint = L.d0_slot;
                        alive = L.sample < (uint32_t)P.st.spp;
                    }
                    --budget;
                }
                fresh = false;
                if (alive && budget > 0) {
                    if (!has_d || L.depth == 0) {
                        // depth 0 starts at the same point for every sample of the pixel: cached
                        S.a += 1u + ((L.depth == 0) ? 0x10000u : 0u);
                        T.best = Closest{L.d0_d2, L.d0_slot};
                        mode = MODE_WAIT;
                    } else {
                        T = trav_begin(slot_candidate(P.dm, L.hint, L.px, L.py));
                        if (!SLACK && T.best.d2 > P.dm.far2) {
                            mode = MODE_FAR;      // not started, not counted: the SLACK launch takes the step from here
                            if (REFILL) {
                                // a resident lane must go on draining the input queue (parked, it would strand the unread slots
                                // of its wave's reservation, and a launch whose lanes all strayed would drop the rest of the
                                // frame): the walker goes to the far end of the output queue right away -- strayed walkers are
                                // few, so one atomic each -- and the lane takes the next pixel
                                const uint32_t k = atomicAdd(P.count_far, 1u);
                                store_lane(P.out, P.out_capacity - 1u - k, L, pix);
                                open = false;
                                mode = MODE_REFILL;
                            }
                        } else {
                            S.a += 1u;
                            // SLACK: a query from so far away that every segment of the mesh ties within rounding is answered
                            // by the whole wave at the end of this trip
                            mode = (SLACK && T.best.d2 > P.dm.huge2) ? MODE_HUGE : MODE_TRAV;
                        }
                    }
                } else {
                    mode = (REFILL && !alive) ? MODE_REFILL : MODE_DONE;
                }
            }
            if (SLACK) {
                unsigned long long hb = __ballot(mode == MODE_HUGE);
                while (hb) {
                    const int src = __builtin_ctzll(hb);
                    const Closest r = closest_point_wave(P.dm, __shfl(L.px, src), __shfl(L.py, src));
                    if ((int)(threadIdx.x & 63) == src) {
                        T.best = r;
                        mode = MODE_WAIT;
                    }
                    hb &= hb - 1;
                }
            }
        } else {
            // ---- traversal phase: every traversing lane visits one node ----
            ++trav_trips;
#ifndef WOST_TRAV_BURST
#define WOST_TRAV_BURST 3       // the burst the kernel is unrolled for (the handle's default trav_burst must equal it)
#endif
            if (!PERSIST && P.trav_burst == WOST_TRAV_BURST) {
                // the default burst, unrolled: no loop counter, and the compiler may start a visit's node load early
#pragma unroll
                for (int b = 0; b < WOST_TRAV_BURST; ++b) {
                    if (mode == MODE_TRAV) {
                        S.visits++;
                        WOST_TRACK_VISIT_PRE();
                        if (!trav_visit<SLACK>(P.dm, L.px, L.py, T, stk)) mode = MODE_WAIT;
                        WOST_TRACK_VISIT_POST();
                    }
                }
            } else {
                for (int b = 0; b < P.trav_burst; ++b) {
                    if (mode == MODE_TRAV) {
                        S.visits++;
                        WOST_TRACK_VISIT_PRE();
                        if (!trav_visit<SLACK>(P.dm, L.px, L.py, T, stk)) mode = MODE_WAIT;
                        WOST_TRACK_VISIT_POST();
                    }
                }
            }
        }
    }
    // ---- resolve finished pixels (reference integrator.cu:616-620) -------------------------
    if (open && !alive) {
        float *f = P.field + 3 * (size_t)((int32_t)pix - P.field_base);
        const float spp = (float)P.st.spp;
        f[0] = L.sr / spp; f[1] = L.sg / spp; f[2] = L.sb / spp;
    }
    // ---- stream compaction of the survivors: ballot + popcount, one atomic per block ------
    const int lane = threadIdx.x & 63;
    const bool far = !SLACK && mode == MODE_FAR;
    uint32_t s = block_push(alive && open && !far, P.count_out);
    if (!SLACK) {
        __syncthreads();      // block_push reuses its shared words
        const uint32_t k = block_push(alive && open && far, P.count_far);
        if (far) s = P.out_capacity - 1u - k;
    }
    if (alive && open) store_lane(P.out, s, L, pix);
    if (REFILL && P.leave_dry) {
        // a persistent launch hands its open pixels to the rounds: how many walk steps a pixel still needs is known well by now --
        // its samples so far took S.a steps (the lane's counters restart with every pixel it takes), the rest will take about as
        // many each.  The host starts the longest remainders at once, four lanes to a walker, beside the rounds of the others.
        const float done = (float)L.sample, left = (float)((uint32_t)P.st.spp - L.sample);
        const float est = left * ((float)max(S.a & 0xffffu, 4u) / fmaxf(done, 1.0f));
        const bool is_long = alive && open && !far && est >= P.long_steps;
        if (alive && open && !far) P.out.est[s] = est;
        const unsigned long long lb = __ballot(is_long);
        if (lane == 0 && lb) atomicAdd(P.count_long, (uint32_t)__popcll(lb));
    }
    // ---- statistics: wave reduction, one atomic per counter per wave -----------------------
    uint32_t v[7] = {(S.a & 0xffffu) + wide[0], (S.a >> 16) + wide[1], (S.b & 0xffffu) + wide[2], (S.b >> 16) + wide[3],
                     S.c + wide[4], S.visits + wide[5], 0u};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        uint32_t x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        v[k] = x;
    }
    if (lane == 0) {
        StatsDev *st = my_stats(P.stats);
        if (v[0]) atomicAdd(&st->steps, (unsigned long long)v[0]);
        if (v[1]) atomicAdd(&st->started, (unsigned long long)v[1]);
        if (v[2]) atomicAdd(&st->absorbed, (unsigned long long)v[2]);
        if (v[3]) atomicAdd(&st->truncated, (unsigned long long)v[3]);
        if (v[4]) atomicAdd(&st->nhits, (unsigned long long)v[4]);
        if (v[5]) atomicAdd(&st->inner_visits, (unsigned long long)v[5]);
        atomicAdd(&st->trav_trips, (unsigned long long)trav_trips);
        atomicAdd(&st->step_trips, (unsigned long long)step_trips);
    }
#ifdef WOST_TRACK
    {
        uint32_t w[4] = {trk_leaf, trk_ge6, trk_ge10, trk_ge14};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            for (int off = 32; off > 0; off >>= 1) w[k] += __shfl_down(w[k], off);
        for (int off = 32; off > 0; off >>= 1) trk_sp = max(trk_sp, __shfl_down(trk_sp, off));
        if (lane == 0) {
            StatsDev *st = my_stats(P.stats);
            atomicAdd(&st->leaf_visits, (unsigned long long)w[0]);
            atomicAdd(&st->sp_ge6, (unsigned long long)w[1]);
            atomicAdd(&st->sp_ge10, (unsigned long long)w[2]);
            atomicAdd(&st->sp_ge14, (unsigned long long)w[3]);
            atomicMax(&st->max_stack, (unsigned long long)trk_sp);
        }
    }
#endif
}


// ------------------------------------------------------------------------------------------
// the walk round of an under-filled launch: four lanes per walker (wost_quad.h)
// ------------------------------------------------------------------------------------------
// Same rounds, same queue records, same per-walker arithmetic as walk_round_kernel -- and so the same field, counters
// and visiting order -- but a quad of lanes holds ONE walker: the closest-point descent is shared between the four
// lanes (one child box each), everything else runs replicated.  The host launches it when the walkers left would fill
// less than a quarter of the resident lanes, where a launch lasts as long as its longest chain of dependent visits.
template <bool NEUMANN_EMISSIVE, bool NEUMANN_TREE, bool SOURCE, bool SLACK>
__global__ __launch_bounds__(256, 4) void walk_quad_kernel(RoundParams P)
{
    extern __shared__ uint32_t lds_stack[];
    const int j = threadIdx.x & 3;
    const QuadColumn stk{lds_stack + (threadIdx.x >> 2) * P.stack_stride};     // stack_stride = entries per column here
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    // lane_shift > 0 (the very last rounds): only every 2^shift-th quad holds a walker, down to one walker per wave -- such a
    // wave never waits for another walker's query
    const uint32_t quad_id = tid >> 2;
    const uint32_t n_in = *P.count_in;
    uint32_t slot = quad_id >> P.lane_shift;
    bool valid = slot < n_in && (quad_id & ((1u << P.lane_shift) - 1u)) == 0u;
    if (P.thin_blocks > 0u) {
        const uint32_t quads_per_block = blockDim.x >> 2;
        if (blockIdx.x < P.thin_blocks) {
            slot = quad_id >> 2;
            valid = slot < P.thin_count && (quad_id & 3u) == 0u;
        } else {
            slot = P.thin_count + (quad_id - P.thin_blocks * quads_per_block);
            valid = slot < n_in;
        }
    }
    Lane L;
    LaneStats S{0, 0, 0, 0};
    uint32_t pix = 0;
    bool alive = false;
    if (valid) {
        load_lane(P.in, P.order ? P.order[slot] : slot, L, pix);
        alive = L.sample < (uint32_t)P.st.spp;
    }
    const bool open = valid;
    enum { MODE_TRAV = 1, MODE_WAIT = 3, MODE_DONE = 4, MODE_FAR = 6, MODE_HUGE = 7 };
    const bool has_d = P.dm.n_segs > 0;
    int mode = alive ? MODE_WAIT : MODE_DONE;       // quad-uniform throughout
    bool fresh = true;
    int budget = P.steps_per_round;
    Trav T = trav_begin(Closest{WOST_INF, -1});
    uint32_t trav_trips = 0, step_trips = 0;
    for (;;) {
        const int n_trav = __popcll(__ballot(mode == MODE_TRAV));
        const int n_wait = __popcll(__ballot(mode == MODE_WAIT));
        if (n_trav + n_wait == 0) break;
        if (n_wait * P.wait_weight >= n_trav * 8) {
            ++step_trips;
            if (mode == MODE_WAIT) {
                if (!fresh) {
                    const uint32_t status = step_finish<NEUMANN_EMISSIVE, NEUMANN_TREE, SOURCE>(P.dm, P.nm, P.st, L, T.best, stk, P.src);
                    const bool ended = (status & STEP_ENDED) != 0u;
                    S.b += ((status >> 1) & 1u) | (((status >> 2) & 1u) << 16);
                    S.c += (status >> 3) & 1u;
                    if (ended) {
                        L.sample++;
                        L.px = L.x0; L.py = L.y0;
                        L.depth = 0; L.on_n = false; L.nx = 0.0f; L.ny = 0.0f;
                        L.thp = 1.0f;
                        L.hint = L.d0_slot;
                        alive = L.sample < (uint32_t)P.st.spp;
                    }
                    --budget;
                }
                fresh = false;
                if (alive && budget > 0) {
                    if (!has_d || L.depth == 0) {
                        S.a += 1u + ((L.depth == 0) ? 0x10000u : 0u);
                        T.best = Closest{L.d0_d2, L.d0_slot};
                        mode = MODE_WAIT;
                    } else {
                        T = trav_begin(slot_candidate(P.dm, L.hint, L.px, L.py));
                        if (!SLACK && T.best.d2 > P.dm.far2) {
                            mode = MODE_FAR;
                        } else {
                            S.a += 1u;
                            mode = (SLACK && T.best.d2 > P.dm.huge2) ? MODE_HUGE : MODE_TRAV;
                        }
                    }
                } else {
                    mode = MODE_DONE;
                }
            }
            if (SLACK) {
                unsigned long long hb = __ballot(mode == MODE_HUGE);
                while (hb) {
                    const int src = __builtin_ctzll(hb);
                    const Closest r = closest_point_wave(P.dm, __shfl(L.px, src), __shfl(L.py, src));
                    if (((int)(threadIdx.x & 63) >> 2) == (src >> 2)) {
                        T.best = r;
                        mode = MODE_WAIT;
                    }
                    hb &= ~(0xfull << (src & ~3));
                }
            }
        } else {
            ++trav_trips;
            for (int b = 0; b < P.trav_burst; ++b) {
                if (mode == MODE_TRAV) {
                    S.visits++;
                    if (!quad_visit<SLACK>(P.dm, L.px, L.py, T, stk, j)) mode = MODE_WAIT;
                }
            }
        }
    }
    const bool lead = j == 0;       // one lane of the quad speaks for the walker
    if (open && !alive && lead) {
        float *f = P.field + 3 * (size_t)((int32_t)pix - P.field_base);
        const float spp = (float)P.st.spp;
        f[0] = L.sr / spp; f[1] = L.sg / spp; f[2] = L.sb / spp;
    }
    const int lane = threadIdx.x & 63;
    const bool far = !SLACK && mode == MODE_FAR;
    uint32_t s = block_push(alive && open && !far && lead, P.count_out);
    if (!SLACK) {
        __syncthreads();
        const uint32_t k = block_push(alive && open && far && lead, P.count_far);
        if (far) s = P.out_capacity - 1u - k;
    }
    if (alive && open && lead) store_lane(P.out, s, L, pix);
    uint32_t v[7] = {(S.a & 0xffffu), (S.a >> 16), (S.b & 0xffffu), (S.b >> 16), S.c, S.visits, 0u};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        uint32_t x = lead ? v[k] : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        v[k] = x;
    }
    if (lane == 0) {
        StatsDev *st = my_stats(P.stats);
        if (v[0]) atomicAdd(&st->steps, (unsigned long long)v[0]);
        if (v[1]) atomicAdd(&st->started, (unsigned long long)v[1]);
        if (v[2]) atomicAdd(&st->absorbed, (unsigned long long)v[2]);
        if (v[3]) atomicAdd(&st->truncated, (unsigned long long)v[3]);
        if (v[4]) atomicAdd(&st->nhits, (unsigned long long)v[4]);
        if (v[5]) atomicAdd(&st->inner_visits, (unsigned long long)v[5]);
        atomicAdd(&st->trav_trips, (unsigned long long)trav_trips);
        atomicAdd(&st->step_trips, (unsigned long long)step_trips);
    }
}

// ------------------------------------------------------------------------------------------
// batch query kernels (the lbvh::query_device call sites, exposed for tests and SDF renders)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void closest_point_kernel(DevMesh m, const float *pts, int n, int32_t *out_idx,
                                                            float *out_dist, float *out_uv, int32_t *out_side,
                                                            int stack_stride)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float qx = pts[2 * i], qy = pts[2 * i + 1];
    const Closest c = closest_point(m, qx, qy, slot_candidate(m, 0, qx, qy), stack, stack_stride);
    const float4 a = m.segA[c.slot];
    const float inv = m.segInv[c.slot];
    const float wx = qx - a.x, wy = qy - a.y;
    const float cr = cross2(a.z, a.w, wx, wy);
    if (out_idx) out_idx[i] = m.segOrig[c.slot];
    if (out_dist) out_dist[i] = sqrtf(c.d2);
    if (out_uv) out_uv[i] = dot2(wx, wy, a.z, a.w) * inv;
    if (out_side) out_side[i] = (0.0f < cr) - (cr < 0.0f);
}

__global__ __launch_bounds__(256) void sdf_kernel(DevMesh m, DevProbe probe, int width, int height, int which,
                                                  float *out, int stack_stride)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
    const int pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= width * height) return;
    float x, y;
    eval_point(probe, pid % width, pid / width, width, height, x, y);
    float d = WOST_INF;
    if (m.n_segs > 0) {
        if (which == WOST_MESH_DIRICHLET) {
            d = sqrtf(closest_point(m, x, y, slot_candidate(m, 0, x, y), stack, stack_stride).d2);
        } else {
            d = (m.n_segs <= WOST_FLAT_MAX) ? closest_silhouette_flat(m, x, y, WOST_INF)
                                             : closest_silhouette_tree(m, x, y, WOST_INF, LdsColumn{stack, (uint32_t)stack_stride});
        }
    }
    out[pid] = d;
}

__global__ __launch_bounds__(256) void source_kernel(DevSource src, DevProbe probe, int width, int height, float *out)
{
    const int pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= width * height) return;
    float x, y, r = 0.0f, g = 0.0f, b = 0.0f;
    eval_point(probe, pid % width, pid / width, width, height, x, y);
    if (src.rgb) source_eval(src, x, y, r, g, b);
    out[3 * (size_t)pid] = r; out[3 * (size_t)pid + 1] = g; out[3 * (size_t)pid + 2] = b;
}

__global__ __launch_bounds__(256) void silhouette_kernel(DevMesh m, const float *pts, const float *rmax, int n,
                                                         float *out)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk{lds_stack + threadIdx.x, (uint32_t)blockDim.x};
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = rmax ? rmax[i] : WOST_INF;
    out[i] = (m.n_segs <= WOST_FLAT_MAX) ? closest_silhouette_flat(m, pts[2 * i], pts[2 * i + 1], r)
                                         : closest_silhouette_tree(m, pts[2 * i], pts[2 * i + 1], r, stk);
}

__global__ __launch_bounds__(256) void ray_kernel(DevMesh m, const float *o, const float *d, const float *tmax, int n,
                                                  int32_t *out_hit, float *out_t, int32_t *out_idx)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk{lds_stack + threadIdx.x, (uint32_t)blockDim.x};
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t;
    int idx;
    const bool hit = (m.n_segs <= WOST_FLAT_MAX)
                         ? ray_closest_flat(m, o[2 * i], o[2 * i + 1], d[2 * i], d[2 * i + 1], tmax[i], t, idx)
                         : ray_tree<false>(m, o[2 * i], o[2 * i + 1], d[2 * i], d[2 * i + 1], tmax[i], t, idx, stk);
    out_hit[i] = hit ? 1 : 0;
    out_t[i] = t;
    out_idx[i] = idx;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

// shared with the other translation units of the library (wost_vmm.hip)
int set_error(int code, const std::string &msg) { return fail(code, msg); }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(WOST_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

struct DeviceMeshStorage {
    DevMesh view{};
    std::vector<void *> allocs;
    HostTree host;
};


template <class T>
static hipError_t upload(std::vector<void *> &allocs, const T *src, size_t count, const T **dst)
{
    *dst = nullptr;
    if (count == 0) return hipSuccess;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) return e;
    allocs.push_back(p);
    e = hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
    *dst = reinterpret_cast<const T *>(p);
    return e;
}

// the run boxes over consecutive ORIGINAL indices and the sampler's compact copies (an emissive boundary on the tree only):
// a function of the flat records alone
static int upload_run_boxes(const std::vector<FlatSeg> &flat_recs, DeviceMeshStorage &s);

// the host builder's tree (lbvh_build.cpp), uploaded array by array: the checker of the device build, small meshes,
// WOST_HOST_BUILD=1 and the tree-quality knobs
static int upload_mesh_host(const wost_mesh_desc &d, DeviceMeshStorage &s)
{
    if (d.n_segs < 0 || d.n_verts < 0) return fail(WOST_ERR_INVALID, "negative mesh size");
    // developer knobs for tree-quality experiments (defaults: refined, minimal depth)
    const char *e_refine = getenv("WOST_TREE_REFINE");
    const char *e_extra = getenv("WOST_TREE_EXTRA_LEVELS");
    const bool refine = e_refine ? atoi(e_refine) != 0 : true;
    const int extra = e_extra ? std::max(0, std::min(3, atoi(e_extra))) : 0;
    if (build_tree(d.n_verts, d.verts, d.n_segs, d.segs, d.colors, &s.host, refine, extra) != 0)
        return fail(WOST_ERR_INVALID, "mesh: segment index out of range or null arrays");
    const HostTree &t = s.host;
    DevMesh &v = s.view;
    v = DevMesh{};
    v.n_segs = t.n_segs;
    if (t.n_segs == 0) return WOST_OK;
    v.n_sil = (int32_t)t.sil.size();
    v.levels = t.levels;
    {
        // the boxes are padded by 2^-21 of the largest coordinate; the rounding of a box distance grows with the distance
        // to the box (about 10^-7 of it): up to one and a half extents from the closest segment the padding covers it
        float ext = 0.0f;
        for (int i = 0; i < d.n_segs; ++i)
            for (int e = 0; e < 2; ++e) {
                const int32_t vi = d.segs[2 * i + e];
                ext = std::max(ext, std::max(std::fabs(d.verts[2 * vi]), std::fabs(d.verts[2 * vi + 1])));
            }
        v.far2 = 2.25f * ext * ext;
        v.huge2 = 4096.0f * ext * ext;       // 64 extents
    }
    v.first_leaf = t.first_leaf;
    v.emissive = 0;
    for (float c : t.flatCol)
        if (c != 0.0f) v.emissive = 1;
    HIP_TRY(upload(s.allocs, reinterpret_cast<const float4 *>(t.nodes.data()), t.nodes.size() / 4, &v.nodes));
    HIP_TRY(upload(s.allocs, reinterpret_cast<const float4 *>(t.segA.data()), t.segA.size() / 4, &v.segA));
    HIP_TRY(upload(s.allocs, t.segInv.data(), t.segInv.size(), &v.segInv));
    HIP_TRY(upload(s.allocs, t.segOrig.data(), t.segOrig.size(), &v.segOrig));
    HIP_TRY(upload(s.allocs, t.segCol.data(), t.segCol.size(), &v.segCol));
    HIP_TRY(upload(s.allocs, reinterpret_cast<const DevFlatSeg *>(t.flat.data()), t.flat.size(), &v.flat));
    HIP_TRY(upload(s.allocs, t.flatCol.data(), t.flatCol.size(), &v.flatCol));
    HIP_TRY(upload(s.allocs, reinterpret_cast<const DevSilVertex *>(t.sil.data()), t.sil.size(), &v.sil));
    {
        // the normals of every vertex's two segments beside it: the flat silhouette test of a small boundary reads them with the vertex
        std::vector<float> sn(t.sil.size() * 4, 0.0f);
        for (size_t k = 0; k < t.sil.size(); ++k) {
            const SilVertex &sv = t.sil[k];
            if (sv.prev >= 0) { sn[4 * k] = t.flat[(size_t)sv.prev].nx; sn[4 * k + 1] = t.flat[(size_t)sv.prev].ny; }
            if (sv.next >= 0) { sn[4 * k + 2] = t.flat[(size_t)sv.next].nx; sn[4 * k + 3] = t.flat[(size_t)sv.next].ny; }
        }
        HIP_TRY(upload(s.allocs, reinterpret_cast<const float4 *>(sn.data()), t.sil.size(), &v.silN));
    }
    HIP_TRY(upload(s.allocs, reinterpret_cast<const float4 *>(t.cones.data()), t.cones.size() / 4, &v.cones));
    {
        // closest_point_wave: the occupied slots, compact; the operands are those of the leaf-level node records
        std::vector<float> box, hl;
        std::vector<int32_t> id;
        for (size_t k = 0; k < t.segOrig.size(); ++k) {
            if (t.segOrig[k] == kFarIndex) continue;
            const float *nd = &t.nodes[((size_t)t.first_leaf + k / 4) * WOST_NODE_FLOATS] + (k & 3);
            box.insert(box.end(), {nd[0], nd[4], nd[8], nd[12]});
            hl.push_back(nd[16]);
            id.push_back((int32_t)k);
            id.push_back(t.segOrig[k]);
        }
        while (hl.size() % 256) {
            box.insert(box.end(), {1.0e18f, 1.0e18f, 1.0f, 0.0f});
            hl.push_back(0.0f);
            id.push_back(-1);
            id.push_back(kFarIndex);
        }
        v.n_scan = (int32_t)hl.size();
        HIP_TRY(upload(s.allocs, reinterpret_cast<const float4 *>(box.data()), hl.size(), &v.scanBox));
        HIP_TRY(upload(s.allocs, hl.data(), hl.size(), &v.scanHl));
        HIP_TRY(upload(s.allocs, reinterpret_cast<const int2 *>(id.data()), hl.size(), &v.scanId));
    }
    HIP_TRY(upload(s.allocs, reinterpret_cast<const int2 *>(t.segVerts.data()), t.segVerts.size() / 2, &v.segVerts));
    if (v.emissive && t.n_segs > WOST_FLAT_MAX) return upload_run_boxes(t.flat, s);
    return WOST_OK;
}

static int upload_run_boxes(const std::vector<FlatSeg> &flat_recs, DeviceMeshStorage &s)
{
    DevMesh &v = s.view;
    const int n_segs = (int)flat_recs.size();
    // boxes over runs of consecutive original indices, for the index-ordered sampling of emissive Neumann meshes
    // (sample_in_sphere_tree); padded like the tree's boxes, so that rounding never hides a segment the flat loop takes
    {
        float ext = 0.0f;
        for (const FlatSeg &f : flat_recs) ext = std::max(ext, std::max(std::max(std::fabs(f.ax), std::fabs(f.ay)), std::max(std::fabs(f.ax + f.ex), std::fabs(f.ay + f.ey))));
        const float pad = ext * 0x1p-18f + 1e-30f;
        std::vector<float> ob;
        int levels = 0;
        size_t prev_off = 0, prev_n = 0;
        for (int l = 0; l < 12; ++l) {
            const size_t run = (size_t)4 << (2 * l), n_runs = ((size_t)n_segs + run - 1) / run;
            v.obox_off[l] = (int32_t)(ob.size() / 4);
            for (size_t r = 0; r < n_runs; ++r) {
                float lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
                if (l == 0) {
                    for (size_t i = r * 4; i < std::min<size_t>(r * 4 + 4, (size_t)n_segs); ++i) {
                        const FlatSeg &f = flat_recs[i];
                        lo[0] = std::min(lo[0], std::min(f.ax, f.ax + f.ex)); hi[0] = std::max(hi[0], std::max(f.ax, f.ax + f.ex));
                        lo[1] = std::min(lo[1], std::min(f.ay, f.ay + f.ey)); hi[1] = std::max(hi[1], std::max(f.ay, f.ay + f.ey));
                    }
                    lo[0] -= pad; lo[1] -= pad; hi[0] += pad; hi[1] += pad;
                } else {
                    for (size_t c = r * 4; c < std::min(r * 4 + 4, prev_n); ++c) {
                        const float *b = &ob[(prev_off + c) * 4];
                        lo[0] = std::min(lo[0], b[0]); lo[1] = std::min(lo[1], b[1]); hi[0] = std::max(hi[0], b[2]); hi[1] = std::max(hi[1], b[3]);
                    }
                }
                ob.insert(ob.end(), {lo[0], lo[1], hi[0], hi[1]});
            }
            prev_off = (size_t)v.obox_off[l];
            prev_n = n_runs;
            levels = l + 1;
            if (n_runs <= 1) break;
        }
        v.obox_levels = levels;
        HIP_TRY(upload(s.allocs, reinterpret_cast<const float4 *>(ob.data()), ob.size() / 4, &v.obox));
        const size_t n4 = (flat_recs.size() + 3) / 4 * 4;
        std::vector<float> lens(n4, 0.0f), hl(n4, 0.0f), box(n4 * 4, 1.0e18f);
        for (size_t i = 0; i < flat_recs.size(); ++i) {
            const FlatSeg &f = flat_recs[i];
            lens[i] = f.len; hl[i] = f.hl;
            box[4 * i] = f.cx; box[4 * i + 1] = f.cy; box[4 * i + 2] = f.ux; box[4 * i + 3] = f.uy;
        }
        HIP_TRY(upload(s.allocs, lens.data(), lens.size(), &v.lens));
        HIP_TRY(upload(s.allocs, hl.data(), hl.size(), &v.sampHl));
        HIP_TRY(upload(s.allocs, reinterpret_cast<const float4 *>(box.data()), n4, &v.sampBox));
    }
    return WOST_OK;
}

// Problem<2>::build_bvh: on the device (wost_build2.hip) for every mesh worth the launches; the host builder for the small ones
// (a four-segment Neumann box is done before the first kernel would start), behind WOST_HOST_BUILD=1, and for the tree-quality
// knobs.  Both give the same arrays bit for bit (wost_mesh_build_check, tests/test_gpu_build2.py).
static int upload_mesh(const wost_mesh_desc &d, DeviceMeshStorage &s)
{
    if (d.n_segs < 0 || d.n_verts < 0) return fail(WOST_ERR_INVALID, "negative mesh size");
    const char *e_host = getenv("WOST_HOST_BUILD");
    const bool knobs = getenv("WOST_TREE_REFINE") || getenv("WOST_TREE_EXTRA_LEVELS");
    int min_segs = 512;
    if (const char *e = getenv("WOST_DEVICE_BUILD_MIN")) min_segs = std::max(1, atoi(e));
    if ((e_host && atoi(e_host) != 0) || knobs || d.n_segs < min_segs) return upload_mesh_host(d, s);
    DeviceTree2 t;
    const int rc = build_tree_device(d, t);
    if (rc != WOST_OK) return rc;
    s.view = t.view;
    if (t.alloc) s.allocs.push_back(t.alloc);
    if (s.view.emissive && d.n_segs > WOST_FLAT_MAX) {
        // (rare: an emissive boundary on the tree) the run boxes are built on the host from the flat records
        std::vector<FlatSeg> flat_recs((size_t)d.n_segs);
        HIP_TRY(hipMemcpy(flat_recs.data(), s.view.flat, flat_recs.size() * sizeof(FlatSeg), hipMemcpyDeviceToHost));
        return upload_run_boxes(flat_recs, s);
    }
    return WOST_OK;
}

static_assert(sizeof(FlatSeg) == sizeof(DevFlatSeg), "flat segment layout");
static_assert(sizeof(SilVertex) == sizeof(DevSilVertex), "silhouette vertex layout");

}  // namespace wost

using namespace wost;

struct wost_context {
    int device = 0;
    wost_settings settings{};
    DevSettings dst{};
    DevProbe probe{};
    DeviceMeshStorage dm, nm;
    uint8_t *mask = nullptr;
    DevSource src{};                // rgb == nullptr: no source term
    size_t n_pixels = 0;
    // queues
    void *queue_mem[2] = {nullptr, nullptr};
    WalkQueue queue[2]{};
    uint32_t *counts = nullptr;     // [2]
    StatsDev *stats = nullptr;
    float *field = nullptr;         // n_pixels * 3 (used by wost_solve)
    uint32_t *host_count = nullptr; // pinned
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // options
    int steps_per_round = 0;   // 0 = automatic (see run_solve)
    int thin_waves = 1;        // spread the walkers of an under-full launch over more waves
    int block_size = 256;
    int wait_weight = 8;
    bool wait_weight_set = false;   // by wost_set_option: otherwise steps with tree queries on the Neumann side use weight 1
    int trav_burst = 3;
    bool trav_burst_set = false;    // by wost_set_option: otherwise the persistent launch runs bursts of 5 with wait weight 6
    int time_kernels = 1;
    int refill = -1;       // -1 = automatic (few samples per pixel), 0 = never, 1 = always
    int persist = -1;      // many samples per pixel, more walkers than resident lanes: the first launch is persistent (lanes take pixels,
                           //   longest expected chain first, until none is left; the rest of the solve runs in rounds): -1 = automatic, 0, 1
    int persist_order = 1; // ... in the order of wost_order.h (0: queue order; comparison runs)
    int few_order = 1;     // the one-launch path of few samples per pixel in that order too (config 2's frame at 1 / 2 / 4 spp: 2.64 -> 2.48, 4.09 -> 3.55, 6.79 -> 5.68 ms)
    int resident_blocks = 0;  // blocks of a one-launch / persistent launch; 0 = as many as the chip holds (tests: a few blocks drain a small frame)
    // what a persistent launch hands over (run_solve): the pixels expected to need `long_steps` walk steps or more run to their end at
    // once, four lanes to a walker, on a stream of higher priority beside the rounds of the others (0 = none do); the first round
    // takes the others in the order of their expected remainders (tail_sort), so that a wave's walkers finish together
    int long_steps = 1024;
    int long_cap = 32768;
    int tail_sort = 1;
    void *long_mem = nullptr;
    WalkQueue long_queue{};
    int long_thin = 2048;         // ... the first long_thin of them four to a wave, the others sixteen
    // Their stream has the highest priority: they are the critical path of the solve's end, and the runtime keeps the hardware
    // queues of a priority level apart from those of the others -- a stream of ordinary priority can land on the hardware queue
    // of the caller's stream (four queues per level), and then the launch meant to run BESIDE the rounds runs behind them (seen
    // in bench.py, which makes a second handle first: the solve waited 27 ms for it).
    hipStream_t long_stream = nullptr;
    hipEvent_t long_ev0 = nullptr, long_ev1 = nullptr;
    WalkOrder order;
    int quad = -1;         // four lanes per walker in under-filled launches: -1 = automatic, 0 = never, 1 = every ordinary round
    double quad_fill = 1.0;   // automatic: when 4 x walkers <= quad_fill x resident lanes
    int coop = 1;             // a Neumann mesh on the tree: its silhouette and ray queries by the wave as a whole (wost_coop.h); 0 = per lane
    int pool_cap = 0;         // ... tasks per pool and wave; 0 = automatic: 128 per level of the Neumann tree (a deeper tree keeps more tasks in flight)
    int ray_slot_trigger = 32;
    uint32_t *cursor = nullptr;
    hipStream_t far_stream = nullptr;          // the launches that take strayed walkers through the SLACK kernel (run_solve)
    hipEvent_t far_ev0 = nullptr, far_ev1 = nullptr;
    int n_cus = 256;
    StatsDev *host_stats = nullptr;            // pinned: the counters as they stood after each launch (last_launches)
    std::vector<wost_launch_info> last_launches;
};

namespace wost {
SceneView scene_view(wost_handle h)
{
    return SceneView{h->device, h->dm.view, h->nm.view, h->dst, h->probe, h->mask, h->stream, h->src};
}
}  // namespace wost

static void carve_queue(void *mem, size_t n, WalkQueue &q)
{
    // 8-byte array first, then 4-byte arrays
    char *p = reinterpret_cast<char *>(mem);
    q.rng = reinterpret_cast<uint64_t *>(p); p += n * 8;
    auto take = [&](size_t bytes) { void *r = p; p += bytes; return r; };
    q.pix = (uint32_t *)take(n * 4);
    q.x0 = (float *)take(n * 4); q.y0 = (float *)take(n * 4);
    q.px = (float *)take(n * 4); q.py = (float *)take(n * 4);
    q.meta = (uint32_t *)take(n * 4);
    q.nx = (float *)take(n * 4); q.ny = (float *)take(n * 4);
    q.hint = (int32_t *)take(n * 4);
    q.thp = (float *)take(n * 4);
    q.sr = (float *)take(n * 4); q.sg = (float *)take(n * 4); q.sb = (float *)take(n * 4);
    q.d0_d2 = (float *)take(n * 4);
    q.d0_slot = (int32_t *)take(n * 4);
    q.est = (float *)take(n * 4);
}

static void destroy_ctx(wost_context *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (void *p : c->dm.allocs) (void)hipFree(p);
    for (void *p : c->nm.allocs) (void)hipFree(p);
    if (c->mask) (void)hipFree(c->mask);
    if (c->src.rgb) (void)hipFree(const_cast<float *>(c->src.rgb));
    for (int i = 0; i < 2; ++i)
        if (c->queue_mem[i]) (void)hipFree(c->queue_mem[i]);
    if (c->counts) (void)hipFree(c->counts);
    if (c->stats) (void)hipFree(c->stats);
    if (c->field) (void)hipFree(c->field);
    if (c->cursor) (void)hipFree(c->cursor);
    order_free(c->order);
    if (c->long_mem) (void)hipFree(c->long_mem);
    if (c->long_stream) (void)hipStreamDestroy(c->long_stream);
    if (c->long_ev0) (void)hipEventDestroy(c->long_ev0);
    if (c->long_ev1) (void)hipEventDestroy(c->long_ev1);
    if (c->far_stream) (void)hipStreamDestroy(c->far_stream);
    if (c->far_ev0) (void)hipEventDestroy(c->far_ev0);
    if (c->far_ev1) (void)hipEventDestroy(c->far_ev1);
    if (c->host_count) (void)hipHostFree(c->host_count);
    if (c->host_stats) (void)hipHostFree(c->host_stats);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

namespace {
template <class T>
int64_t differing_bytes2(const T *a, const T *b, size_t count, int64_t *compared)
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

extern "C" {

// 0.2: the sync callback of a shared guiding network is also asked for the number of ranks (WOST_SYNC_RANKS_I64_HOST)
const char *wost_version(void) { return "wost-hip 0.2 (gfx950)"; }


int wost_mesh_build_check(const wost_mesh_desc *mesh, int device, int32_t repeat, double *host_ms, double *device_ms, int64_t *mismatch)
{
    if (!mesh || !mismatch) return fail(WOST_ERR_INVALID, "null argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(WOST_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n_dev) return fail(WOST_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    if (mesh->n_segs < 0 || mesh->n_verts < 0) return fail(WOST_ERR_INVALID, "negative mesh size");
    struct Owned {
        DeviceMeshStorage m;
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
        HIP_TRY(hipDeviceSynchronize());
        auto t0 = clock::now();
        int rc = upload_mesh_host(*mesh, ha.m);
        HIP_TRY(hipDeviceSynchronize());
        auto t1 = clock::now();
        if (rc != WOST_OK) return rc;
        DeviceTree2 t;
        rc = build_tree_device(*mesh, t);
        if (t.alloc) hb.m.allocs.push_back(t.alloc);
        hb.m.view = t.view;
        HIP_TRY(hipDeviceSynchronize());
        auto t2 = clock::now();
        if (rc != WOST_OK) return rc;
        best_host = std::min(best_host, std::chrono::duration<double, std::milli>(t1 - t0).count());
        best_dev = std::min(best_dev, std::chrono::duration<double, std::milli>(t2 - t1).count());
        std::swap(a.m, ha.m);
        std::swap(b.m, hb.m);
    }
    if (host_ms) *host_ms = best_host;
    if (device_ms) *device_ms = best_dev;
    const DevMesh &x = a.m.view, &y = b.m.view;
    for (int i = 0; i < 16; ++i) mismatch[i] = 0;
    const int64_t scalars = (x.n_segs != y.n_segs) + (x.n_sil != y.n_sil) + (x.levels != y.levels) + (x.first_leaf != y.first_leaf) + (x.emissive != y.emissive) +
                            (std::memcmp(&x.far2, &y.far2, 4) != 0) + (std::memcmp(&x.huge2, &y.huge2, 4) != 0) + (x.n_scan != y.n_scan);
    mismatch[14] = scalars;
    if (x.n_segs == 0 || scalars) return WOST_OK;
    const size_t cap = 3 * (size_t)x.first_leaf + 1, n_all = (size_t)x.first_leaf + cap, n_slots = cap * 4, n = (size_t)x.n_segs, nv = (size_t)x.n_sil;
    int64_t *cmp = &mismatch[15];
    mismatch[0] = differing_bytes2(x.nodes, y.nodes, n_all * (WOST_NODE_FLOATS / 4), cmp);
    mismatch[1] = differing_bytes2(x.cones, y.cones, n_all * 5, cmp);
    mismatch[2] = differing_bytes2(x.segA, y.segA, n_slots, cmp);
    mismatch[3] = differing_bytes2(x.segInv, y.segInv, n_slots, cmp);
    mismatch[4] = differing_bytes2(x.segOrig, y.segOrig, n_slots, cmp);
    mismatch[5] = differing_bytes2(x.segCol, y.segCol, n_slots * 12, cmp);
    mismatch[6] = differing_bytes2(x.segVerts, y.segVerts, n_slots, cmp);
    mismatch[7] = differing_bytes2(x.flat, y.flat, n, cmp);
    mismatch[8] = differing_bytes2(x.flatCol, y.flatCol, n * 12, cmp);
    mismatch[9] = differing_bytes2(x.sil, y.sil, nv, cmp);
    mismatch[10] = differing_bytes2(x.silN, y.silN, nv, cmp);
    mismatch[11] = differing_bytes2(x.scanBox, y.scanBox, (size_t)x.n_scan, cmp);
    mismatch[12] = differing_bytes2(x.scanHl, y.scanHl, (size_t)x.n_scan, cmp);
    mismatch[13] = differing_bytes2(x.scanId, y.scanId, (size_t)x.n_scan, cmp);
    return WOST_OK;
}

const char *wost_last_error(void) { return g_last_error.c_str(); }

int wost_create(const wost_scene_desc *scene, const wost_settings *settings, int device, wost_handle *out)
{
    if (!scene || !settings || !out) return fail(WOST_ERR_INVALID, "null argument");
    *out = nullptr;
    if (settings->width <= 0 || settings->height <= 0 || settings->spp < 0 || settings->max_depth <= 0)
        return fail(WOST_ERR_INVALID, "bad settings");
    if (settings->spp >= (1 << 20) || settings->max_depth >= (1 << 10))
        return fail(WOST_ERR_UNSUPPORTED, "spp must be < 2^20 and max_depth < 2^10");
    if ((int64_t)settings->width * settings->height > (1 << 28))
        return fail(WOST_ERR_UNSUPPORTED, "frame too large");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(WOST_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n_dev) return fail(WOST_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    wost_context *c = new (std::nothrow) wost_context();
    if (!c) return fail(WOST_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->settings = *settings;
    c->dst = DevSettings{settings->width, settings->height, settings->spp, settings->max_depth, settings->eps_shell,
                         scene->dirichlet_intensity, scene->neumann_intensity};
    c->probe = DevProbe{scene->probe_scale, scene->probe_pos[0], scene->probe_pos[1], scene->probe_up[0],
                        scene->probe_up[1]};
    c->n_pixels = (size_t)settings->width * settings->height;
    int rc = upload_mesh(scene->dirichlet, c->dm);
    if (rc == WOST_OK) rc = upload_mesh(scene->neumann, c->nm);
    auto bail = [&](int code) {
        destroy_ctx(c);
        return code;
    };
    if (rc != WOST_OK) return bail(rc);
#define HIP_TRY_C(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            fail(WOST_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));                   \
            return bail(WOST_ERR_DEVICE);                                                                \
        }                                                                                                \
    } while (0)
    if (scene->mask) {
        HIP_TRY_C(hipMalloc((void **)&c->mask, c->n_pixels));
        HIP_TRY_C(hipMemcpy(c->mask, scene->mask, c->n_pixels, hipMemcpyHostToDevice));
    }
    if (scene->source.nx > 0 && scene->source.ny > 0) {
        const wost_source_desc &sd = scene->source;
        if (!sd.rgb) {
            fail(WOST_ERR_INVALID, "source grid without data");
            return bail(WOST_ERR_INVALID);
        }
        const size_t bytes = (size_t)sd.nx * sd.ny * 3 * sizeof(float);
        float *dev = nullptr;
        HIP_TRY_C(hipMalloc((void **)&dev, bytes));
        c->src = DevSource{dev, sd.nx, sd.ny, sd.index_scale[0], sd.index_scale[1], sd.index_offset[0], sd.index_offset[1],
                           sd.intensity};
        HIP_TRY_C(hipMemcpy(dev, sd.rgb, bytes, hipMemcpyHostToDevice));
    }
    const size_t qbytes = c->n_pixels * 4 * kQueueWords;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY_C(hipMalloc(&c->queue_mem[i], qbytes));
        carve_queue(c->queue_mem[i], c->n_pixels, c->queue[i]);
    }
    HIP_TRY_C(hipMalloc((void **)&c->counts, 12 * sizeof(uint32_t)));   // two queue counts, the far count and its input copy; [4..7]: the hand-over of a persistent launch
    HIP_TRY_C(hipMalloc((void **)&c->stats, kStatCopies * sizeof(StatsDev)));
    HIP_TRY_C(hipMalloc((void **)&c->cursor, sizeof(uint32_t)));
    HIP_TRY_C(hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device));
    HIP_TRY_C(hipMalloc((void **)&c->field, c->n_pixels * 3 * sizeof(float)));
    HIP_TRY_C(hipHostMalloc((void **)&c->host_count, 8 * sizeof(uint32_t)));
    HIP_TRY_C(hipHostMalloc((void **)&c->host_stats, kStatCopies * sizeof(StatsDev)));
    HIP_TRY_C(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY_C(hipStreamCreateWithFlags(&c->far_stream, hipStreamNonBlocking));
    HIP_TRY_C(hipEventCreateWithFlags(&c->far_ev0, hipEventDisableTiming));
    HIP_TRY_C(hipEventCreateWithFlags(&c->far_ev1, hipEventDisableTiming));
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        HIP_TRY_C(hipStreamCreateWithPriority(&c->long_stream, hipStreamNonBlocking, greatest));
    }
    HIP_TRY_C(hipEventCreateWithFlags(&c->long_ev0, hipEventDisableTiming));
    HIP_TRY_C(hipEventCreateWithFlags(&c->long_ev1, hipEventDisableTiming));
    HIP_TRY_C(hipEventCreate(&c->ev0));
    HIP_TRY_C(hipEventCreate(&c->ev1));
    {
        // the order of a persistent / one-launch solve: its buffers and the first use of its sort kernels (their code objects load
        // then: 2.5 ms) belong here, beside the tree builds, not inside the first solve -- time-to-1spp of a fresh handle
        HIP_TRY_C((hipError_t)order_alloc(c->order, c->n_pixels));
        const uint32_t n_warm = (uint32_t)std::min<size_t>(c->n_pixels, 1024);
        const uint32_t *unused = nullptr;
        HIP_TRY_C(hipMemsetAsync(c->queue[0].d0_d2, 0, n_warm * sizeof(float), c->stream));
        HIP_TRY_C(hipMemsetAsync(c->queue[0].est, 0, n_warm * sizeof(float), c->stream));
        HIP_TRY_C((hipError_t)order_by_distance(c->order, c->queue[0].d0_d2, n_warm, c->stream, &unused));
        HIP_TRY_C((hipError_t)order_by_estimate(c->order, c->queue[0].est, n_warm, c->stream, &unused));
        HIP_TRY_C(hipStreamSynchronize(c->stream));
    }
#undef HIP_TRY_C
    *out = c;
    return WOST_OK;
}

int wost_destroy(wost_handle h)
{
    destroy_ctx(h);
    return WOST_OK;
}

int wost_set_option(wost_handle h, const char *key, double value)
{
    if (!h || !key) return fail(WOST_ERR_INVALID, "null argument");
    const std::string k(key);
    if (k == "steps_per_round") {
        if (value < 0 || value > 32767) return fail(WOST_ERR_INVALID, "steps_per_round must be in 0..32767 (0 = automatic)");
        h->steps_per_round = (int)value;
    } else if (k == "block_size") {
        const int b = (int)value;
        if (b != 64 && b != 128 && b != 256) return fail(WOST_ERR_INVALID, "block_size must be 64, 128 or 256");
        h->block_size = b;
    } else if (k == "wait_weight") {
        if (value < 1 || value > 512) return fail(WOST_ERR_INVALID, "wait_weight must be in 1..512");
        h->wait_weight = (int)value;
        h->wait_weight_set = true;
    } else if (k == "trav_burst") {
        if (value < 1 || value > 16) return fail(WOST_ERR_INVALID, "trav_burst must be in 1..16");
        h->trav_burst = (int)value;
        h->trav_burst_set = true;
    } else if (k == "spp") {
        if (value < 0 || value >= (1 << 20)) return fail(WOST_ERR_INVALID, "spp must be in 0..2^20-1");
        h->settings.spp = (int32_t)value;
        h->dst.spp = (int32_t)value;
    } else if (k == "refill") {
        if (value != -1 && value != 0 && value != 1) return fail(WOST_ERR_INVALID, "refill must be -1 (auto), 0 or 1");
        h->refill = (int)value;
    } else if (k == "persist") {
        if (value != -1 && value != 0 && value != 1) return fail(WOST_ERR_INVALID, "persist must be -1 (auto), 0 or 1");
        h->persist = (int)value;
    } else if (k == "persist_order") {
        h->persist_order = value != 0;
    } else if (k == "few_order") {
        h->few_order = value != 0;
    } else if (k == "long_steps") {
        if (value < 0 || value > 65528 || ((int)value & 7)) return fail(WOST_ERR_INVALID, "long_steps must be a multiple of 8 in 0..65528 (0 = no pixel runs beside the rounds)");
        h->long_steps = (int)value;
    } else if (k == "long_cap") {
        if (value < 1 || value > (1 << 20)) return fail(WOST_ERR_INVALID, "long_cap must be in 1..2^20");
        h->long_cap = (int)value;
        if (h->long_mem) (void)hipFree(h->long_mem);
        h->long_mem = nullptr;
    } else if (k == "long_thin") {
        if (value < 0 || value > (1 << 20)) return fail(WOST_ERR_INVALID, "long_thin must be in 0..2^20");
        h->long_thin = (int)value;
    } else if (k == "tail_sort") {
        h->tail_sort = value != 0;
    } else if (k == "resident_blocks") {
        if (value < 0 || value > 65535) return fail(WOST_ERR_INVALID, "resident_blocks must be in 0..65535 (0 = what the chip holds)");
        h->resident_blocks = (int)value;
    } else if (k == "quad") {
        if (value != -1 && value != 0 && value != 1) return fail(WOST_ERR_INVALID, "quad must be -1 (auto), 0 or 1");
        h->quad = (int)value;
    } else if (k == "quad_fill") {
        if (!(value > 0) || value > 64) return fail(WOST_ERR_INVALID, "quad_fill must be in (0, 64]");
        h->quad_fill = value;
    } else if (k == "coop") {
        h->coop = value != 0;
    } else if (k == "pool_cap") {
        if (value != 0 && (value < 96 || value > 2048)) return fail(WOST_ERR_INVALID, "pool_cap must be 0 (automatic) or in 96..2048");
        h->pool_cap = (int)value;
    } else if (k == "ray_slot_trigger") {
        if (value < 1 || value > 64) return fail(WOST_ERR_INVALID, "ray_slot_trigger must be in 1..64");
        h->ray_slot_trigger = (int)value;
    } else if (k == "thin_waves") {
        h->thin_waves = value != 0;
    } else if (k == "time_kernels") {
        h->time_kernels = value != 0;
    } else {
        return fail(WOST_ERR_INVALID, "unknown option: " + k);
    }
    return WOST_OK;
}

}  // extern "C"

// the instantiation of the round kernel for a launch
template <bool SLACK>
static void launch_round(bool has_src, bool refill, bool ntree, bool emissive, unsigned grid, int bs, size_t lds_round, hipStream_t stream,
                         const RoundParams &rp, bool persist = false)
{
    if (!SLACK && persist && refill) {
#define WOST_PERSIST_CASE(E, T, S)                                                                                                           \
    if (emissive == E && ntree == T && has_src == S) {                                                                                       \
        hipLaunchKernelGGL((walk_round_kernel<E, T, true, S, false, true>), dim3(grid), dim3(bs), lds_round, stream, rp);                    \
        return;                                                                                                                              \
    }
        WOST_PERSIST_CASE(false, false, false) WOST_PERSIST_CASE(true, false, false) WOST_PERSIST_CASE(false, true, false) WOST_PERSIST_CASE(true, true, false)
        WOST_PERSIST_CASE(false, false, true) WOST_PERSIST_CASE(true, false, true) WOST_PERSIST_CASE(false, true, true) WOST_PERSIST_CASE(true, true, true)
#undef WOST_PERSIST_CASE
    }
    if (has_src) {
        // problems with a source term: the SOURCE instantiations (one extra stage per step)
        if (ntree) {
            if (emissive) hipLaunchKernelGGL((walk_round_kernel<true, true, false, true, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
            else hipLaunchKernelGGL((walk_round_kernel<false, true, false, true, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
        } else {
            if (emissive) hipLaunchKernelGGL((walk_round_kernel<true, false, false, true, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
            else hipLaunchKernelGGL((walk_round_kernel<false, false, false, true, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
        }
    } else if (refill) {
        if (ntree) {
            if (emissive) hipLaunchKernelGGL((walk_round_kernel<true, true, true, false, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
            else hipLaunchKernelGGL((walk_round_kernel<false, true, true, false, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
        } else {
            if (emissive) hipLaunchKernelGGL((walk_round_kernel<true, false, true, false, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
            else hipLaunchKernelGGL((walk_round_kernel<false, false, true, false, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
        }
    } else if (ntree) {
        if (emissive) hipLaunchKernelGGL((walk_round_kernel<true, true, false, false, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
        else hipLaunchKernelGGL((walk_round_kernel<false, true, false, false, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
    } else {
        if (emissive) hipLaunchKernelGGL((walk_round_kernel<true, false, false, false, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
        else hipLaunchKernelGGL((walk_round_kernel<false, false, false, false, SLACK>), dim3(grid), dim3(bs), lds_round, stream, rp);
    }
}

// the quad instantiation (four lanes per walker) for an ordinary launch
template <bool SLACK = false>
static void launch_quad(bool has_src, bool ntree, bool emissive, unsigned grid, int bs, size_t lds, hipStream_t stream, const RoundParams &rp)
{
#define WOST_QUAD_CASE(E, T, S)                                                                                                  \
    if (emissive == E && ntree == T && has_src == S) {                                                                           \
        hipLaunchKernelGGL((walk_quad_kernel<E, T, S, SLACK>), dim3(grid), dim3(bs), lds, stream, rp);                          \
        return;                                                                                                                  \
    }
    WOST_QUAD_CASE(false, false, false) WOST_QUAD_CASE(true, false, false) WOST_QUAD_CASE(false, true, false) WOST_QUAD_CASE(true, true, false)
    WOST_QUAD_CASE(false, false, true) WOST_QUAD_CASE(true, false, true) WOST_QUAD_CASE(false, true, true) WOST_QUAD_CASE(true, true, true)
#undef WOST_QUAD_CASE
}

static WalkQueue queue_from(const WalkQueue &q, size_t k)
{
    WalkQueue r = q;
    r.pix += k; r.x0 += k; r.y0 += k; r.px += k; r.py += k; r.rng += k; r.meta += k; r.nx += k; r.ny += k; r.hint += k; r.thp += k;
    r.sr += k; r.sg += k; r.sb += k; r.d0_d2 += k; r.d0_slot += k; r.est += k;
    return r;
}

// the shared solve driver: field_dev indexed by (pix - field_base)
static int run_solve(wost_context *c, int32_t pixel_begin, int32_t pixel_end, int32_t shard_index, int32_t shard_count,
                     float *field_dev, int32_t field_base, hipStream_t stream, wost_stats *stats)
{
    const auto t_start = std::chrono::high_resolution_clock::now();
    HIP_TRY(hipSetDevice(c->device));
    const int bs = c->block_size;
    const int levels = c->dm.view.n_segs > 0 ? c->dm.view.levels : 1;
    const int levels_any = std::max(levels, c->nm.view.n_segs > 0 ? c->nm.view.levels : 1);
    // closest point: pushes happen on the inner levels 0..L-1, at most 3 per level, and the
    // branch-free pushes never write beyond entry 3L-1; the Neumann tree queries push up to 4
    // and pop 1 per inner level: 3L+1 entries
    const int stack_depth = 3 * levels_any + 1;
    const size_t lds = (size_t)stack_depth * bs * sizeof(uint32_t);
    HIP_TRY(hipMemsetAsync(c->counts, 0, 12 * sizeof(uint32_t), stream));
    HIP_TRY(hipMemsetAsync(c->stats, 0, kStatCopies * sizeof(StatsDev), stream));

    const int tiles_x = (c->settings.width + 7) / 8, tiles_y = (c->settings.height + 7) / 8;
    InitParams ip{};
    ip.dm = c->dm.view;
    ip.st = c->dst;
    ip.probe = c->probe;
    ip.out = c->queue[0];
    ip.count_out = c->counts + 0;
    ip.mask = c->mask;
    ip.field = field_dev;
    ip.field_base = field_base;
    ip.pixel_begin = pixel_begin;
    ip.pixel_end = pixel_end;
    ip.shard_index = shard_index;
    ip.shard_count = shard_count;
    ip.tiles_x = tiles_x;
    ip.tiles_y = tiles_y;
    ip.stack_stride = bs;
    const long long n_threads = (long long)tiles_x * tiles_y * 64;
    const unsigned init_grid = (unsigned)((n_threads + bs - 1) / bs);
    hipLaunchKernelGGL(init_kernel, dim3(init_grid), dim3(bs), lds, stream, ip);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->host_count, c->counts, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    uint32_t n_active = c->host_count[0];

    double kernel_ms = 0.0;
    uint32_t launches = 0;
    int cur = 0;
    c->last_launches.clear();
    const bool emissive = c->nm.view.n_segs > 0 && c->nm.view.emissive;
    const bool ntree = c->nm.view.n_segs > WOST_FLAT_MAX;
    uint32_t pending_far = 0;     // walkers at the far end of queue[cur] that the previous launch could not serve (see below)
    bool handed_over = false;     // the previous launch was persistent: queue[cur] holds what it handed over, with estimates
    bool long_pending = false;    // the launch of the long remainders is under way on long_stream
    while (n_active > 0 || pending_far > 0) {
        const int nxt = cur ^ 1;
        HIP_TRY(hipMemsetAsync(c->counts + nxt, 0, sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(c->counts + 2, 0, sizeof(uint32_t), stream));
        RoundParams rp{};
        rp.count_long = c->counts + 4;
        rp.long_steps = c->long_steps > 0 ? (float)c->long_steps : WOST_INF;
        rp.dm = c->dm.view;
        rp.nm = c->nm.view;
        rp.st = c->dst;
        rp.probe = c->probe;
        rp.src = c->src;
        rp.in = c->queue[cur];
        rp.out = c->queue[nxt];
        rp.count_in = c->counts + cur;
        rp.count_out = c->counts + nxt;
        rp.field = field_dev;
        rp.field_base = field_base;
        rp.stats = c->stats;
        rp.count_far = c->counts + 2;
        rp.out_capacity = (uint32_t)c->n_pixels;
        // a slot regenerates its pixel's next sample inside a round, so rounds are long: 256 steps
        // unless the caller chose otherwise (shorter rounds only added launches: 128^2 at 1 spp
        // 0.95 -> 1.39 ms with 8-step rounds)
        rp.steps_per_round = c->steps_per_round > 0 ? c->steps_per_round : 256;
        rp.stack_stride = bs;
        // a step that answers its Neumann queries on the tree is long and divergent: served when eight ninths of the
        // busy lanes wait (tools/probes/bench2d_wiggly.py: 3.87 -> 4.40 x 10^8 walk-steps/s on 3000 segments)
        rp.wait_weight = (ntree && !c->wait_weight_set) ? 1 : c->wait_weight;
        // bursts of five visits unless the caller chose (round 6: + 1-2 % in rounds and shards as well as in the persistent launch,
        // profiles/r06_v_*; a Neumann mesh on the tree keeps the three its step scheduling was tuned with)
        rp.trav_burst = (c->trav_burst_set || ntree) ? c->trav_burst : 5;
        // a Neumann mesh on the tree: the task pools of every wave behind the stack columns (wost_coop.h)
        rp.coop = (ntree && c->coop && c->nm.view.levels <= 11) ? 1 : 0;
        rp.pool_cap = c->pool_cap > 0 ? c->pool_cap : std::min(1024, std::max(384, 128 * c->nm.view.levels));
        rp.pool_offset = (int32_t)(lds / sizeof(uint32_t));
        rp.ray_slot_trigger = c->ray_slot_trigger;
        auto pools_bytes = [&](int cap) { return (size_t)(bs / 64) * (2 * (size_t)cap + kPoolOwnerWords) * sizeof(uint32_t) + 8; };
        // (the automatic size gives way to the stack columns of a deep Dirichlet tree; with no room at all: one descent per lane)
        while (c->pool_cap <= 0 && rp.pool_cap > 256 && lds + pools_bytes(rp.pool_cap) > 64 * 1024) rp.pool_cap -= 64;
        size_t lds_pools = rp.coop ? pools_bytes(rp.pool_cap) : 0;
        if (lds + lds_pools > 64 * 1024) {
            rp.coop = 0;
            lds_pools = 0;
        }
        // developer experiment: extra LDS per block lowers the number of resident blocks (occupancy sensitivity)
        const size_t lds_round = lds + lds_pools + (getenv("WOST_EXP_LDS_PAD") ? (size_t)atoi(getenv("WOST_EXP_LDS_PAD")) : 0);
        const bool has_src = c->src.rgb != nullptr;
        // blocks a CU holds: the register bound of the instantiation (launch bounds: 6 waves per SIMD, 4 with the tree queries), and
        // no more than fit its 160 KB of LDS -- with the wave task pools a block asks for about 64 KB (two per CU, not four)
        const unsigned blocks_by_regs = (unsigned)((ntree ? 4 : 6) * 4 * 64 / bs);
        const unsigned blocks_by_lds = (unsigned)std::max<size_t>(1, (size_t)160 * 1024 / std::max<size_t>(lds_round, 1));
        const unsigned blocks_per_cu = std::max(1u, std::min(blocks_by_regs, blocks_by_lds));
        const unsigned resident_threads = (unsigned)c->n_cus * blocks_per_cu * (unsigned)bs;
        if (c->time_kernels) HIP_TRY(hipEventRecord(c->ev0, stream));
        // What a persistent launch handed over: one partly solved pixel per lane, each with an estimate of the walk steps it still
        // needs (config 2: 387 000 pixels, 12 % of the solve's steps; half of them need 650 steps more, a few thousand over 1 500),
        // and the walkers that strayed beyond the plain visits' range during that launch, parked since with most of their samples
        // ahead of them.  In rounds the few long ones add launch after launch of a nearly empty chip (16 of the 52 ms that followed
        // the persistent launch), so they start NOW and run to their end -- four lanes to a walker, exact at any distance (SLACK: no
        // walker leaves such a launch), the longest a thousand or two four to a wave, the others sixteen -- on streams of their own
        // beside the rounds of the rest; and the first round takes the rest sorted by estimate: the walkers of a wave finish
        // together and the wave leaves, instead of a third of the lanes of every wave idling behind finished pixels.
        uint32_t n_round = n_active;        // walkers of this iteration's ordinary launch
        uint32_t beside_far = 0;            // strayed walkers that joined the long remainders
        if (handed_over && (n_active > 0 || pending_far > 0)) {
            const bool beside = c->long_steps > 0;
            const uint32_t n_far = (beside && pending_far <= 1024u && pending_far <= (uint32_t)c->long_cap) ? pending_far : 0u;
            const uint32_t n_long = beside ? std::min<uint32_t>(std::min<uint32_t>(c->host_count[4], (uint32_t)c->long_cap - n_far), n_active) : 0u;
            const uint32_t *ord = nullptr;
            if (n_active > 0 && (n_long > 0 || c->tail_sort)) {
                if (c->order.cap < c->n_pixels) HIP_TRY((hipError_t)order_alloc(c->order, c->n_pixels));
                HIP_TRY((hipError_t)order_by_estimate(c->order, c->queue[cur].est, n_active, stream, &ord));
                rp.order = ord + n_long;
            }
            const uint32_t n_beside = n_far + n_long;
            if (n_beside > 0) {
                if (!c->long_mem) {
                    HIP_TRY(hipMalloc(&c->long_mem, (size_t)c->long_cap * 4 * kQueueWords));
                    carve_queue(c->long_mem, (size_t)c->long_cap, c->long_queue);
                }
                // the strayed first (most of a pixel ahead of them as a rule), then the long remainders, longest first
                if (n_far > 0) {
                    hipLaunchKernelGGL(gather_walkers_kernel, dim3((n_far + 255) / 256), dim3(256), 0, stream, queue_from(c->queue[cur], c->n_pixels - n_far),
                                       (const uint32_t *)nullptr, n_far, c->long_queue);
                    HIP_TRY(hipGetLastError());
                    pending_far = 0;
                    beside_far = n_far;
                }
                if (n_long > 0) {
                    hipLaunchKernelGGL(gather_walkers_kernel, dim3((n_long + 255) / 256), dim3(256), 0, stream, c->queue[cur], ord, n_long, queue_from(c->long_queue, n_far));
                    HIP_TRY(hipGetLastError());
                }
                const uint32_t n_thin = std::min<uint32_t>(n_beside, (uint32_t)std::max(c->long_thin, 0));
                c->host_count[5] = n_beside;
                c->host_count[6] = n_active - n_long;
                HIP_TRY(hipMemcpyAsync(c->counts + 5, c->host_count + 5, 2 * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                HIP_TRY(hipEventRecord(c->long_ev0, stream));
                HIP_TRY(hipStreamWaitEvent(c->long_stream, c->long_ev0, 0));
                RoundParams lp = rp;
                lp.in = c->long_queue;
                lp.order = nullptr;
                lp.count_in = c->counts + 5;
                lp.count_out = c->counts + 8;       // (nothing comes out: every walker runs to the end of its pixel)
                lp.count_far = c->counts + 8;
                lp.steps_per_round = 0x7fffffff;
                lp.stack_stride = stack_depth;
                lp.lane_shift = 0;
                const uint32_t quads_per_block = (uint32_t)bs / 4u;
                lp.thin_count = n_thin;
                lp.thin_blocks = (n_thin * 4u + quads_per_block - 1u) / quads_per_block;
                const unsigned lgrid = lp.thin_blocks + (n_beside - n_thin + quads_per_block - 1u) / quads_per_block;
                if (lp.thin_blocks == 0u) lp.thin_count = 0u;
                launch_quad<true>(has_src, ntree, emissive, lgrid, bs, (size_t)stack_depth * (bs / 4) * sizeof(uint32_t), c->long_stream, lp);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(c->long_ev1, c->long_stream));
                long_pending = true;
                rp.count_in = c->counts + 6;
                n_round = n_active - n_long;
            }
        }
        // Walkers that left the previous launch at a query beyond the plain kernel's range wait at the far end of its output queue,
        // which is this launch's input queue.  The SLACK instantiation takes them through max_depth steps -- the walk that strayed
        // ends within that many -- on a stream of its own, next to this launch, and appends them to the same output queue (both
        // kernels only read the input queue and claim output slots from one counter).
        const uint32_t far_now = pending_far;
        if (far_now > 0) {
            c->host_count[3] = far_now;
            HIP_TRY(hipMemcpyAsync(c->counts + 3, c->host_count + 3, sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipEventRecord(c->far_ev0, stream));                      // the counters are ready
            HIP_TRY(hipStreamWaitEvent(c->far_stream, c->far_ev0, 0));
            RoundParams fp = rp;
            fp.order = nullptr;
            fp.in = queue_from(c->queue[cur], c->n_pixels - far_now);
            fp.count_in = c->counts + 3;
            fp.steps_per_round = std::max(1, c->settings.max_depth);
            // a few strayed walkers (leaks of a closed scene): one per wave -- a query from very far away is a scan of the whole
            // mesh by the 64 lanes (closest_point_wave), and the launch lasts as long as its slowest wave; many (an open scene:
            // every walk that misses the boundary strays): every lane loaded
            fp.lane_shift = far_now <= 4096u ? 6 : 0;
            launch_round<true>(has_src, false, ntree, emissive, (unsigned)((((uint64_t)far_now << fp.lane_shift) + bs - 1) / bs), bs, lds_round, c->far_stream, fp);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(c->far_ev1, c->far_stream));
            // (not counted in `launches`: kernel_ms / kernel_launches stays the average duration of the ordinary launches)
        }
        // When the walkers left fill less than 1/16 of the resident threads, spread them out: the
        // duration of such a launch is the latency of its slowest wave, and a wave is as slow as the
        // longest query among its walkers (config 2's last three launches: 13.1 -> 8.9 ms; with 2 or
        // 4 lanes per walker the extra waves cost more than they save: 9.2 -> 11.8 ms).
        rp.lane_shift = 0;
        if (c->thin_waves) {
            while (rp.lane_shift < 6 && ((uint64_t)n_round << (rp.lane_shift + 1)) <= resident_threads) ++rp.lane_shift;
            if (rp.lane_shift < 4) rp.lane_shift = 0;
        }
        unsigned grid = (unsigned)((((uint64_t)n_round << rp.lane_shift) + bs - 1) / bs);
        // REFILL launch: as many resident threads as the chip holds, each draining the input queue.
        // Worth it when regeneration cannot keep the lanes busy (measured on config 2's frame:
        // 1 spp 3.5 -> 3.0 ms, 4 spp 8.3 -> 7.9 ms, 8 spp 13.0 -> 13.5 ms) and the queue is larger
        // than one residency; the 16-bit lane counters bound spp * max_depth.
        const unsigned resident = c->resident_blocks > 0 ? std::min((unsigned)c->resident_blocks, (unsigned)c->n_cus * blocks_per_cu) : (unsigned)c->n_cus * blocks_per_cu;
        // (the one-launch form of few samples has no instantiation with a source term; the persistent launch has)
        const bool can_persist = (int64_t)c->settings.spp * c->settings.max_depth < 65535 && launches == 0;
        const bool can_refill = can_persist && !has_src;
        const bool few = can_refill && (c->refill == 1 || (c->refill == -1 && c->settings.spp <= 4 && grid > resident));
        // PERSISTENT first launch (many samples per pixel, more walkers than resident lanes): the same resident threads, but the
        // lanes take whole pixels -- all of a pixel's samples, the pixels in the order of wost_order.h, longest expected chain first
        // -- until the input queue is dry; then every wave hands what it holds to the output queue and the rest of the solve runs
        // in rounds.  No lane idles behind a finished pixel and no launch ends while pixels are unread (in rounds, config 2 spent
        // 108 of its 252 ms in launches where a third of the lanes had finished their pixel, EXPERIMENTS 25); what the rounds
        // get is the remainder of one pixel per lane.
        const bool persist = can_persist && !few && c->refill != 1 &&
                             (c->persist == 1 || (c->persist == -1 && c->settings.spp > 4 && n_active > resident_threads));
        const bool refill = few || persist;
        if (refill) {
            rp.lane_shift = 0;
            grid = std::min((unsigned)((n_active + bs - 1) / bs), resident);
            c->host_count[1] = grid * (unsigned)bs;      // first unread slot (pinned staging word)
            HIP_TRY(hipMemcpyAsync(c->cursor, c->host_count + 1, sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            rp.cursor = c->cursor;
            rp.steps_per_round = 0x7fffffff;
            rp.reserve = 64;
            if (persist ? c->persist_order : c->few_order) {
                if (c->order.cap < c->n_pixels) HIP_TRY((hipError_t)order_alloc(c->order, c->n_pixels));
                HIP_TRY((hipError_t)order_by_distance(c->order, c->queue[cur].d0_d2, n_active, stream, &rp.order));
            }
            if (persist) {
                rp.reserve = 0;
                rp.leave_dry = 1;
                // every lane holds a live walker throughout: longer traversal bursts and an earlier step trip pay here (config 2
                // 222 -> 217 ms, config 3 327 -> 317, profiles/r06_i_burst_variants.txt), in rounds they do not
                if (!c->trav_burst_set) rp.trav_burst = 5;
                if (!c->wait_weight_set && !ntree) rp.wait_weight = 6;
            }
        }
        // Under-filled launch: four lanes per walker (walk_quad_kernel).  Such a launch lasts as long as its longest chain of
        // dependent node visits; sharing a descent between the lanes of a quad halves that chain.
        const bool quad = !refill && n_round > 0 &&
                          (c->quad == 1 || (c->quad == -1 && 4.0 * (double)n_round <= c->quad_fill * (double)resident_threads));
        // (the last strayed walkers can outlive the ordinary queue: then only their launch runs)
        if (quad) {
            // few walkers: fewer quads per wave (16 -> 4 -> 1), as long as the waves still fit the chip
            rp.lane_shift = 0;
            while (rp.lane_shift < 4 && ((uint64_t)n_round << (2 + rp.lane_shift + 2)) <= resident_threads) rp.lane_shift += 2;
            rp.stack_stride = stack_depth;      // entries per (contiguous) quad column
            grid = (unsigned)((((uint64_t)n_round << (2 + rp.lane_shift)) + bs - 1) / bs);
            launch_quad(has_src, ntree, emissive, grid, bs, (size_t)stack_depth * (bs / 4) * sizeof(uint32_t), stream, rp);
            HIP_TRY(hipGetLastError());
        } else if (n_round > 0) {
            launch_round<false>(has_src, refill, ntree, emissive, grid, bs, lds_round, stream, rp, persist);
            HIP_TRY(hipGetLastError());
        }
        if (far_now > 0) HIP_TRY(hipStreamWaitEvent(stream, c->far_ev1, 0));
        if (c->time_kernels) HIP_TRY(hipEventRecord(c->ev1, stream));
        HIP_TRY(hipMemcpyAsync(c->host_count, c->counts + nxt, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(c->host_count + 2, c->counts + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        if (persist) HIP_TRY(hipMemcpyAsync(c->host_count + 4, c->counts + 4, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        if (c->time_kernels) HIP_TRY(hipMemcpyAsync(c->host_stats, c->stats, kStatCopies * sizeof(StatsDev), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        handed_over = persist;
        if (c->time_kernels) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
            kernel_ms += ms;
            wost_launch_info li{};
            li.kind = persist ? WOST_LAUNCH_PERSISTENT : (refill ? WOST_LAUNCH_ONE : (quad ? WOST_LAUNCH_QUAD : WOST_LAUNCH_ROUND));
            li.walkers = n_round;
            li.walkers_beside = n_active - n_round + far_now + beside_far;
            li.grid = n_round > 0 ? grid : 0u;
            li.ms = ms;
            for (int k = 0; k < kStatCopies; ++k) li.walk_steps_done += c->host_stats[k].steps;
            c->last_launches.push_back(li);
            if (getenv("WOST_TRACE_LAUNCHES")) fprintf(stderr, "launch %d: walkers %u of %u (+ %u strayed) grid %u %.3f ms -> %u left, %u strayed%s\n", launches, n_round, n_active, far_now, grid, ms, c->host_count[0], c->host_count[2], persist ? " (persistent)" : "");
        }
        if (n_round > 0) ++launches;
        pending_far = c->host_count[2];
        n_active = c->host_count[0];
        cur = nxt;
    }
    if (long_pending) {
        // the long remainders may outlast the rounds: what is left of their launch counts as kernel time of the solve
        if (c->time_kernels) HIP_TRY(hipEventRecord(c->ev0, stream));
        HIP_TRY(hipStreamWaitEvent(stream, c->long_ev1, 0));
        if (c->time_kernels) {
            HIP_TRY(hipEventRecord(c->ev1, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
            kernel_ms += ms;
            wost_launch_info li{};
            li.kind = WOST_LAUNCH_WAIT;
            li.ms = ms;
            c->last_launches.push_back(li);
            if (getenv("WOST_TRACE_LAUNCHES")) fprintf(stderr, "waited %.3f ms for the launch of the long remainders\n", ms);
        }
    }
    std::vector<StatsDev> copies(kStatCopies);
    HIP_TRY(hipMemcpyAsync(copies.data(), c->stats, kStatCopies * sizeof(StatsDev), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    StatsDev sd{};
    for (const StatsDev &k : copies) {
        sd.steps += k.steps; sd.started += k.started; sd.absorbed += k.absorbed; sd.truncated += k.truncated;
        sd.nhits += k.nhits; sd.inner_visits += k.inner_visits; sd.leaf_visits += k.leaf_visits;
        sd.trav_trips += k.trav_trips; sd.step_trips += k.step_trips;
        sd.max_stack = std::max(sd.max_stack, k.max_stack);
        sd.sp_ge6 += k.sp_ge6; sd.sp_ge10 += k.sp_ge10; sd.sp_ge14 += k.sp_ge14;
    }
#ifdef WOST_TRACK
    fprintf(stderr, "WOST_TRACK: visits %llu leaf %llu max_stack %llu visits leaving >=6 / >=10 / >=14 entries: %llu / %llu / %llu\n", sd.inner_visits,
            sd.leaf_visits, sd.max_stack, sd.sp_ge6, sd.sp_ge10, sd.sp_ge14);
#endif
    if (stats) {
        const auto t_end = std::chrono::high_resolution_clock::now();
        stats->walk_steps = sd.steps;
        stats->walks_started = sd.started;
        stats->walks_absorbed = sd.absorbed;
        stats->walks_truncated = sd.truncated;
        stats->neumann_hits = sd.nhits;
        stats->inner_visits = sd.inner_visits;
        stats->leaf_visits = sd.leaf_visits;
        stats->trav_trips = sd.trav_trips;
        stats->step_trips = sd.step_trips;
        stats->reserved = (uint32_t)sd.max_stack;
        stats->kernel_ms = kernel_ms;
        stats->kernel_launches = launches;
        stats->solve_ms = std::chrono::duration<double, std::milli>(t_end - t_start).count();
    }
    return WOST_OK;
}

static DeviceMeshStorage *pick_mesh(wost_handle h, int which)
{
    if (which == WOST_MESH_DIRICHLET) return &h->dm;
    if (which == WOST_MESH_NEUMANN) return &h->nm;
    return nullptr;
}

// small RAII helper for scratch device buffers of the batch queries
struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch()
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

int wost_solve(wost_handle h, int32_t pixel_begin, int32_t pixel_end, float *field_rgb, wost_stats *stats)
{
    if (!h || !field_rgb) return fail(WOST_ERR_INVALID, "null argument");
    if (pixel_begin < 0 || pixel_end > (int64_t)h->n_pixels || pixel_begin > pixel_end)
        return fail(WOST_ERR_INVALID, "pixel range outside the frame");
    const auto t0 = std::chrono::high_resolution_clock::now();
    const size_t n = (size_t)(pixel_end - pixel_begin);
    if (n == 0) {
        if (stats) std::memset(stats, 0, sizeof(*stats));
        return WOST_OK;
    }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemsetAsync(h->field, 0, n * 3 * sizeof(float), h->stream));
    int rc = run_solve(h, pixel_begin, pixel_end, 0, 1, h->field, pixel_begin, h->stream, stats);
    if (rc != WOST_OK) return rc;
    HIP_TRY(hipMemcpyAsync(field_rgb, h->field, n * 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (stats)
        stats->solve_ms =
            std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    return WOST_OK;
}

int wost_solve_sharded(wost_handle h, int32_t shard_index, int32_t shard_count, float *field_rgb_dev, void *stream,
                       wost_stats *stats)
{
    if (!h || !field_rgb_dev) return fail(WOST_ERR_INVALID, "null argument");
    if (shard_count <= 0 || shard_index < 0 || shard_index >= shard_count)
        return fail(WOST_ERR_INVALID, "bad shard");
    // NULL is the legacy default stream, which is also torch's default stream
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return run_solve(h, 0, (int32_t)h->n_pixels, shard_index, shard_count, field_rgb_dev, 0, s, stats);
}

int wost_last_launches(wost_handle h, wost_launch_info *out, int32_t capacity, int32_t *count)
{
    if (!h || !count || (capacity > 0 && !out) || capacity < 0) return fail(WOST_ERR_INVALID, "null argument");
    *count = (int32_t)h->last_launches.size();
    for (int32_t i = 0; i < std::min(*count, capacity); ++i) out[i] = h->last_launches[(size_t)i];
    return WOST_OK;
}

int wost_render_sdf(wost_handle h, int which_mesh, float *out_dist)
{
    if (!h || !out_dist) return fail(WOST_ERR_INVALID, "null argument");
    DeviceMeshStorage *m = pick_mesh(h, which_mesh);
    if (!m) return fail(WOST_ERR_INVALID, "unknown mesh selector");
    HIP_TRY(hipSetDevice(h->device));
    const int bs = 256;
    const int levels = m->view.n_segs > 0 ? m->view.levels : 1;
    const size_t lds = (size_t)(3 * levels + 1) * bs * sizeof(uint32_t);
    const int n = (int)h->n_pixels;
    float *d_out = h->field;  // reuse: n_pixels floats fit in the field buffer
    hipLaunchKernelGGL(sdf_kernel, dim3((n + bs - 1) / bs), dim3(bs), lds, h->stream, m->view, h->probe,
                       h->settings.width, h->settings.height, which_mesh, d_out, bs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_dist, d_out, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WOST_OK;
}

int wost_render_source(wost_handle h, float *out_rgb)
{
    if (!h || !out_rgb) return fail(WOST_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    const int n = (int)h->n_pixels;
    Scratch sc;
    float *d = nullptr;
    HIP_TRY(sc.alloc(&d, (size_t)n * 3));
    hipLaunchKernelGGL(source_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->src, h->probe, h->settings.width,
                       h->settings.height, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_rgb, d, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WOST_OK;
}

int wost_closest_point(wost_handle h, int which_mesh, const float *pts, int32_t n, int32_t *out_idx, float *out_dist,
                       float *out_uv, int32_t *out_side)
{
    if (!h || !pts || n < 0) return fail(WOST_ERR_INVALID, "null argument");
    DeviceMeshStorage *m = pick_mesh(h, which_mesh);
    if (!m || m->view.n_segs == 0) return fail(WOST_ERR_INVALID, "mesh is empty or unknown");
    if (n == 0) return WOST_OK;
    HIP_TRY(hipSetDevice(h->device));
    Scratch s;
    float *d_pts, *d_dist, *d_uv;
    int32_t *d_idx, *d_side;
    HIP_TRY(s.alloc(&d_pts, (size_t)n * 2));
    HIP_TRY(s.alloc(&d_dist, n));
    HIP_TRY(s.alloc(&d_uv, n));
    HIP_TRY(s.alloc(&d_idx, n));
    HIP_TRY(s.alloc(&d_side, n));
    HIP_TRY(hipMemcpyAsync(d_pts, pts, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    const int bs = 256;
    const size_t lds = (size_t)(3 * m->view.levels + 1) * bs * sizeof(uint32_t);
    hipLaunchKernelGGL(closest_point_kernel, dim3((n + bs - 1) / bs), dim3(bs), lds, h->stream, m->view, d_pts, n,
                       d_idx, d_dist, d_uv, d_side, bs);
    HIP_TRY(hipGetLastError());
    if (out_idx) HIP_TRY(hipMemcpyAsync(out_idx, d_idx, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    if (out_dist) HIP_TRY(hipMemcpyAsync(out_dist, d_dist, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    if (out_uv) HIP_TRY(hipMemcpyAsync(out_uv, d_uv, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    if (out_side) HIP_TRY(hipMemcpyAsync(out_side, d_side, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WOST_OK;
}

int wost_closest_silhouette(wost_handle h, int which_mesh, const float *pts, const float *rmax, int32_t n,
                            float *out_dist)
{
    if (!h || !pts || !out_dist || n < 0) return fail(WOST_ERR_INVALID, "null argument");
    DeviceMeshStorage *m = pick_mesh(h, which_mesh);
    if (!m) return fail(WOST_ERR_INVALID, "unknown mesh selector");
    if (n == 0) return WOST_OK;
    HIP_TRY(hipSetDevice(h->device));
    Scratch s;
    float *d_pts, *d_rmax = nullptr, *d_out;
    HIP_TRY(s.alloc(&d_pts, (size_t)n * 2));
    HIP_TRY(s.alloc(&d_out, n));
    HIP_TRY(hipMemcpyAsync(d_pts, pts, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (rmax) {
        HIP_TRY(s.alloc(&d_rmax, n));
        HIP_TRY(hipMemcpyAsync(d_rmax, rmax, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
    const int bs = 256;
    const size_t lds = (size_t)(3 * (m->view.n_segs > 0 ? m->view.levels : 1) + 1) * bs * sizeof(uint32_t);
    hipLaunchKernelGGL(silhouette_kernel, dim3((n + bs - 1) / bs), dim3(bs), lds, h->stream, m->view, d_pts, d_rmax, n,
                       d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_dist, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WOST_OK;
}

int wost_ray_intersect(wost_handle h, int which_mesh, const float *origins, const float *dirs, const float *tmax,
                       int32_t n, int32_t *out_hit, float *out_t, int32_t *out_idx)
{
    if (!h || !origins || !dirs || !tmax || !out_hit || !out_t || !out_idx || n < 0)
        return fail(WOST_ERR_INVALID, "null argument");
    DeviceMeshStorage *m = pick_mesh(h, which_mesh);
    if (!m) return fail(WOST_ERR_INVALID, "unknown mesh selector");
    if (n == 0) return WOST_OK;
    HIP_TRY(hipSetDevice(h->device));
    Scratch s;
    float *d_o, *d_d, *d_tm, *d_t;
    int32_t *d_hit, *d_idx;
    HIP_TRY(s.alloc(&d_o, (size_t)n * 2));
    HIP_TRY(s.alloc(&d_d, (size_t)n * 2));
    HIP_TRY(s.alloc(&d_tm, n));
    HIP_TRY(s.alloc(&d_t, n));
    HIP_TRY(s.alloc(&d_hit, n));
    HIP_TRY(s.alloc(&d_idx, n));
    HIP_TRY(hipMemcpyAsync(d_o, origins, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_d, dirs, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_tm, tmax, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    const int bs = 256;
    const size_t lds = (size_t)(3 * (m->view.n_segs > 0 ? m->view.levels : 1) + 1) * bs * sizeof(uint32_t);
    hipLaunchKernelGGL(ray_kernel, dim3((n + bs - 1) / bs), dim3(bs), lds, h->stream, m->view, d_o, d_d, d_tm, n, d_hit,
                       d_t, d_idx);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_hit, d_hit, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(out_t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(out_idx, d_idx, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return WOST_OK;
}

}  // extern "C"
