// wost_guided.hip -- the GUIDED Walk-on-Stars integrator on MI355X (SURVEY.md 8a rows a21, a22,
// a25, a26, a27) behind the C-ABI (include/wost.h, wost_guided_*).  gfx950 only.
//
// Every pixel consults ONE guiding network that is retrained between samples (reference
// integrator/guided/integrator.cu:968-1094), and the reference issues ~10 kernels and 2 stream syncs
// per depth over 45-byte AoS-like work items.  Here a whole sample is ONE launch when the network has
// the reference's shape (guided_sample_kernel below: the uniform integrator's persistent scheduler with
// the network evaluated inside the wave, up to 64 samples per launch once training has ended).  For
// other networks, and as the comparison path (WOST_GUIDED_FUSED=0), one depth is three launches over
// a compact SoA queue:
//   separate_kernel  closest point on the Dirichlet LBVH (LDS stack, temporal hint), epsilon
//                    shell -> boundary colour into the pixel; else closest silhouette, R_B,
//                    Neumann sampling, normalised network input; ballot/popcount compaction
//                    (one atomic per wave) into the out-of-shell queue
//   net_forward      the guiding network on exactly the live entries (size read on the device)
//   sample_kernel    routing by the learned selection probability, mixture or uniform
//                    direction with one-sample MIS, boundary intersection, throughput, training
//                    record; in place -- the out-of-shell queue becomes the next depth's queue
//   tail_kernel      from depth >= maxGuidedDepth on nothing needs the network: every remaining
//                    walker runs to its end in registers, ONE launch per sample instead of two
//                    per depth
// Free choices of the reference that made it irreproducible are fixed: the training set is
// ordered by (pixel, record) through a prefix sum instead of by atomics, so two runs -- and the
// CPU restatement the tests compare against -- see the same batches.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/wost.h"
#include "wost_internal.h"
#include "wost_order.h"
#include "wost_net_device.h"
#include "wost_vmm_device.h"
#include "wost_walk.h"

namespace wost {

constexpr int kMaxTrainDepth = 4;     // reference parameters.h:7 (record slots per pixel)
constexpr int kRecFields = 12;        // sol rgb, pos xy, dir xy, pdf, thp, normal xy, onNeumann
constexpr uint32_t kDead = 0xffffffffu;
constexpr uint32_t kOnNeumann = 0x80000000u;

struct GQueue {
    uint32_t *pid;     // pixel id | kOnNeumann; kDead = entry dropped
    float *x, *y, *thp, *nx, *ny, *rb;
    int32_t *hint;     // slot of the closest Dirichlet segment of the previous step
};

// Counters of a solve.  Atomics on ONE cache line are served one after the other by its L2 channel
// (measured: about 120 M/s, 270 us for the 16 384 waves of one launch), so the counters exist in
// kStatCopies copies 256 bytes apart, a wave adds to the copy of its block, and the host sums them.
constexpr int kStatCopies = 64;
struct alignas(256) GStatsDev {
    unsigned long long steps, started, absorbed, truncated, nhits, guided, net_points;
};

// HIP-event pairs around selected launches, read back lazily: a pair is only waited for when its
// slot of the ring comes up again, so the host keeps running ahead of the GPU.
struct EventRing {
    std::vector<hipEvent_t> ev;     // 2 per slot
    size_t next = 0, pending = 0;
    double total_ms = 0.0;
    hipError_t init(size_t slots)
    {
        ev.resize(2 * slots, nullptr);
        for (hipEvent_t &e : ev) {
            hipError_t rc = hipEventCreate(&e);
            if (rc != hipSuccess) return rc;
        }
        return hipSuccess;
    }
    void destroy()
    {
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
        ev.clear();
    }
    void collect_one()
    {
        const size_t slots = ev.size() / 2, oldest = (next + slots - pending) % slots;
        float ms = 0.0f;
        if (hipEventSynchronize(ev[2 * oldest + 1]) == hipSuccess &&
            hipEventElapsedTime(&ms, ev[2 * oldest], ev[2 * oldest + 1]) == hipSuccess)
            total_ms += ms;
        --pending;
    }
    size_t begin(hipStream_t s)
    {
        const size_t slots = ev.size() / 2;
        if (pending == slots) collect_one();
        const size_t slot = next;
        (void)hipEventRecord(ev[2 * slot], s);
        return slot;
    }
    void end(size_t slot, hipStream_t s)
    {
        (void)hipEventRecord(ev[2 * slot + 1], s);
        next = (next + 1) % (ev.size() / 2);
        ++pending;
    }
    double drain()
    {
        while (pending) collect_one();
        const double t = total_ms;
        total_ms = 0.0;
        return t;
    }
};

struct GAabb {
    float minx, miny, maxx, maxy;   // scene.aabb: contains() test (Eigen AlignedBox semantics)
    float cx, cy, ex, ey;           // centre and extent of the box inflated by 0.5 % of its diagonal
};

__device__ __forceinline__ bool aabb_contains(const GAabb &b, float x, float y)
{
    return b.minx <= x && x <= b.maxx && b.miny <= y && y <= b.maxy;
}

// normalizeSpatialCoord (reference integrator/guided/train.h:149-155)
__device__ __forceinline__ void normalize_coord(const GAabb &b, float x, float y, float &ox, float &oy)
{
    ox = 0.5f + (x - b.cx) / b.ex;
    oy = 0.5f + (y - b.cy) / b.ey;
}

struct GParams {
    DevMesh dm, nm;
    DevSettings st;
    DevProbe probe;
    DevSource src;
    GAabb box;
    const uint8_t *mask;
    GQueue in, out;
    const uint32_t *count_in;
    uint32_t *count_out;
    uint64_t *rng;            // per pixel (inc == 1)
    float *sol;               // per pixel rgb
    uint32_t *cur_depth;      // per pixel: training records of the current sample
    float *rec;               // [slot][field][rec_ld]: record set j of pixel p at column j * n_pixels + p (one set unless a launch
    size_t rec_ld;            //   trains on several samples of a pixel: "train_group"); cur_depth is indexed by the same column
    int32_t *hint0;           // per pixel: closest slot of the evaluation point
    float *net_in;            // [2 * slot]
    const float *net_out;     // [33][net_ld]: output o of queue slot s at net_out[o * net_ld + s]
    size_t net_ld;
    GStatsDev *stats;
    int32_t n_pixels;
    int32_t depth;
    int32_t stack_stride;
    // guiding state of this sample
    int32_t training;         // trainState.enableTraining
    int32_t guiding;          // depth < maxGuidedDepth
    int32_t max_train_depth;
    uint32_t train_offset, train_stride;
    float uniform_fraction;
    int32_t first_sample;
    int32_t last_depth;
    int32_t shard_index, shard_count;   // this solve owns the 8x8 pixel tiles t with t % count == index
    // fused sample kernel only
    float *d0_d2;             // per pixel: squared distance of the evaluation point to the Dirichlet boundary (with hint0)
    uint32_t *cursor;         // next unread pixel slot of this launch
    int32_t max_guided_depth;
    int32_t stack_words;      // LDS words of a lane's traversal stack
    int32_t wait_weight, trav_burst;
    int32_t tail_chunk, tail_margin_pct;   // reservation size near the end of the launch; how near, in % of the launch's lanes
    int32_t n_samples;        // samples of every pixel in this launch (> 1: nothing is trained between them)
    uint32_t *pstate;         // n_samples > 1: per pixel, samples of this launch that have arrived << 16 | that are complete (zero at the start)
    int32_t d0_valid;         // d0_d2 holds the query of every evaluation point (after the first fused launch of a solve)
    const uint32_t *order;    // item k of a launch belongs to pixel order[k % n_pixels] (longest expected walk first, wost_order.h); nullptr: tile order
    unsigned long long *dbg;  // WOST_GUIDED_DEBUG: [0] first start, [1] first wave out of pixels, [2] last wave out of pixels, [3] end (100 MHz ticks)
};

// the network as the fused kernel needs it: half precision (image) or fp32 (frag32 / grid32)
struct FusedNet {
    const uint2 *image;       // half: MFMA fragments (n_frag entries of 8 bytes), then the grid entries
    uint32_t n_frag;
    uint32_t w_off4[4];
    const float *frag32, *grid32;   // fp32: MFMA fragments (n_mlp floats) and the grid of the inference weights
    uint32_t n_mlp;
    uint32_t w_off[4];
    float scale[8];
    uint32_t res[8], off[9];
};

// Per-pixel state that one lane leaves and ANOTHER lane -- another CU, another XCD -- may pick up inside the same launch (the
// samples of a pixel are handed from lane to lane, guided_sample_kernel): its loads and stores bypass the CU's vector L1, which
// no other CU's store ever refreshes (agent-scope relaxed = `sc1`), and the hand-over itself is an agent-scope atomic on the
// pixel's state word after the storing wave has drained its stores.
template <class T>
__device__ __forceinline__ T ld_px(const T *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ void st_px(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void add_px(float *s, float r, float g, float b)
{
    st_px(s, r + ld_px(s));
    st_px(s + 1, g + ld_px(s + 1));
    st_px(s + 2, b + ld_px(s + 2));
}

__device__ __forceinline__ bool is_training_pixel(const GParams &P, uint32_t pid)
{
    return P.training && ((pid - P.train_offset) % P.train_stride == 0u);
}

// `rp` = record column of the pixel: pid + (record set of the sample) * n_pixels
__device__ __forceinline__ float &rec_at(const GParams &P, int slot, int field, uint32_t rp)
{
    return P.rec[((size_t)slot * kRecFields + field) * P.rec_ld + rp];
}

// recordSolution / recordSourceContribution (reference guided.h:48-68): add to every record
// this walk has already created.  (The reference's inclusive variant also touches the slot of
// the record not yet created, which incrementDepth then wipes: no observable effect.)
__device__ __forceinline__ void record_solution(const GParams &P, uint32_t rp, float r, float g, float b)
{
    const uint32_t n = min(P.cur_depth[rp], (uint32_t)kMaxTrainDepth);
    for (uint32_t i = 0; i < n; ++i) {
        rec_at(P, i, 0, rp) = rec_at(P, i, 0, rp) + r;
        rec_at(P, i, 1, rp) = rec_at(P, i, 1, rp) + g;
        rec_at(P, i, 2, rp) = rec_at(P, i, 2, rp) + b;
    }
}

// wave-level compaction: returns the output slot of this lane (valid when `keep`)
__device__ __forceinline__ uint32_t wave_push(bool keep, uint32_t *counter)
{
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == 0 && bal) base = atomicAdd(counter, (uint32_t)__popcll(bal));
    base = __shfl(base, 0);
    return base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ void wave_count(bool pred, unsigned long long *counter)
{
    const unsigned long long bal = __ballot(pred);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(counter, (unsigned long long)__popcll(bal));
}

__device__ __forceinline__ GStatsDev *my_stats(GStatsDev *stats) { return stats + (blockIdx.x & (kStatCopies - 1)); }

// ---- start of a sample: every unmasked pixel queues its evaluation point ---------------------
// (reference prepareSolve :112-128 on the first sample, reset + generateEvaluationPoints
// :131-150 on every sample)
__global__ __launch_bounds__(256) void begin_sample_kernel(GParams P)
{
    // queue order = 8x8 pixel tiles (a wave starts with 64 walkers that are neighbours in BOTH
    // directions and visit the same nodes of the tree), row-major when the frame is not made of
    // whole tiles.  The order of the queue has no influence on any result.
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (((P.st.width | P.st.height) & 7) == 0 && p < P.n_pixels) {
        const int tiles_x = P.st.width >> 3, tile = p >> 6, in_tile = p & 63;
        p = ((tile / tiles_x) * 8 + (in_tile >> 3)) * P.st.width + (tile % tiles_x) * 8 + (in_tile & 7);
    }
    const bool in_frame = p < P.n_pixels;
    bool active = false;
    float x = 0, y = 0;
    if (in_frame) {
        if (P.first_sample) {
            Pcg rng;
            pcg_seed_pixel(rng, p, P.st.width);
            P.rng[p] = rng.state;
            P.sol[3 * (size_t)p] = 0.0f; P.sol[3 * (size_t)p + 1] = 0.0f; P.sol[3 * (size_t)p + 2] = 0.0f;
            P.hint0[p] = 0;
        }
        P.cur_depth[p] = 0;
        const int px = p % P.st.width, py = p / P.st.width;
        const int tile = (py >> 3) * ((P.st.width + 7) >> 3) + (px >> 3);
        active = (tile % P.shard_count) == P.shard_index && (P.mask == nullptr || P.mask[p] != 0);
        if (active) eval_point(P.probe, p % P.st.width, p / P.st.width, P.st.width, P.st.height, x, y);
    }
    const uint32_t s = block_push(active, P.count_out);
    wave_count(active, &my_stats(P.stats)->started);
    if (active) {
        P.out.pid[s] = (uint32_t)p;
        P.out.x[s] = x; P.out.y[s] = y;
        P.out.thp[s] = 1.0f;
        P.out.nx[s] = 0.0f; P.out.ny[s] = 0.0f;
        P.out.hint[s] = P.hint0[p];
    }
}

// ---- separateEvaluationPoint + handleBoundary + sampleSource + sampleNeumann ------------------
// (reference guided/integrator.cu:153-249, 252-274, 277-364, 367-494) for ONE walker at the start
// of a step.  Returns SEP_ABSORBED (colour added to the pixel and its records), SEP_DROPPED (no
// boundary at all) or SEP_KEEP with R_B set and the Neumann / source terms added.
enum { SEP_ABSORBED = 0, SEP_DROPPED = 1, SEP_KEEP = 2 };

// everything of a step after the closest-point query on the Dirichlet boundary (`cp`, ignored when
// that boundary is empty)
template <bool EMISSIVE, bool TREE, bool SOURCE>
__device__ __forceinline__ int separate_finish(const GParams &P, uint32_t pid, bool on_n, float x, float y, float thp, float nx,
                                               float ny, int depth, Closest cp, int32_t &hint, float &R_B, Pcg &rng,
                                               const LdsColumn &stk, uint32_t rofs = 0u)
{
    const bool train_px = is_training_pixel(P, pid);
    const float eps = P.st.eps;
    float R_D = WOST_INF;
    if (P.dm.n_segs > 0) {
        hint = cp.slot;
        if (depth == 0) st_px(P.hint0 + pid, (int32_t)cp.slot);
        const float4 a = P.dm.segA[cp.slot];
        const float inv = P.dm.segInv[cp.slot];
        const float wx = x - a.x, wy = y - a.y;
        const float uv = dot2(wx, wy, a.z, a.w) * inv;
        const float cr = cross2(a.z, a.w, wx, wy);
        const int side = (0.0f < cr) - (cr < 0.0f);
        R_D = sqrtf(cp.d2);
        if ((R_D < eps) && (uv > 0.0f && uv < 1.0f)) {
            float r, g, b;
            surface_color(P.dm.segCol + 12 * (size_t)cp.slot, side, uv, r, g, b);
            r *= P.st.dirichlet_intensity; g *= P.st.dirichlet_intensity; b *= P.st.dirichlet_intensity;
            r *= thp; g *= thp; b *= thp;
            add_px(P.sol + 3 * (size_t)pid, r, g, b);
            if (train_px) record_solution(P, pid + rofs, r, g, b);
            return SEP_ABSORBED;
        }
    }
    float R_N = WOST_INF;
    if (P.nm.n_segs > 0) R_N = closest_silhouette<TREE>(P.nm, x, y, R_D, stk);
    R_B = fmaxf(WOST_R_B_FLOOR, fminf(R_D, R_N));     // no 0.99 in the guided integrator (:238-239)
    if (isinf(R_B)) return SEP_DROPPED;               // no boundary at all: nothing to walk to
    if (SOURCE) {
        float cr, cg, cb;
        if (source_sample<TREE>(P.src, P.nm, eps, x, y, R_B, on_n, nx, ny, thp, rng, stk, cr, cg, cb)) {
            add_px(P.sol + 3 * (size_t)pid, cr, cg, cb);
            if (train_px) record_solution(P, pid + rofs, cr, cg, cb);
        }
    }
    if (P.nm.n_segs > 0) {
        float cr, cg, cb;
        if (neumann_sample<EMISSIVE, TREE>(P.nm, P.st.neumann_intensity, eps, x, y, R_B, on_n, nx, ny, thp, rng, stk, cr, cg, cb)) {
            add_px(P.sol + 3 * (size_t)pid, cr, cg, cb);
            if (train_px) record_solution(P, pid + rofs, cr, cg, cb);
        }
    }
    return SEP_KEEP;
}

template <bool EMISSIVE, bool TREE, bool SOURCE>
__device__ __forceinline__ int separate_step(const GParams &P, uint32_t pid, bool on_n, float x, float y, float thp, float nx,
                                             float ny, int depth, int32_t &hint, float &R_B, Pcg &rng, uint32_t *stack,
                                             const LdsColumn &stk)
{
    Closest cp{WOST_INF, -1};
    if (P.dm.n_segs > 0) cp = closest_point(P.dm, x, y, slot_candidate(P.dm, hint, x, y), stack, P.stack_stride);
    return separate_finish<EMISSIVE, TREE, SOURCE>(P, pid, on_n, x, y, thp, nx, ny, depth, cp, hint, R_B, rng, stk);
}

// incrementDepth (reference guided.h:21-46): the vertex BEFORE the step becomes a training record
__device__ __forceinline__ void record_vertex(const GParams &P, uint32_t rp, float x, float y, float dirx, float diry, float pdf,
                                              float thp, bool on_n, float nx, float ny)
{
    const uint32_t d = P.cur_depth[rp];
    if (d >= (uint32_t)kMaxTrainDepth) return;
    rec_at(P, d, 0, rp) = 0.0f; rec_at(P, d, 1, rp) = 0.0f; rec_at(P, d, 2, rp) = 0.0f;
    rec_at(P, d, 3, rp) = x; rec_at(P, d, 4, rp) = y;
    rec_at(P, d, 5, rp) = dirx; rec_at(P, d, 6, rp) = diry;
    rec_at(P, d, 7, rp) = pdf;
    rec_at(P, d, 8, rp) = thp;
    rec_at(P, d, 9, rp) = nx; rec_at(P, d, 10, rp) = ny;
    rec_at(P, d, 11, rp) = on_n ? 1.0f : 0.0f;
    P.cur_depth[rp] = d + 1;
}

template <bool EMISSIVE, bool TREE, bool SOURCE>
__global__ __launch_bounds__(256) void separate_kernel(GParams P)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
    const LdsColumn stk{stack, (uint32_t)P.stack_stride};
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_in = *P.count_in;
    uint32_t pidf = kDead;
    if (i < n_in) pidf = P.in.pid[i];
    const bool live = pidf != kDead;
    wave_count(live, &my_stats(P.stats)->steps);
    int status = SEP_DROPPED;
    float x = 0, y = 0, thp = 0, nx = 0, ny = 0, R_B = 0;
    int32_t hint = 0;
    const uint32_t pid = pidf & ~kOnNeumann;
    const bool on_n = (pidf & kOnNeumann) != 0u;
    if (live) {
        x = P.in.x[i]; y = P.in.y[i]; thp = P.in.thp[i]; nx = P.in.nx[i]; ny = P.in.ny[i];
        hint = P.in.hint[i];
        Pcg rng{P.rng[pid], 1};
        status = separate_step<EMISSIVE, TREE, SOURCE>(P, pid, on_n, x, y, thp, nx, ny, P.depth, hint, R_B, rng, stack, stk);
        P.rng[pid] = rng.state;
    }
    const bool keep = live && status == SEP_KEEP;
    wave_count(live && status == SEP_ABSORBED, &my_stats(P.stats)->absorbed);
    const uint32_t s = block_push(keep, P.count_out);
    if (keep) {
        P.out.pid[s] = pidf;
        P.out.x[s] = x; P.out.y[s] = y; P.out.thp[s] = thp;
        P.out.nx[s] = nx; P.out.ny[s] = ny;
        P.out.rb[s] = R_B;
        P.out.hint[s] = hint;
        float ix, iy;
        normalize_coord(P.box, x, y, ix, iy);
        P.net_in[2 * (size_t)s] = ix;
        P.net_in[2 * (size_t)s + 1] = iy;
    }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// ---- the unguided tail of a sample ------------------------------------------------------------
// From depth >= maxGuidedDepth on, the reference keeps issuing separate / oneStepWalk launches
// over a queue that shrinks to a few walkers (integrator.cu:1014,1027-1041): 40 % of the walk
// phase of config 4 went into launches whose duration is one walker's latency.  Nothing there
// needs the network, so every walker simply runs to its end in registers: one launch per sample
// for all remaining depths, the same per-pixel arithmetic in the same order.
template <bool EMISSIVE, bool TREE, bool SOURCE>
__global__ __launch_bounds__(256) void tail_kernel(GParams P)
{
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
    const LdsColumn stk{stack, (uint32_t)P.stack_stride};
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_in = *P.count_in;
    uint32_t pidf = kDead;
    if (i < n_in) pidf = P.in.pid[i];
    bool live = pidf != kDead;
    const uint32_t pid = pidf & ~kOnNeumann;
    uint32_t steps = 0, absorbed = 0, truncated = 0, hits = 0;
    if (live) {
        bool on_n = (pidf & kOnNeumann) != 0u;
        float x = P.in.x[i], y = P.in.y[i], thp = P.in.thp[i], nx = P.in.nx[i], ny = P.in.ny[i];
        int32_t hint = P.in.hint[i];
        Pcg rng{P.rng[pid], 1};
        const bool train_px = is_training_pixel(P, pid);
        for (int depth = P.depth; depth < P.st.max_depth; ++depth) {
            ++steps;
            float R_B = 0.0f;
            const int status = separate_step<EMISSIVE, TREE, SOURCE>(P, pid, on_n, x, y, thp, nx, ny, depth, hint, R_B, rng, stack, stk);
            if (status != SEP_KEEP) {
                absorbed = status == SEP_ABSORBED ? 1u : 0u;
                live = false;
                break;
            }
            // oneStepWalk (reference guided/integrator.cu:883-965)
            float dirx, diry, pdf, alpha, nxt_x, nxt_y, hnx, hny;
            uniform_direction(on_n, nx, ny, rng, dirx, diry, pdf, alpha);
            const bool hit = walk_advance<TREE>(P.nm, P.st.eps, x, y, R_B, on_n, nx, ny, dirx, diry, stk, nxt_x, nxt_y, hnx, hny);
            if (train_px && depth < P.max_train_depth) record_vertex(P, pid, x, y, dirx, diry, pdf, thp, on_n, nx, ny);
            thp = thp / pdf / alpha / WOST_2PI;
            x = nxt_x; y = nxt_y; on_n = hit; nx = hnx; ny = hny;
            hits += hit ? 1u : 0u;
        }
        if (live) truncated = 1u;       // still walking after the last depth
        P.rng[pid] = rng.state;
    }
    const uint32_t s_steps = wave_sum(steps), s_abs = wave_sum(absorbed), s_tr = wave_sum(truncated), s_hit = wave_sum(hits);
    if ((threadIdx.x & 63) == 0) {
        GStatsDev *st = my_stats(P.stats);
        if (s_steps) atomicAdd(&st->steps, (unsigned long long)s_steps);
        if (s_abs) atomicAdd(&st->absorbed, (unsigned long long)s_abs);
        if (s_tr) atomicAdd(&st->truncated, (unsigned long long)s_tr);
        if (s_hit) atomicAdd(&st->nhits, (unsigned long long)s_hit);
    }
}

// ---- handleOutShellPoint + handleGuidedSampling + handleUniformSampling / oneStepWalk --------
// (reference guided/integrator.cu:497-526, 782-880, 671-779, 883-965) for ONE walker whose R_B is known.
// raw(j) = output j of the network for this walker (only called when `guiding`).
struct SampleOut {
    bool dropped, guided_step, hit_n;
    float x, y, thp, nx, ny;
};

// The mixture of a walker, as sample_step uses it.  MixRegs: built by the walker's lane from the raw network outputs, in
// registers (the per-depth sample_kernel).  MixCol: prepared by the lanes of the network unit and left in the lane's LDS
// column (guided_sample_kernel).  Same arithmetic, same bits.
template <class RAW>
struct MixRegs {
    const RAW &raw;
    Vmm m;
    __device__ __forceinline__ explicit MixRegs(const RAW &r) : raw(r) {}
    __device__ __forceinline__ float logit() const { return raw(32); }
    __device__ __forceinline__ void prepare() { m.build(raw); }
    __device__ __forceinline__ void sample(Pcg &rng, float &ox, float &oy) const { m.sample(rng, ox, oy); }
    __device__ __forceinline__ void pdf_pair(float ax, float ay, float bx, float by, bool two, float &pa, float &pb) const { m.pdf_pair(ax, ay, bx, by, two, pa, pb); }
};
struct MixCol {
    const LdsColumn &col;
    __device__ __forceinline__ explicit MixCol(const LdsColumn &c) : col(c) {}
    __device__ __forceinline__ float logit() const { return vmm_col(col, 40); }
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ void sample(Pcg &rng, float &ox, float &oy) const { vmm_col_sample(col, rng, ox, oy); }
    __device__ __forceinline__ void pdf_pair(float ax, float ay, float bx, float by, bool two, float &pa, float &pb) const { vmm_col_pdf_pair(col, ax, ay, bx, by, two, pa, pb); }
};

template <bool TREE, class MIX>
__device__ __forceinline__ SampleOut sample_step(const GParams &P, uint32_t pid, bool on_n, float x, float y, float thp, float nx, float ny,
                                                 float R_B, int depth, bool guiding, Pcg &rng, MIX &m, const LdsColumn &stk, uint32_t rofs = 0u)
{
    SampleOut o{false, false, false, x, y, thp, nx, ny};
    const bool record = is_training_pixel(P, pid) && depth < P.max_train_depth;
    float dirx = 0, diry = 0, pdf = 1, alpha = 1;
    if (!guiding) {
        uniform_direction(on_n, nx, ny, rng, dirx, diry, pdf, alpha);
    } else {
        const float sel = 1 / (1.f + det_expf(-m.logit()));               // logistic (functors.h:182)
        const bool inside = aabb_contains(P.box, x, y);
        // the draw precedes the box test and is skipped for uniform fraction 0 (:518)
        bool to_guided = (P.uniform_fraction == 0.0f) || (pcg_next_float(rng) < sel);
        to_guided = to_guided && inside;
        o.dropped = to_guided && !(P.uniform_fraction < 1.0f);            // kernel never launched (:1031)
        // the mixture is needed by both branches inside the box: for the sample and its pdf,
        // or for the MIS weight of a uniform sample
        const bool use_vmm = inside && !o.dropped;
        if (use_vmm) m.prepare();
        float uniform_pdf = 0.0f;
        if (to_guided) {
            if (!o.dropped) {
                m.sample(rng, dirx, diry);
                uniform_pdf = on_n ? (float)(1.0 / 3.14159265358979323846) : 1.0f / WOST_2PI;
                alpha = on_n ? 0.5f : 1.0f;
                o.guided_step = true;
            }
        } else {
            uniform_direction(on_n, nx, ny, rng, dirx, diry, pdf, alpha);
            uniform_pdf = pdf;
        }
        if (use_vmm) {
            // on a Neumann boundary the density of the mirrored direction is added; a guided
            // sample pointing out of the domain is replaced by its mirror image afterwards
            const float dd = 2 * (dirx * nx + diry * ny);
            const float rx = dirx - dd * nx, ry = diry - dd * ny;
            float guided_pdf, mirrored_pdf;
            m.pdf_pair(dirx, diry, rx, ry, on_n, guided_pdf, mirrored_pdf);
            if (on_n) {
                guided_pdf += mirrored_pdf;
                if (to_guided && nx * dirx + ny * diry <= 0) { dirx = rx; diry = ry; }
            }
            pdf = sel * guided_pdf + (1.0f - sel) * uniform_pdf;
        }
    }
    if (!o.dropped) {
        float nxt_x, nxt_y, hnx, hny;
        o.hit_n = walk_advance<TREE>(P.nm, P.st.eps, x, y, R_B, on_n, nx, ny, dirx, diry, stk, nxt_x, nxt_y, hnx, hny);
        if (record) record_vertex(P, pid + rofs, x, y, dirx, diry, pdf, thp, on_n, nx, ny);
        o.x = nxt_x; o.y = nxt_y;
        o.thp = thp / pdf / alpha / WOST_2PI;
        o.nx = hnx; o.ny = hny;
    }
    return o;
}

// one launch per depth, in place on the queue
template <bool TREE>
__global__ __launch_bounds__(256) void sample_kernel(GParams P)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk{lds_stack + threadIdx.x, (uint32_t)P.stack_stride};
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_in = *P.count_in;
    const bool live = i < n_in;       // separate_kernel only pushes live entries
    bool guided_step = false, hit_n = false, alive_after = false;
    if (live) {
        const uint32_t pidf = P.in.pid[i];
        const uint32_t pid = pidf & ~kOnNeumann;
        const bool on_n = (pidf & kOnNeumann) != 0u;
        const float x = P.in.x[i], y = P.in.y[i], thp = P.in.thp[i], nx = P.in.nx[i], ny = P.in.ny[i];
        const float R_B = P.in.rb[i];
        Pcg rng{P.rng[pid], 1};
        // raw output j of this walker (one array per output: neighbouring lanes, neighbouring words)
        // all 33 loads are issued together, ahead of the branches of the arithmetic below
        // (issued one by one between those branches they cost a memory round trip each:
        // 74 % of the kernel's wave cycles were spent waiting)
        Vmm::f32x8 r0, r1, r2, r3;
        float raw32 = 0.0f;
        if (P.guiding) {
            const float *rp = P.net_out + i;
            const size_t ld = P.net_ld;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                r0[j] = rp[j * ld]; r1[j] = rp[(8 + j) * ld]; r2[j] = rp[(16 + j) * ld]; r3[j] = rp[(24 + j) * ld];
            }
            raw32 = rp[32 * ld];
        }
        const auto raw = [&](int j) { return j < 8 ? r0[j & 7] : j < 16 ? r1[j & 7] : j < 24 ? r2[j & 7] : j < 32 ? r3[j & 7] : raw32; };
        MixRegs<decltype(raw)> mix(raw);
        const SampleOut o = sample_step<TREE>(P, pid, on_n, x, y, thp, nx, ny, R_B, P.depth, P.guiding != 0, rng, mix, stk);
        P.rng[pid] = rng.state;
        guided_step = o.guided_step;
        if (o.dropped) {
            P.in.pid[i] = kDead;
        } else {
            hit_n = o.hit_n;
            P.in.pid[i] = pid | (hit_n ? kOnNeumann : 0u);
            P.in.x[i] = o.x; P.in.y[i] = o.y;
            P.in.thp[i] = o.thp;
            P.in.nx[i] = o.nx; P.in.ny[i] = o.ny;
            alive_after = true;
        }
    }
    wave_count(guided_step, &my_stats(P.stats)->guided);
    wave_count(live && P.guiding, &my_stats(P.stats)->net_points);
    wave_count(hit_n, &my_stats(P.stats)->nhits);
    if (P.last_depth) wave_count(alive_after, &my_stats(P.stats)->truncated);
}

// the per-pixel state of a solve before its first sample (prepareSolve, reference :112-128), for the launches that cannot set
// it on the fly (several samples per pixel: the items of a pixel arrive in any order)
__global__ __launch_bounds__(256) void guided_init_kernel(GParams P)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n_pixels) return;
    Pcg r0;
    pcg_seed_pixel(r0, p, P.st.width);
    P.rng[p] = r0.state;
    P.sol[3 * (size_t)p] = 0.0f; P.sol[3 * (size_t)p + 1] = 0.0f; P.sol[3 * (size_t)p + 2] = 0.0f;
    P.hint0[p] = 0;
}

// ---- a whole sample in one launch (half-precision network only) -----------------------------------
// The guiding network is small: its f16 weight fragments take 26 KB of LDS and its grid 123 KB of L2, and one
// evaluation is 52 matrix instructions per 16 walkers.  So a wave can evaluate the network for its OWN walkers,
// and nothing forces the walk through a global queue and a launch boundary at every depth.  This kernel is the
// uniform integrator's persistent round kernel (wost_hip.hip) with the guided step in its step phase:
//   lanes hold walkers; a lane is traversing the Dirichlet tree (TRAV), waits with a finished query (WAIT),
//   or wants the next pixel of the launch (REFILL).  Each trip of the loop runs the body more lanes are ready
//   for.  The step body = separate_finish, then -- wave-uniform, on the matrix cores -- the network for the
//   lanes that need it (compacted through LDS into 16-point units), then sample_step and the next query.
// Per pixel the arithmetic and the order of the random draws are those of separate_kernel / sample_kernel /
// tail_kernel, and the network arithmetic is that of net_forward_h_kernel (same device functions): the field,
// the training records and hence the trained weights are bit-identical to the one-launch-per-depth path.
// waves per CU sharing one copy of the weight fragments: 12 with the f16 fragments (26 KB), 10 with the fp32 ones (53 KB)
// (Round 5 tried two waves per SIMD here against the half-precision run-to-run difference: 18 of 20 full-size solve pairs identical at
// 512 threads, 17 of 20 at 768 -- not this kernel.  The difference was the training forward's first tile after a light kernel:
// wost_net.hip net_forward_h_kernel, EXPERIMENTS 20; with that fixed, 25 of 25 pairs are identical.)
#ifndef WOST_FUSED_THREADS_H
#define WOST_FUSED_THREADS_H 768
#endif
constexpr int fused_threads(bool half) { return half ? WOST_FUSED_THREADS_H : 640; }
// per wave, beside the lanes' columns: which lane the k-th point of the wave's network batch belongs to (64 bytes)
constexpr int fused_xch_words(bool) { return 16; }
// LDS fence + barrier between lanes of ONE wave that hand data to each other through LDS
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <bool EMISSIVE, bool TREE, bool SOURCE, bool HALF>
__global__ __launch_bounds__(fused_threads(HALF)) void guided_sample_kernel(GParams P, FusedNet F)
{
    constexpr int kFusedThreads = fused_threads(HALF), kXchWords = fused_xch_words(HALF);
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    __shared__ float s_scale[8];
    __shared__ uint32_t s_res[8], s_off[9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // traversal stacks wave by wave (entry i of a lane at [i * 64 + lane]: the column stride must be a power of two)
    const LdsColumn stk{lds_all + (size_t)wave * P.stack_words * 64 + lane, 64u};
    uint32_t *wbase = lds_all + (size_t)P.stack_words * kFusedThreads;
    uint2 *wf = reinterpret_cast<uint2 *>(wbase);               // HALF
    float *wf32 = reinterpret_cast<float *>(wbase);             // fp32
    const uint32_t w_words = HALF ? 2u * F.n_frag : F.n_mlp;
    uint8_t *xch_own = reinterpret_cast<uint8_t *>(wbase + w_words + (size_t)wave * kXchWords);
    if (HALF) {
        for (uint32_t e = threadIdx.x; e < F.n_frag; e += kFusedThreads) wf[e] = F.image[e];
    } else {
        for (uint32_t e = threadIdx.x; e < F.n_mlp; e += kFusedThreads) wf32[e] = F.frag32[e];
    }
    if (threadIdx.x < 9) {
        s_off[threadIdx.x] = F.off[threadIdx.x];
        if (threadIdx.x < 8) {
            s_scale[threadIdx.x] = F.scale[threadIdx.x];
            s_res[threadIdx.x] = F.res[threadIdx.x];
        }
    }
    __syncthreads();
    const uint2 *grid = HALF ? F.image + F.n_frag : nullptr;
    const int li = lane & 15, lg = lane >> 4;

    enum { MODE_TRAV = 1, MODE_WAIT = 3, MODE_DONE = 4, MODE_REFILL = 5, MODE_HUGE = 7 };
    int mode = MODE_REFILL;
    uint32_t pid = 0;
    bool on_n = false;
    float x = 0, y = 0, thp = 0, nx = 0, ny = 0;
    int32_t hint = 0;
    int depth = 0;
    Pcg rng{0, 1};
    Trav T = trav_begin(Closest{WOST_INF, -1});
    uint32_t sidx = 0;                     // which sample of its pixel in this launch the lane is walking
    uint32_t rofs = 0;                     // record column offset of that sample (training launches of several samples: one set each)
    uint32_t pool_next = 0, pool_end = 0, last_base = 0;
    const uint32_t tail_margin = (uint32_t)P.tail_margin_pct * (gridDim.x * blockDim.x / 100u);
    uint32_t c_steps = 0, c_started = 0, c_abs = 0, c_trunc = 0, c_hits = 0, c_guided = 0, c_net = 0;
    const bool has_d = P.dm.n_segs > 0;
    const uint32_t n_slots = (uint32_t)P.n_pixels;
    // A launch of several samples per pixel hands out (pixel, one more sample) items: item i belongs to pixel slot i % n_slots.
    // The samples of a pixel are sequential (one random stream, reference integrator.cu:71-77) but need not stay in one lane:
    // the pixel's state word counts the items that ARRIVED and the samples that are COMPLETE.  A lane that arrives at an idle
    // pixel (arrived == complete) walks its next sample; one that arrives while another lane is walking the pixel leaves its
    // item there as a credit and takes the next item; a lane that completes a sample and finds credits walks the next sample
    // itself.  Every transition is ONE atomic on that word, so no credit is lost, a pixel gets exactly n_samples samples, in
    // order, and no lane waits -- where a lane that kept its pixel for all the samples held the last pixels taken for n_samples
    // walks while the chip ran empty.  Which lane walks a sample has no influence on any result.
    const bool handed = P.n_samples > 1;
    const uint32_t n_items = n_slots * (uint32_t)P.n_samples;
    const bool tiled = ((P.st.width | P.st.height) & 7) == 0;
    bool dbg_out = false;
    if (P.dbg && lane == 0) atomicMin(P.dbg + 0, wall_clock64());

    // start sample `sidx` of pixel `pid` in this lane (begin_sample_kernel for one pixel)
    auto begin_walk = [&]() {
        const int px = (int)pid % P.st.width, py = (int)pid / P.st.width;
        eval_point(P.probe, px, py, P.st.width, P.st.height, x, y);
        on_n = false; thp = 1.0f; nx = 0.0f; ny = 0.0f; depth = 0;
        hint = ld_px(P.hint0 + pid);
        rng.state = ld_px(P.rng + pid);
        rofs = P.training ? sidx * n_slots : 0u;
        ++c_started;
        if (!has_d) {
            T.best = Closest{WOST_INF, -1};
            mode = MODE_WAIT;
        } else if (P.d0_valid || sidx > 0) {
            // the evaluation point of a pixel is the same for every sample: its query is cached
            T.best = Closest{ld_px(P.d0_d2 + pid), hint};
            mode = MODE_WAIT;
        } else {
            T = trav_begin(slot_candidate(P.dm, hint, x, y));
            mode = MODE_TRAV;
        }
    };

    for (;;) {
        // ---- lanes without a walker take the next item of the launch (begin_sample_kernel) ----
        const unsigned long long need = __ballot(mode == MODE_REFILL);
        if (need) {
            const uint32_t needed = (uint32_t)__popcll(need), avail = pool_end - pool_next;
            // Items are reserved 64 at a time (one atomic on the shared cursor per 64 items: same-address atomics
            // serialise in L2).  A reservation is private to its wave, so near the end of the launch the items a wave
            // holds back would start only when ITS lanes come free while other waves idle: within `tail_margin` items
            // of the end a wave reserves in small chunks.
            uint32_t fresh_base = 0, chunk = 64u;
            if (needed > avail) {
                if (P.tail_chunk > 0 && last_base + tail_margin >= n_items) chunk = (needed - avail + (uint32_t)P.tail_chunk - 1u) / (uint32_t)P.tail_chunk * (uint32_t)P.tail_chunk;
                if (lane == 0) fresh_base = atomicAdd(P.cursor, chunk);
                fresh_base = __shfl(fresh_base, 0);
                last_base = fresh_base;
            }
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            const uint32_t s2 = rank < avail ? pool_next + rank : fresh_base + (rank - avail);
            if (needed > avail) {
                pool_next = fresh_base + (needed - avail);
                pool_end = fresh_base + chunk;
            } else {
                pool_next += needed;
            }
            if (mode == MODE_REFILL) {
                if (s2 >= n_items) {
                    mode = MODE_DONE;
                    if (P.dbg && !dbg_out) {
                        dbg_out = true;
                        const unsigned long long t = wall_clock64();
                        atomicMin(P.dbg + 1, t);
                        atomicMax(P.dbg + 2, t);
                    }
                } else {
                    int p = (int)(handed ? s2 % n_slots : s2);
                    if (P.order) {
                        p = (int)P.order[p];
                    } else if (tiled) {
                        const int tiles_x = P.st.width >> 3, tile = p >> 6, in_tile = p & 63;
                        p = ((tile / tiles_x) * 8 + (in_tile >> 3)) * P.st.width + (tile % tiles_x) * 8 + (in_tile & 7);
                    }
                    if (!handed) {
                        // (a launch of several samples per pixel has these set before it starts: its items of a pixel may arrive in any order)
                        if (P.first_sample) {
                            Pcg r0;
                            pcg_seed_pixel(r0, p, P.st.width);
                            st_px(P.rng + p, r0.state);
                            st_px(P.sol + 3 * (size_t)p, 0.0f); st_px(P.sol + 3 * (size_t)p + 1, 0.0f); st_px(P.sol + 3 * (size_t)p + 2, 0.0f);
                            st_px(P.hint0 + p, (int32_t)0);
                        }
                        P.cur_depth[p] = 0;
                    }
                    const int px = p % P.st.width, py = p / P.st.width;
                    const int tile = (py >> 3) * ((P.st.width + 7) >> 3) + (px >> 3);
                    const bool active = (tile % P.shard_count) == P.shard_index && (P.mask == nullptr || P.mask[p] != 0);
                    if (active) {
                        bool start = true;
                        sidx = 0;
                        if (handed) {
                            const uint32_t old = __hip_atomic_fetch_add(P.pstate + p, 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            start = (old >> 16) == (old & 0xffffu);     // nobody is walking this pixel: its next sample is this lane's
                            sidx = old & 0xffffu;
                        }
                        if (start) {
                            pid = (uint32_t)p;
                            begin_walk();
                        }
                    }
                    // inactive pixel (mask, other shard) or a credit left with the pixel's walker: the lane asks again on the next trip
                }
            }
        }
        const int n_trav = __popcll(__ballot(mode == MODE_TRAV));
        const int n_wait = __popcll(__ballot(mode == MODE_WAIT));
        if (n_trav + n_wait == 0) {
            if (__ballot(mode == MODE_REFILL)) continue;
            break;
        }
        const bool step_phase = n_wait * P.wait_weight >= n_trav * 8;
        if (step_phase) {
            // ---- step phase -------------------------------------------------------------------
            const bool act = mode == MODE_WAIT;
            int status = SEP_DROPPED;
            float R_B = 0.0f;
            if (act) {
                ++c_steps;
                if (has_d && depth == 0 && !P.d0_valid && sidx == 0) st_px(P.d0_d2 + pid, T.best.d2);   // the pixel's first walk of the launch
                status = separate_finish<EMISSIVE, TREE, SOURCE>(P, pid, on_n, x, y, thp, nx, ny, depth, T.best, hint, R_B, rng, stk, rofs);
                if (status == SEP_ABSORBED) ++c_abs;
            }
            const bool keep = act && status == SEP_KEEP;
            const bool guiding = keep && depth < P.max_guided_depth;
            // ---- the network for the lanes that need it: all 64 lanes take part (matrix instructions) ----
            // Everything the network and the mixture exchange between lanes goes through the LDS COLUMNS of the walkers' lanes: a lane
            // in the step phase has finished its query, so its traversal stack is empty and its column (at least kVmmColWords deep)
            // is free.  The walker's lane leaves its input there; the sixteen-point unit writes the point's mixture back -- prepared,
            // four lanes per point: lane (i, g) owns the lobes g and 4 + g (vmm_lobe: two exponentials, the normalised mean, log I0)
            // and their weights -- and sample_step reads one lobe at a time (MixCol).  Forty registers of mixture per lane, and the
            // eight lobes computed by one lane in sixteen of a wave's 64, were what this kernel spilled and waited for.
            const unsigned long long bal = __ballot(guiding);
            if (bal) {
                const int n_need = __popcll(bal);
                const int rank = __popcll(bal & ((1ull << lane) - 1ull));
                float *col0 = reinterpret_cast<float *>(lds_all + (size_t)wave * P.stack_words * 64);     // entry e of lane L at col0[e * 64 + L]
                if (guiding) {
                    float ix, iy;
                    normalize_coord(P.box, x, y, ix, iy);
                    col0[lane] = ix;
                    col0[64 + lane] = iy;
                    xch_own[rank] = (uint8_t)lane;
                    ++c_net;
                }
                wave_lds_sync();
                for (int u = 0; 16 * u < n_need; ++u) {
                    const int q = 16 * u + li;
                    const bool qv = q < n_need;
                    float *ocol = col0 + (qv ? (int)xch_own[q] : lane);      // the column of the point's walker (never written when !qv)
                    const float qx = qv ? ocol[0] : 0.5f, qy = qv ? ocol[64] : 0.5f;
                    VmmLobe lobe[2];
                    float logit = 0.0f;
                    if (HALF) {
                        h4_t enc[2], out[3];
                        const h4_t hzero = h4_t{(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
                        enc[0] = hzero; enc[1] = hzero;
                        if (qv) {       // (a unit's unused points cost no gathers and no arithmetic: only the matrix instructions need every lane)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int lv = lg + 4 * h;
                                enc[h] = half_encode_level(grid, s_scale[lv], s_res[lv], s_off[lv], s_off[lv + 1] - s_off[lv], qx, qy);
                            }
                        }
                        half_mlp_unit(wf, F.w_off4, lane, enc, out);
                        // lane (i, g) receives the outputs 16 rt + 4 g + c of point i: the lobes g (rt = 0) and 4 + g (rt = 1), whole
                        lobe[0] = lobe[1] = VmmLobe{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                        if (qv) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) lobe[h] = vmm_lobe((float)out[h][0], (float)out[h][1], (float)out[h][2], (float)out[h][3]);
                        }
                        logit = (float)out[2][0];      // (output 32 in the lanes g = 0)
                    } else {
                        // the encoding goes through LDS once, as in net_forward_mfma_kernel: the matrix instruction wants feature
                        // 4 s + g of point i in lane (i, g), the lane has computed the features 4 lv .. 4 lv + 3 of its two levels
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int lv = lg + 4 * h;
                            const float4 f = f32_encode_level(F.grid32, s_scale[lv], s_res[lv], s_off[lv], s_off[lv + 1] - s_off[lv], qx, qy);
                            if (qv) {
                                ocol[(4 * lv + 0) * 64] = f.x; ocol[(4 * lv + 1) * 64] = f.y; ocol[(4 * lv + 2) * 64] = f.z; ocol[(4 * lv + 3) * 64] = f.w;
                            }
                        }
                        wave_lds_sync();
                        float b0[8], out[12];
#pragma unroll
                        for (int s_ = 0; s_ < 8; ++s_) b0[s_] = qv ? ocol[(4 * s_ + lg) * 64] : 0.0f;
                        wave_lds_sync();
                        f32_mlp_unit(wf32, F.w_off, lane, b0, out);
                        // out[4 rt + c] = output 16 rt + 4 c + g: component g of the lobes 4 rt + c -- turned lobe-major through the column
                        if (qv) {
#pragma unroll
                            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                                for (int c = 0; c < 4; ++c) ocol[(4 * (4 * rt + c) + lg) * 64] = out[4 * rt + c];
                            if (lg == 0) ocol[32 * 64] = out[8];
                        }
                        wave_lds_sync();
                        float r[2][4];
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int c = 0; c < 4; ++c) r[h][c] = ocol[(4 * (lg + 4 * h) + c) * 64];
                        logit = ocol[32 * 64];
                        wave_lds_sync();
#pragma unroll
                        for (int h = 0; h < 2; ++h) lobe[h] = vmm_lobe(r[h][0], r[h][1], r[h][2], r[h][3]);
                    }
                    if (qv) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int k = lg + 4 * h;
                            ocol[k * 64] = lobe[h].lambda; ocol[(8 + k) * 64] = lobe[h].kappa; ocol[(16 + k) * 64] = lobe[h].lb;
                            ocol[(24 + k) * 64] = lobe[h].mux; ocol[(32 + k) * 64] = lobe[h].muy;
                        }
                        if (lg == 0) ocol[40 * 64] = logit;
                    }
                    wave_lds_sync();
                    // the weights lambda / sum: the sum over the eight lobes in their order (Vmm::finish), by each of the point's four lanes
                    float total = 0.0f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) total += ocol[k * 64];
                    const float w0 = lobe[0].lambda / total, w1 = lobe[1].lambda / total;
                    wave_lds_sync();
                    if (qv) {
                        ocol[lg * 64] = w0;
                        ocol[(4 + lg) * 64] = w1;
                    }
                    wave_lds_sync();
                }
            }
            if (act) {
                bool ended = !keep;
                if (keep) {
                    MixCol mix(stk);
                    const SampleOut o = sample_step<TREE>(P, pid, on_n, x, y, thp, nx, ny, R_B, depth, guiding, rng, mix, stk, rofs);
                    if (o.guided_step) ++c_guided;
                    if (o.dropped) {
                        ended = true;
                    } else {
                        if (o.hit_n) ++c_hits;
                        x = o.x; y = o.y; thp = o.thp; nx = o.nx; ny = o.ny; on_n = o.hit_n;
                        ++depth;
                        if (depth >= P.st.max_depth) {
                            ++c_trunc;
                            ended = true;
                        }
                    }
                }
                if (ended) {
                    st_px(P.rng + pid, rng.state);
                    mode = MODE_REFILL;
                    if (handed) {
                        // the sample is complete: every store of the pixel's state has left this CU before the state word says so
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        const uint32_t old = __hip_atomic_fetch_add(P.pstate + pid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((old >> 16) > (old & 0xffffu) + 1u) {
                            // items that arrived meanwhile left their credit: the pixel's next sample starts right away, here
                            // (guiding phase, or a training launch of several samples: the pixel's random stream simply continues)
                            sidx = (old & 0xffffu) + 1u;
                            begin_walk();
                        }
                    }
                } else if (has_d) {
                    T = trav_begin(slot_candidate(P.dm, hint, x, y));
                    // a walker that strayed so far that the whole mesh ties within rounding: answered by the wave below
                    mode = T.best.d2 > P.dm.huge2 ? MODE_HUGE : MODE_TRAV;
                } else {
                    T.best = Closest{WOST_INF, -1};
                    mode = MODE_WAIT;
                }
            }
            {
                unsigned long long hb = __ballot(mode == MODE_HUGE);
                while (hb) {
                    const int src = __builtin_ctzll(hb);
                    const Closest r = closest_point_wave(P.dm, __shfl(x, src), __shfl(y, src));
                    if (lane == src) {
                        T.best = r;
                        mode = MODE_WAIT;
                    }
                    hb &= hb - 1;
                }
            }
        } else {
            // ---- traversal phase: every traversing lane visits nodes ----
            for (int b = 0; b < P.trav_burst; ++b) {
                if (mode == MODE_TRAV) {
                    if (!trav_visit<true>(P.dm, x, y, T, stk)) mode = MODE_WAIT;     // the form that is exact at any distance from the mesh (wost_device.h)
                }
            }
        }
    }
    if (P.dbg && lane == 0) atomicMax(P.dbg + 3, wall_clock64());
    const uint32_t s_steps = wave_sum(c_steps), s_st = wave_sum(c_started), s_abs = wave_sum(c_abs), s_tr = wave_sum(c_trunc),
                   s_hit = wave_sum(c_hits), s_g = wave_sum(c_guided), s_net = wave_sum(c_net);
    if (lane == 0) {
        GStatsDev *st = my_stats(P.stats);
        if (s_steps) atomicAdd(&st->steps, (unsigned long long)s_steps);
        if (s_st) atomicAdd(&st->started, (unsigned long long)s_st);
        if (s_abs) atomicAdd(&st->absorbed, (unsigned long long)s_abs);
        if (s_tr) atomicAdd(&st->truncated, (unsigned long long)s_tr);
        if (s_hit) atomicAdd(&st->nhits, (unsigned long long)s_hit);
        if (s_g) atomicAdd(&st->guided, (unsigned long long)s_g);
        if (s_net) atomicAdd(&st->net_points, (unsigned long long)s_net);
    }
}

// ---- training set: generate_training_data (reference train.h:423-471), ordered ----------------
struct TrainSet {
    float *xy, *dir, *sol, *li, *pdf, *nrm;
    uint8_t *onn;
};

struct TParams {
    GAabb box;
    const uint32_t *cur_depth;
    const float *rec;          // column 0 of the record set to gather
    size_t rec_ld;
    int32_t n_pixels;
    uint32_t train_offset, train_stride;
    int32_t n_train_pixels;
    uint32_t *block_sums;     // [n_blocks + 1]
    TrainSet ts;
};

__device__ __forceinline__ bool record_valid(const TParams &T, int slot, uint32_t pid, float out[kRecFields])
{
    for (int f = 0; f < kRecFields; ++f) out[f] = T.rec[((size_t)slot * kRecFields + f) * T.rec_ld + pid];
    if (!aabb_contains(T.box, out[3], out[4])) return false;
    // |solution / thp| per channel, 0 where the throughput vanished
    for (int c = 0; c < 3; ++c) {
        float v = 0.0f;
        if (fabsf(out[8]) > 1e-5f) v = out[c] / out[8];
        out[c] = fabsf(v);
    }
    float ix, iy;
    normalize_coord(T.box, out[3], out[4], ix, iy);
    const bool bad = isnan(ix) || isnan(iy) || isnan(out[5]) || isnan(out[6]) || isnan(out[7]) || out[7] == 0 ||
                     isnan(out[0]) || isnan(out[1]) || isnan(out[2]);
    out[3] = ix;
    out[4] = iy;
    return !bad;
}

// pass 1: samples per training pixel -> per-block sums; pass 3: scatter at the scanned offsets
template <bool SCATTER>
__global__ __launch_bounds__(256) void train_set_kernel(TParams T)
{
    __shared__ uint32_t sh[256];
    const int t = blockIdx.x * 256 + threadIdx.x;
    uint32_t cnt = 0;
    uint32_t pid = 0, depth = 0;
    if (t < T.n_train_pixels) {
        pid = T.train_offset + (uint32_t)t * T.train_stride;
        depth = min(T.cur_depth[pid], (uint32_t)kMaxTrainDepth);
    }
    float r[kMaxTrainDepth][kRecFields];
    bool ok[kMaxTrainDepth];
#pragma unroll
    for (int k = 0; k < kMaxTrainDepth; ++k) {
        ok[k] = (uint32_t)k < depth && record_valid(T, k, pid, r[k]);
        cnt += ok[k] ? 1u : 0u;
    }
    // block-level exclusive scan of cnt
    sh[threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0u;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    if (!SCATTER) {
        if (threadIdx.x == 255) T.block_sums[blockIdx.x] = sh[255];
        return;
    }
    uint32_t o = T.block_sums[blockIdx.x] + sh[threadIdx.x] - cnt;
#pragma unroll
    for (int k = 0; k < kMaxTrainDepth; ++k) {
        if (!ok[k]) continue;
        const float *q = r[k];
        T.ts.xy[2 * (size_t)o] = q[3]; T.ts.xy[2 * (size_t)o + 1] = q[4];
        T.ts.dir[2 * (size_t)o] = q[5]; T.ts.dir[2 * (size_t)o + 1] = q[6];
        T.ts.sol[3 * (size_t)o] = q[0]; T.ts.sol[3 * (size_t)o + 1] = q[1]; T.ts.sol[3 * (size_t)o + 2] = q[2];
        T.ts.li[o] = (q[0] + q[1] + q[2]) / 3.0f;      // Color::mean() (train.h:519)
        T.ts.pdf[o] = q[7];
        T.ts.nrm[2 * (size_t)o] = q[9]; T.ts.nrm[2 * (size_t)o + 1] = q[10];
        T.ts.onn[o] = q[11] != 0.0f ? 1 : 0;
        ++o;
    }
}

// pass 2: exclusive scan of the block sums by one block; block_sums[n_blocks] = total
__global__ __launch_bounds__(256) void train_scan_kernel(uint32_t *block_sums, int n_blocks)
{
    __shared__ uint32_t sh[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 256) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sums[i] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            uint32_t a = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0u;
            __syncthreads();
            sh[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < n_blocks) block_sums[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[n_blocks] = carry;
}

__global__ void resolve_kernel(const float *sol, int n, float spp, float *field)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * n) field[i] = sol[i] / spp;
}

}  // namespace wost

using namespace wost;

struct wost_guided {
    int device = 0;
    wost_guided_settings gs{};
    wost_handle scene = nullptr;       // owns the uploaded meshes
    wost_net_handle net = nullptr;
    SceneView view{};
    size_t n_pixels = 0;
    std::vector<void *> allocs;
    GQueue q[2]{};
    uint32_t *counts = nullptr;        // [2]
    uint32_t *host_counts = nullptr;   // pinned [32]: [1 + j] = size of the training set of record set j
    uint32_t *depth_counts = nullptr;  // pinned [max_depth]: queue size after every depth, copied back without waiting
    std::vector<hipEvent_t> depth_events;
    uint64_t *rng = nullptr;
    float *sol = nullptr, *field = nullptr, *rec = nullptr, *net_in = nullptr, *net_out = nullptr;
    uint32_t *cur_depth = nullptr;
    int32_t *hint0 = nullptr;
    float *d0_d2 = nullptr;            // fused sample kernel: cached query of every evaluation point
    uint32_t *cursor = nullptr;        // fused sample kernel: next pixel slot of the launch
    uint32_t *pstate = nullptr;        // fused sample kernel, several samples per launch: arrived / complete counts per pixel
    unsigned long long *dbg = nullptr; // fused sample kernel: WOST_GUIDED_DEBUG timeline
    GStatsDev *stats = nullptr;
    uint32_t *block_sums = nullptr;
    int n_train_blocks = 0, n_train_pixels = 0;
    uint64_t host_rng_state = 0, host_rng_inc = 3;   // mGuiding.sampler of the reference (trainPixelOffset draws)
    TrainSet ts{};
    uint32_t last_train_n = 0;
    uint32_t last_train_offset = 0;    // trainPixelOffset of the most recent solve
    GAabb box{};
    wost_sync_fn sync = nullptr;       // shared-network mode: collective hooks of the caller
    void *sync_user = nullptr;
    wost_frame_fn frame_fn = nullptr;  // intermediate frames (saveSppMetrics / saveTimeMetrics)
    void *frame_user = nullptr;
    int32_t frame_spp_every = 0, frame_spp_until = 0, frame_time_every = 0;
    EventRing net_events;              // timing of the network-evaluating launches
    // opt-in pipelined training order (wost_guided_set_option "pipeline"): the training pass of sample k runs on its own
    // stream while sample k + 1 walks with a frozen copy of the weights of pass k - 1
    int pipeline = 0;
    // opt-in "train_group" S: a training launch walks S samples of every pixel back to back (one record set per sample) and the S
    // training passes follow the launch: the drain of the longest walks is paid once per S samples; S = 1 is the reference's order
    int train_group = 1;
    // the pixels of a fused launch in the order of wost_order.h -- longest expected walk first, by the cached distance of the
    // evaluation points -- from the second launch of a solve on (the first fills the cache): the drain of a launch is its longest
    // walks, and they start first (the uniform one-launch path: 2.64 -> 2.48 ms at 1 spp)
    int walk_order = 1;
    WalkOrder order;
    int rec_sets = 1;                     // record sets rec / cur_depth are allocated for
    std::vector<TrainSet> ts_more;        // training-set arrays of the sets 1 .. (pipelined groups train while the next group walks)
    hipStream_t train_stream = nullptr;
    void *snap[2] = {nullptr, nullptr};   // frozen inference images, used alternately
    hipEvent_t ev_ts = nullptr;           // the training set of a sample is complete (walk stream)
    hipEvent_t ev_train[2] = {nullptr, nullptr};   // training pass k is complete and its weights are in snap[k % 2] (training stream)
    EventRing train_events;               // device time of the training passes in that mode
};

static uint32_t host_pcg_next(wost_guided *g)
{
    const uint64_t old = g->host_rng_state;
    g->host_rng_state = old * WOST_PCG32_MULT + g->host_rng_inc;
    const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

#define G_TRY(expr)                                                                                      \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return set_error(WOST_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
static hipError_t galloc(wost_guided *g, T **p, size_t count)
{
    void *v = nullptr;
    hipError_t e = hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) {
        g->allocs.push_back(v);
        *p = reinterpret_cast<T *>(v);
    }
    return e;
}

// release one buffer of galloc before the handle goes (a buffer that is replaced by a larger one)
static void gfree_one(wost_guided *g, void *p)
{
    if (!p) return;
    for (size_t i = 0; i < g->allocs.size(); ++i)
        if (g->allocs[i] == p) {
            g->allocs.erase(g->allocs.begin() + (long)i);
            (void)hipFree(p);
            return;
        }
}

static void guided_free(wost_guided *g)
{
    if (!g) return;
    (void)hipSetDevice(g->device);
    for (void *p : g->allocs) (void)hipFree(p);
    if (g->host_counts) (void)hipHostFree(g->host_counts);
    if (g->depth_counts) (void)hipHostFree(g->depth_counts);
    for (hipEvent_t e : g->depth_events) (void)hipEventDestroy(e);
    g->net_events.destroy();
    g->train_events.destroy();
    if (g->ev_ts) (void)hipEventDestroy(g->ev_ts);
    for (hipEvent_t e : g->ev_train)
        if (e) (void)hipEventDestroy(e);
    if (g->train_stream) (void)hipStreamDestroy(g->train_stream);
    order_free(g->order);
    if (g->net) wost_net_destroy(g->net);
    if (g->scene) wost_destroy(g->scene);
    delete g;
}

extern "C" {

int wost_guided_create(const wost_scene_desc *scene, const wost_guided_settings *s, const wost_net_config *net,
                       uint64_t net_seed, int device, wost_guided_handle *out)
{
    if (!scene || !s || !net || !out) return set_error(WOST_ERR_INVALID, "null argument");
    *out = nullptr;
    if (s->width <= 0 || s->height <= 0 || s->spp < 0 || s->max_depth <= 0 || s->train_spp_count < 0 ||
        s->max_guided_depth_training < 0 || s->max_guided_depth_guiding < 0 || s->batch_size < 128 ||
        s->min_batch_size < 1 || s->batches_per_spp < 0 || s->train_pixel_stride < 1 || s->train_pixel_offset < -1 ||
        s->train_pixel_offset >= s->train_pixel_stride || !(s->loss_scale > 0.0f) ||
        !(s->aabb_min[0] < s->aabb_max[0]) || !(s->aabb_min[1] < s->aabb_max[1]))
        return set_error(WOST_ERR_INVALID, "bad guided settings");
    if (s->max_train_depth < 0 || s->max_train_depth > kMaxTrainDepth)
        return set_error(WOST_ERR_UNSUPPORTED, "max_train_depth must be in 0..4 (record slots per pixel)");
    if (net->n_output != 33) return set_error(WOST_ERR_UNSUPPORTED, "the 2-D mixture needs 33 network outputs");
    if ((int64_t)s->width * s->height >= (1ll << 31)) return set_error(WOST_ERR_UNSUPPORTED, "frame too large");
    wost_guided *g = new (std::nothrow) wost_guided();
    if (!g) return set_error(WOST_ERR_NOMEM, "out of host memory");
    g->device = device;
    g->gs = *s;
    auto bail = [&](int code) {
        guided_free(g);
        return code;
    };
    // the scene upload and LBVH build are the uniform integrator's (spp of the base settings is unused here)
    wost_settings base{s->width, s->height, 1, s->max_depth, s->eps_shell};
    int rc = wost_create(scene, &base, device, &g->scene);
    if (rc != WOST_OK) return bail(rc);
    rc = wost_net_create(device, net, net_seed, &g->net);
    if (rc != WOST_OK) return bail(rc);
    g->view = scene_view(g->scene);
    g->n_pixels = (size_t)s->width * s->height;
    const size_t N = g->n_pixels;
    {
        // normalizeSpatialCoord's inflated box, evaluated once in fp32 exactly like train.h:149-155
        GAabb &b = g->box;
        b.minx = s->aabb_min[0]; b.miny = s->aabb_min[1]; b.maxx = s->aabb_max[0]; b.maxy = s->aabb_max[1];
        const float ex = b.maxx - b.minx, ey = b.maxy - b.miny;
        const float infl = std::sqrt(ex * ex + ey * ey) * 0.005f;
        const float lox = b.minx - infl, loy = b.miny - infl, hix = b.maxx + infl, hiy = b.maxy + infl;
        b.cx = (lox + hix) / 2.0f; b.cy = (loy + hiy) / 2.0f;
        b.ex = hix - lox; b.ey = hiy - loy;
    }
    hipError_t e = hipSuccess;
#define GA(ptr, count) if (e == hipSuccess) e = galloc(g, &(ptr), (count))
    for (int k = 0; k < 2; ++k) {
        GA(g->q[k].pid, N); GA(g->q[k].x, N); GA(g->q[k].y, N); GA(g->q[k].thp, N); GA(g->q[k].nx, N); GA(g->q[k].ny, N);
        GA(g->q[k].rb, N); GA(g->q[k].hint, N);
    }
    GA(g->counts, 2); GA(g->rng, N); GA(g->sol, 3 * N); GA(g->field, 3 * N);
    GA(g->rec, (size_t)kMaxTrainDepth * kRecFields * N);
    GA(g->net_in, 2 * N); GA(g->net_out, 33 * N); GA(g->cur_depth, N); GA(g->hint0, N); GA(g->stats, kStatCopies);
    GA(g->dbg, 4); GA(g->d0_d2, N); GA(g->cursor, 1); GA(g->pstate, N);
    // sized for offset 0 (the largest set); the offset of a solve may be drawn per solve (run_guided)
    g->n_train_pixels = (int)((N + (size_t)s->train_pixel_stride - 1) / (size_t)s->train_pixel_stride);
    g->n_train_blocks = (g->n_train_pixels + 255) / 256;
    // the integrator's host sampler: setSeed(ELAINA_DEFAULT_RNG_SEED = 42) with the default sequence 1
    // (reference integrator/guided/integrator.cu:1134, core/sampler.h:20-27, core/config.h:7)
    g->host_rng_inc = (1ull << 1u) | 1ull;
    g->host_rng_state = 0;
    (void)host_pcg_next(g);
    g->host_rng_state += 42ull;
    (void)host_pcg_next(g);
    GA(g->block_sums, (size_t)g->n_train_blocks + 1);
    const size_t M = (size_t)g->n_train_pixels * kMaxTrainDepth;
    GA(g->ts.xy, 2 * M); GA(g->ts.dir, 2 * M); GA(g->ts.sol, 3 * M); GA(g->ts.li, M); GA(g->ts.pdf, M); GA(g->ts.nrm, 2 * M);
    GA(g->ts.onn, M);
#undef GA
    if (e == hipSuccess) e = hipHostMalloc((void **)&g->host_counts, 32 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **)&g->depth_counts, (size_t)std::max(1, s->max_depth) * sizeof(uint32_t));
    for (int d = 0; e == hipSuccess && d < s->max_depth; ++d) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) g->depth_events.push_back(ev);
    }
    if (e == hipSuccess) e = g->net_events.init(1024);
    if (e != hipSuccess) {
        set_error(WOST_ERR_DEVICE, std::string("guided allocation: ") + hipGetErrorString(e));
        return bail(WOST_ERR_DEVICE);
    }
    *out = g;
    return WOST_OK;
}

int wost_guided_destroy(wost_guided_handle h)
{
    guided_free(h);
    return WOST_OK;
}

int wost_guided_network(wost_guided_handle h, wost_net_handle *net)
{
    if (!h || !net) return set_error(WOST_ERR_INVALID, "null argument");
    *net = h->net;
    return WOST_OK;
}

int wost_guided_set_frame_callback(wost_guided_handle h, wost_frame_fn fn, void *user, int32_t spp_every, int32_t spp_until,
                                   int32_t time_every)
{
    if (!h) return set_error(WOST_ERR_INVALID, "null argument");
    h->frame_fn = fn;
    h->frame_user = user;
    h->frame_spp_every = spp_every;
    h->frame_spp_until = spp_until;
    h->frame_time_every = time_every;
    return WOST_OK;
}

int wost_guided_set_sync(wost_guided_handle h, wost_sync_fn fn, void *user)
{
    if (!h) return set_error(WOST_ERR_INVALID, "null argument");
    h->sync = fn;
    h->sync_user = user;
    return WOST_OK;
}

int wost_guided_set_option(wost_guided_handle h, const char *key, double value)
{
    if (!h || !key) return set_error(WOST_ERR_INVALID, "null argument");
    const std::string k(key);
    if (k == "pipeline") {
        if (!(value == 0.0 || value == 1.0)) return set_error(WOST_ERR_INVALID, "pipeline is 0 or 1");
        h->pipeline = (int)value;
        return WOST_OK;
    }
    if (k == "walk_order") {
        if (!(value == 0.0 || value == 1.0)) return set_error(WOST_ERR_INVALID, "walk_order is 0 or 1");
        h->walk_order = (int)value;
        return WOST_OK;
    }
    if (k == "train_group") {
        if (!(value >= 1.0 && value <= 16.0 && value == (double)(int)value)) return set_error(WOST_ERR_INVALID, "train_group is an integer in 1..16");
        h->train_group = (int)value;
        return WOST_OK;
    }
    return set_error(WOST_ERR_INVALID, "unknown option '" + k + "'");
}

int wost_guided_scene(wost_guided_handle h, wost_handle *scene)
{
    if (!h || !scene) return set_error(WOST_ERR_INVALID, "null argument");
    *scene = h->scene;
    return WOST_OK;
}

int wost_guided_query_network(wost_guided_handle h, const float *pts, int32_t n, float *raw)
{
    if (!h || !pts || !raw || n < 0) return set_error(WOST_ERR_INVALID, "bad argument");
    std::vector<float> xy((size_t)n * 2);
    const GAabb &b = h->box;
    for (int i = 0; i < n; ++i) {
        xy[2 * (size_t)i] = 0.5f + (pts[2 * (size_t)i] - b.cx) / b.ex;
        xy[2 * (size_t)i + 1] = 0.5f + (pts[2 * (size_t)i + 1] - b.cy) / b.ey;
    }
    return wost_net_inference(h->net, xy.data(), n, raw, 1);
}

int wost_guided_train_set(wost_guided_handle h, int32_t capacity, int32_t *n, float *xy, float *dir, float *solution,
                          float *dir_pdf, float *normal, uint8_t *on_neumann)
{
    if (!h || !n || capacity < 0) return set_error(WOST_ERR_INVALID, "bad argument");
    G_TRY(hipSetDevice(h->device));
    *n = (int32_t)h->last_train_n;
    const size_t m = std::min<size_t>(h->last_train_n, (size_t)capacity);
    if (m == 0) return WOST_OK;
    if (xy) G_TRY(hipMemcpy(xy, h->ts.xy, m * 2 * sizeof(float), hipMemcpyDeviceToHost));
    if (dir) G_TRY(hipMemcpy(dir, h->ts.dir, m * 2 * sizeof(float), hipMemcpyDeviceToHost));
    if (solution) G_TRY(hipMemcpy(solution, h->ts.sol, m * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (dir_pdf) G_TRY(hipMemcpy(dir_pdf, h->ts.pdf, m * sizeof(float), hipMemcpyDeviceToHost));
    if (normal) G_TRY(hipMemcpy(normal, h->ts.nrm, m * 2 * sizeof(float), hipMemcpyDeviceToHost));
    if (on_neumann) G_TRY(hipMemcpy(on_neumann, h->ts.onn, m, hipMemcpyDeviceToHost));
    return WOST_OK;
}

}  // extern "C"

// the shared driver: field_host (n_pixels*3, may be null) and/or field_dev (device, n_pixels*3)
static int run_guided(wost_guided *g, int shard_index, int shard_count, float *field_host, float *field_dev,
                      wost_guided_stats *stats)
{
    const auto t_start = std::chrono::high_resolution_clock::now();
    G_TRY(hipSetDevice(g->device));
    const wost_guided_settings &s = g->gs;
    const SceneView &v = g->view;
    hipStream_t stream = v.stream;
    const int N = (int)g->n_pixels;
    const bool emissive = v.nm.n_segs > 0 && v.nm.emissive;
    const bool tree = v.nm.n_segs > WOST_FLAT_MAX;
    const int d_levels = v.dm.n_segs > 0 ? v.dm.levels : 1, n_levels = v.nm.n_segs > 0 ? v.nm.levels : 1;
    const int stack_words = 3 * std::max(d_levels, n_levels) + 1;
    const size_t lds = (size_t)stack_words * 256 * sizeof(uint32_t);
    uint32_t launches = 0;
    const uint64_t net_launches_before = net_launch_count(g->net);
    double train_ms = 0.0;
    uint64_t train_samples = 0;
    const int opt_before = net_optimizer_steps(g->net);

    G_TRY(hipMemsetAsync(g->stats, 0, kStatCopies * sizeof(GStatsDev), stream));
    if (g->sync) {
        // shared network: the summed gradients are divided by the number of ranks (include/wost.h)
        // (WOST_SYNC_RANKS_I64_HOST was added to the callback's ops in library version 0.2: a callback written against 0.1
        // answers WOST_SYNC_UNSUPPORTED and keeps the old behaviour -- the summed gradient is used undivided; any other
        // non-zero return is a failure and ends the solve)
        int64_t ranks = 1;
        const int src = g->sync(g->sync_user, WOST_SYNC_RANKS_I64_HOST, &ranks, 1);
        if (src == WOST_SYNC_UNSUPPORTED) {
            fprintf(stderr, "wost: the sync callback does not know WOST_SYNC_RANKS_I64_HOST; shared gradients stay undivided\n");
            ranks = 1;
        } else if (src != 0 || ranks < 1) {
            // a genuine failure (or a nonsensical answer) must not train on: the other ranks would step with another gradient
            return set_error(WOST_ERR_DEVICE, "sync callback failed (rank count)");
        }
        net_set_gradient_divisor(g->net, (float)ranks);
    } else {
        net_set_gradient_divisor(g->net, 1.0f);
    }
    GParams P{};
    P.dm = v.dm; P.nm = v.nm; P.st = v.st; P.probe = v.probe; P.src = v.src; P.box = g->box; P.mask = v.mask;
    P.rng = g->rng; P.sol = g->sol; P.cur_depth = g->cur_depth; P.rec = g->rec; P.hint0 = g->hint0;
    P.rec_ld = (size_t)N; P.net_in = g->net_in; P.net_out = g->net_out; P.net_ld = (size_t)N; P.stats = g->stats; P.n_pixels = N; P.stack_stride = 256;
    P.max_train_depth = s.max_train_depth;
    // prepareSolve (integrator.cu:126): trainPixelOffset = stride <= 1 ? 0 : sampler.get1D() * stride, one draw per
    // solve from the integrator's host sampler; a caller-fixed offset (>= 0) overrides the draw
    uint32_t train_offset = 0;
    if (s.train_pixel_stride > 1) {
        if (s.train_pixel_offset >= 0) train_offset = (uint32_t)s.train_pixel_offset;
        else {
            union { uint32_t u; float f; } x;
            x.u = (host_pcg_next(g) >> 9) | 0x3f800000u;
            train_offset = (uint32_t)((x.f - 1.0f) * (float)s.train_pixel_stride);
        }
    }
    g->last_train_offset = train_offset;
    const int n_train_pixels = (int)(((size_t)N - train_offset + (size_t)s.train_pixel_stride - 1) / (size_t)s.train_pixel_stride);
    const int n_train_blocks = (n_train_pixels + 255) / 256;
    P.train_offset = train_offset; P.train_stride = (uint32_t)s.train_pixel_stride;
    P.shard_index = shard_index; P.shard_count = shard_count;

    // A whole sample is one launch (guided_sample_kernel, the network evaluated inside the wave) whenever the network has
    // the reference's shape: with the half-precision image when "precision" is 16, with the fp32 fragments otherwise.
    // WOST_GUIDED_FUSED=0 keeps the one-launch-per-depth path for comparison (same field, same records).
    bool fused = false, fused_half = false;
    FusedNet F{};
    // the members of F that describe the network's shape; false when the fused kernel does not exist for it
    auto fused_shape = [](const NetLayout *L, FusedNet &Fn) {
        if (!(L && L->n_levels == 8 && L->n_features == 4 && L->enc == 32 && L->n_neurons == 64 && L->n_hidden == 3 && L->n_out_padded == 48 &&
              L->n_out == 33))
            return false;
        Fn.n_frag = L->n_mlp / 4;
        Fn.n_mlp = L->n_mlp;
        for (int l = 0; l < 4; ++l) { Fn.w_off[l] = L->w_off[l]; Fn.w_off4[l] = L->w_off[l] / 4; }
        for (int l = 0; l < 8; ++l) { Fn.scale[l] = L->scale[l]; Fn.res[l] = (uint32_t)L->res[l]; }
        for (int l = 0; l <= 8; ++l) Fn.off[l] = L->level_off[l];
        return true;
    };
    {
        const char *env = std::getenv("WOST_GUIDED_FUSED");
        HalfNetView hv{};
        F32NetView fv{};
        const NetLayout *L = nullptr;
        if (env && env[0] == '0') {
        } else if (net_half_view(g->net, &hv) == WOST_OK) {
            L = &hv.L;
            fused_half = true;
            F.image = hv.image;
        } else if (net_f32_view(g->net, &fv) == WOST_OK) {
            L = &fv.L;
            F.frag32 = fv.frag;
            F.grid32 = fv.grid;
        }
        fused = fused_shape(L, F);
    }
    P.d0_d2 = g->d0_d2; P.cursor = g->cursor; P.stack_words = std::max(stack_words, kVmmColWords);      // the fused kernel's columns also hold a walker's mixture
    P.wait_weight = 4; P.trav_burst = 10;     // measured on config 4 (a sweep over both constants)
    if (const char *w = std::getenv("WOST_GUIDED_WAIT_WEIGHT")) P.wait_weight = std::max(1, std::atoi(w));
    if (const char *w = std::getenv("WOST_GUIDED_TRAV_BURST")) P.trav_burst = std::max(1, std::atoi(w));
    P.tail_chunk = 4; P.tail_margin_pct = 300;      // a sweep over both, DESIGN.md 4.9: 3.52 -> 3.37 ms per trained sample of config 4
    if (const char *w = std::getenv("WOST_GUIDED_TAIL_CHUNK")) P.tail_chunk = std::max(0, std::atoi(w));
    if (const char *w = std::getenv("WOST_GUIDED_TAIL_MARGIN")) P.tail_margin_pct = std::max(0, std::atoi(w));
    const int n_fused_threads = fused_threads(fused_half);
    const size_t lds_fused = ((size_t)P.stack_words * n_fused_threads + (size_t)(n_fused_threads / 64) * fused_xch_words(fused_half) +
                              (fused_half ? (size_t)2 * F.n_frag : (size_t)F.n_mlp)) * sizeof(uint32_t);
    if (fused) {
        // the stack columns grow with the depth of the trees: when the fused kernel's LDS no longer fits a block (the fp32
        // fragments with a 10-level tree), the solve takes the one-launch-per-depth path instead of failing at the launch
        int max_lds = 0;
        if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, g->device) != hipSuccess || max_lds <= 0) max_lds = 64 * 1024;
        int dev_optin = 0;
        if (hipDeviceGetAttribute(&dev_optin, hipDeviceAttributeSharedMemPerBlockOptin, g->device) == hipSuccess && dev_optin > max_lds) max_lds = dev_optin;
        if (lds_fused > (size_t)max_lds) fused = false;
    }

    // one launch of the fused sample kernel on `st` with the network image `Fn` and the launch state in P
    const uint32_t *walk_order = nullptr;     // built once per solve, when the cache of the evaluation points' queries is full
    auto launch_walk = [&](const FusedNet &Fn, hipStream_t st) -> int {
        G_TRY(hipMemsetAsync(g->cursor, 0, sizeof(uint32_t), st));
        // (not for the launches of several samples per pixel, whose drain is one walk whatever the order, nor for one rank's shard:
        // seven of eight pixels of the order belong to other ranks and a lane asks eight times for one -- shard 0 of 8 of config 5
        // 0.219 -> 0.246 s; the whole frame, one sample per launch: config 4 1.420 -> 1.400 s in half precision, 2.206 -> 2.174 in fp32)
        const bool ordered = g->walk_order && P.d0_valid && v.dm.n_segs > 0 && P.n_samples == 1 && P.shard_count == 1;
        if (ordered && !walk_order) {
            if (g->order.cap < (size_t)N) G_TRY((hipError_t)order_alloc(g->order, (size_t)N));
            G_TRY((hipError_t)order_by_distance(g->order, g->d0_d2, (uint32_t)N, st, &walk_order));
            launches += 2;
        }
        P.order = ordered ? walk_order : nullptr;
        P.pstate = g->pstate;
        if (P.n_samples > 1) {
            // several samples of every pixel in one launch: the pixel state words start at zero, and what a one-sample launch
            // sets on the fly is set before the launch
            G_TRY(hipMemsetAsync(g->pstate, 0, (size_t)N * sizeof(uint32_t), st));
            if (P.first_sample) {
                hipLaunchKernelGGL(guided_init_kernel, dim3((N + 255) / 256), dim3(256), 0, st, P);
                ++launches;
            }
            if (P.training) G_TRY(hipMemsetAsync(g->cur_depth, 0, (size_t)N * (size_t)P.n_samples * sizeof(uint32_t), st));
            launches += 2;
        }
        const unsigned gridf = (unsigned)std::min<size_t>(256, ((size_t)N + n_fused_threads - 1) / n_fused_threads);
#define LAUNCH_FUSED(E, T)                                                                                                      \
    do {                                                                                                                        \
        auto kfn = v.src.rgb ? (fused_half ? guided_sample_kernel<E, T, true, true> : guided_sample_kernel<E, T, true, false>)    \
                             : (fused_half ? guided_sample_kernel<E, T, false, true> : guided_sample_kernel<E, T, false, false>); \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fused); \
        hipLaunchKernelGGL(kfn, dim3(gridf), dim3(n_fused_threads), lds_fused, st, P, Fn);                                       \
    } while (0)
        if (emissive) { if (tree) LAUNCH_FUSED(true, true); else LAUNCH_FUSED(true, false); }
        else          { if (tree) LAUNCH_FUSED(false, true); else LAUNCH_FUSED(false, false); }
#undef LAUNCH_FUSED
        ++launches;
        G_TRY(hipGetLastError());
        return WOST_OK;
    };
    // the training set of record set j of the launch that has just walked, in (pixel, record) order, into `ts`; its size
    // arrives in host_counts[1 + j]
    auto enqueue_train_set = [&](hipStream_t st, int j, const TrainSet &ts) -> int {
        TParams T{};
        T.box = g->box; T.cur_depth = g->cur_depth + (size_t)j * (size_t)N; T.rec = g->rec + (size_t)j * (size_t)N; T.rec_ld = P.rec_ld; T.n_pixels = N;
        T.train_offset = train_offset; T.train_stride = (uint32_t)s.train_pixel_stride;
        T.n_train_pixels = n_train_pixels; T.block_sums = g->block_sums; T.ts = ts;
        hipLaunchKernelGGL((train_set_kernel<false>), dim3(n_train_blocks), dim3(256), 0, st, T);
        hipLaunchKernelGGL(train_scan_kernel, dim3(1), dim3(256), 0, st, g->block_sums, n_train_blocks);
        hipLaunchKernelGGL((train_set_kernel<true>), dim3(n_train_blocks), dim3(256), 0, st, T);
        launches += 3;
        G_TRY(hipGetLastError());
        G_TRY(hipMemcpyAsync(g->host_counts + 1 + j, g->block_sums + n_train_blocks, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        return WOST_OK;
    };
    // trainStep (:618-668): up to batches_per_spp Adam steps on the n entries of the training set `ts`
    auto train_passes = [&](size_t n, hipStream_t st, const TrainSet &ts) -> int {
        g->last_train_n = (uint32_t)n;
        train_samples += n;
        const size_t bs = (size_t)s.batch_size;
        size_t n_batches = std::min<size_t>(n / bs + 1, (size_t)s.batches_per_spp);
        if (g->sync) {
            // shared network: every rank must take the same number of steps -- the smallest
            // number of full batches any rank has
            size_t usable = 0;
            for (size_t it = 0; it < n_batches; ++it) {
                size_t local = std::min(n - it * bs, bs);
                local -= local % 128;
                if (local < (size_t)s.min_batch_size) break;
                ++usable;
            }
            int64_t vmin = (int64_t)usable;
            if (g->sync(g->sync_user, WOST_SYNC_MIN_I64_HOST, &vmin, 1) != 0)
                return set_error(WOST_ERR_DEVICE, "sync callback failed (batch count)");
            n_batches = (size_t)std::max<int64_t>(vmin, 0);
        }
        for (size_t it = 0; it < n_batches; ++it) {
            size_t local = std::min(n - it * bs, bs);
            local -= local % 128;
            if (local < (size_t)s.min_batch_size) break;
            const size_t o = it * bs;
            float *raw = nullptr, *dl = nullptr;
            int rc = net_forward_train_dev(g->net, ts.xy + 2 * o, (int)local, st, &raw, &dl);
            if (rc != WOST_OK) return rc;
            launch_vmm_loss_gradients(st, raw, ts.dir + 2 * o, ts.li + o, ts.pdf + o, ts.onn + o, ts.nrm + 2 * o, (int)local, s.loss_scale, dl, nullptr);
            ++launches;      // the loss-gradient kernel; the network's own launches are counted by the network
            rc = net_backward_update_dev(g->net, ts.xy + 2 * o, (int)local, s.loss_scale, g->sync ? 0 : 1, st);
            if (rc != WOST_OK) return rc;
            if (g->sync) {
                // sum the fixed-point gradients of all ranks (integer sums: the same network
                // everywhere, bit for bit), then step
                G_TRY(hipStreamSynchronize(st));
                uint64_t count = 0;
                void *gbuf = net_gradient_buffer(g->net, &count);
                if (g->sync(g->sync_user, WOST_SYNC_SUM_I64_DEVICE, gbuf, count) != 0)
                    return set_error(WOST_ERR_DEVICE, "sync callback failed (gradient all-reduce)");
                rc = net_apply_update_dev(g->net, s.loss_scale, st);
                if (rc != WOST_OK) return rc;
            }
        }
        return WOST_OK;
    };

    bool d0_valid = false;
    // ctor state (integrator.cu:1158-1160), prepareSolve (:125-126)
    bool training = true;
    float uniform_fraction = s.uniform_fraction_training;
    int max_guided_depth = s.max_guided_depth_training;
    const bool dbg = std::getenv("WOST_GUIDED_DEBUG") != nullptr;

    // ---- opt-in training orders (never the parity mode) ---------------------------------------------------------------
    // "train_group" S: a training launch walks S samples of every pixel (each with its own record set), then the S training
    // passes follow.  "pipeline" 1: the passes of group g run on a second stream while group g + 1 walks with a frozen copy
    // of the weights that the passes of group g - 1 left.  The estimator is unbiased for ANY network state (a step's
    // direction and its one-sample MIS density come from the same weights, and a training record carries the density it was
    // drawn with): the only change is that the network a sample sees is a few training passes older.  Both need the fused
    // sample kernel; a solve with intermediate frames keeps the reference's order.
    const int n_trained = std::min(s.spp, s.train_spp_count);
    const bool reordered = fused && n_trained > 0 && !g->frame_fn && !dbg && (g->pipeline || g->train_group > 1);
    const int group = reordered ? std::min(std::max(g->train_group, 1), n_trained) : 1;
    // The groups grow with the training: the launch that starts at sample s covers min(S, max(1, s / 2)) samples -- never more than
    // half of what has been trained before it.  The network learns fastest from its first samples; with fixed groups of 16 the first
    // sixteen samples of a solve all walked with the untrained network (on config 4: rel-L2 against a 4096-sample field 0.0262, worse than
    // the uniform integrator's 0.0248; the reference's order 0.0244).
    auto group_at = [&](int first_sample) { return std::min(std::min(group, std::max(1, first_sample / 2)), n_trained - first_sample); };
    if (group > g->rec_sets) {
        // one record set per sample of a training launch (201 MB each at 1024^2: sized for 288 GB of HBM)
        float *rec = nullptr;
        uint32_t *cd = nullptr;
        // (the smaller set goes first: nothing on the device uses it between two solves, and group 16 would otherwise hold
        // 3.2 GB next to the 201 MB it replaces until the handle is destroyed)
        G_TRY(hipStreamSynchronize(stream));
        gfree_one(g, g->rec);
        gfree_one(g, g->cur_depth);
        g->rec = nullptr; g->cur_depth = nullptr; g->rec_sets = 0;
        G_TRY(galloc(g, &rec, (size_t)kMaxTrainDepth * kRecFields * (size_t)N * group));
        G_TRY(galloc(g, &cd, (size_t)N * group));
        g->rec = rec; g->cur_depth = cd; g->rec_sets = group;
        P.rec = g->rec; P.cur_depth = g->cur_depth;
    }
    P.rec_ld = (size_t)N * (size_t)g->rec_sets;
    int sample0 = 0;
    if (reordered && g->pipeline) {
        const size_t snap_bytes = net_snapshot_bytes(g->net);
        if (snap_bytes == 0) return set_error(WOST_ERR_UNSUPPORTED, "pipelined training needs a network image");
        if (!g->train_stream) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);       // lo = least urgent: the training fills what the walk leaves idle
            G_TRY(hipStreamCreateWithPriority(&g->train_stream, hipStreamNonBlocking, lo));
            G_TRY(hipEventCreateWithFlags(&g->ev_ts, hipEventDisableTiming));
            for (hipEvent_t &e : g->ev_train) G_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            G_TRY(g->train_events.init(64));
            for (void *&p : g->snap) {
                uint8_t *q = nullptr;
                G_TRY(galloc(g, &q, snap_bytes));
                p = q;
            }
        }
        // the passes of a group read their training sets while the next group walks over the record sets: one array set per sample
        while ((int)g->ts_more.size() + 1 < group) {
            TrainSet t{};
            const size_t M = (size_t)g->n_train_pixels * kMaxTrainDepth;
            G_TRY(galloc(g, &t.xy, 2 * M)); G_TRY(galloc(g, &t.dir, 2 * M)); G_TRY(galloc(g, &t.sol, 3 * M)); G_TRY(galloc(g, &t.li, M));
            G_TRY(galloc(g, &t.pdf, M)); G_TRY(galloc(g, &t.nrm, 2 * M)); G_TRY(galloc(g, &t.onn, M));
            g->ts_more.push_back(t);
        }
        auto ts_of = [&](int j) -> const TrainSet & { return j == 0 ? g->ts : g->ts_more[(size_t)j - 1]; };
        hipStream_t B = g->train_stream;
        // an error return below must not leave training passes in flight on the second stream behind the caller's back
        struct DrainOnError {
            hipStream_t s;
            bool armed;
            ~DrainOnError() { if (armed) (void)hipStreamSynchronize(s); }
        } drain{B, true};
        FusedNet Fs[2] = {F, F};
        for (int k = 0; k < 2; ++k) {
            int rc = net_snapshot_dev(g->net, g->snap[k], stream);      // both copies start as the weights the solve starts with
            if (rc != WOST_OK) return rc;
            bool half = false;
            HalfNetView hv{};
            F32NetView fv{};
            rc = net_snapshot_views(g->net, g->snap[k], &half, &hv, &fv);
            if (rc != WOST_OK || half != fused_half) return set_error(WOST_ERR_UNSUPPORTED, "pipelined training: no view of the network image");
            if (half) Fs[k].image = hv.image;
            else { Fs[k].frag32 = fv.frag; Fs[k].grid32 = fv.grid; }
        }
        std::vector<int> group_size;
        for (int first = 0; first < n_trained; first += group_size.back()) group_size.push_back(group_at(first));
        const int n_groups = (int)group_size.size();
        auto size_of = [&](int k) { return group_size[(size_t)k]; };
        P.training = 1; P.uniform_fraction = uniform_fraction; P.max_guided_depth = max_guided_depth; P.dbg = nullptr;
        auto walk = [&](int k) -> int {
            P.first_sample = k == 0;
            P.d0_valid = k > 0 ? 1 : 0;
            P.n_samples = size_of(k);
            return launch_walk(Fs[k & 1], stream);
        };
        auto train_sets = [&](int k) -> int {
            for (int j = 0; j < size_of(k); ++j) {
                const int rc = enqueue_train_set(stream, j, ts_of(j));
                if (rc != WOST_OK) return rc;
            }
            return WOST_OK;
        };
        // (the training stream starts behind whatever the walk stream has been given so far)
        G_TRY(hipEventRecord(g->ev_ts, stream));
        G_TRY(hipStreamWaitEvent(B, g->ev_ts, 0));
        int rc = walk(0);
        if (rc == WOST_OK) rc = train_sets(0);
        if (rc != WOST_OK) return rc;
        G_TRY(hipEventRecord(g->ev_ts, stream));
        for (int k = 0; k < n_groups; ++k) {
            if (k + 1 < n_groups) {
                // walk(k + 1) reads snap[(k + 1) % 2] = the weights after the passes of group k - 1
                if (k >= 1) G_TRY(hipStreamWaitEvent(stream, g->ev_train[(k - 1) & 1], 0));
                rc = walk(k + 1);
                if (rc != WOST_OK) return rc;
            }
            G_TRY(hipEventSynchronize(g->ev_ts));          // the training sets of group k are complete; their sizes have arrived
            size_t n_of[16];
            for (int j = 0; j < size_of(k); ++j) n_of[j] = g->host_counts[1 + j];
            G_TRY(hipStreamWaitEvent(B, g->ev_ts, 0));
            const size_t ev = g->train_events.begin(B);
            for (int j = 0; j < size_of(k) && rc == WOST_OK; ++j) rc = train_passes(n_of[j], B, ts_of(j));
            if (rc == WOST_OK) rc = net_snapshot_dev(g->net, g->snap[k & 1], B);
            g->train_events.end(ev, B);
            if (rc != WOST_OK) return rc;
            G_TRY(hipEventRecord(g->ev_train[k & 1], B));
            if (k + 1 < n_groups) {
                G_TRY(hipStreamWaitEvent(stream, g->ev_train[k & 1], 0));     // the arrays of the training sets are free again
                rc = train_sets(k + 1);
                if (rc != WOST_OK) return rc;
                G_TRY(hipEventRecord(g->ev_ts, stream));
            }
        }
        // what follows (the guiding phase, the resolve) reads the network's own images
        G_TRY(hipStreamWaitEvent(stream, g->ev_train[(n_groups - 1) & 1], 0));
        G_TRY(hipStreamSynchronize(B));
        drain.armed = false;
        train_ms += g->train_events.drain();
        sample0 = n_trained;
        d0_valid = true;
    }

    for (int sample = sample0; sample < s.spp; ++sample) {
        if (sample >= s.train_spp_count && training) {       // :991-996
            training = false;
            uniform_fraction = s.uniform_fraction_guiding;
            max_guided_depth = s.max_guided_depth_guiding;
        }
        P.training = training ? 1 : 0;
        P.uniform_fraction = uniform_fraction;
        P.first_sample = sample == 0;
        int n_run = 1;      // samples this iteration covers
        // both paths give the same results and could alternate within a solve; the fused one is used whenever it exists
        const bool fused_now = fused;
        if (fused_now) {
            if (training) {
                n_run = group_at(sample);      // 1 unless "train_group" asks for more
            } else {
                // nothing is trained between the remaining samples: one launch runs them all, up to the next
                // intermediate frame the caller asked for
                int last = s.spp - 1;
                if (g->frame_fn) {
                    for (int j = sample; j < s.spp; ++j) {
                        const bool by_spp = g->frame_spp_every > 0 && j % g->frame_spp_every == 0 && j < g->frame_spp_until;
                        const bool by_time = g->frame_time_every > 0 && j % g->frame_time_every == 0;
                        if (by_spp || by_time) { last = j; break; }
                    }
                }
                n_run = last - sample + 1;
                // a pixel's samples run one after the other in one lane, so the last pixels taken keep a few lanes busy for
                // n_run walks while the chip idles: bounded launches keep that tail short against the launch itself
                int cap = 64;
                if (const char *w = std::getenv("WOST_GUIDED_SAMPLES_PER_LAUNCH")) cap = std::max(1, std::atoi(w));
                n_run = std::min(n_run, cap);
            }
            n_run = std::min(n_run, 0xffff);      // a pixel's state word counts the samples of a launch in 16 bits (arrived << 16 | complete)
            n_run = (int)std::min<uint64_t>((uint64_t)n_run, std::max<uint64_t>(1, 0xffffffffull / (uint64_t)std::max(N, 1)));      // (items of a launch are counted in 32 bits)
            P.n_samples = n_run;
            P.d0_valid = d0_valid ? 1 : 0;
            P.max_guided_depth = max_guided_depth;
            P.dbg = dbg ? g->dbg : nullptr;
            if (dbg) {
                const unsigned long long init[4] = {~0ull, ~0ull, 0ull, 0ull};
                G_TRY(hipMemcpyAsync(g->dbg, init, sizeof(init), hipMemcpyHostToDevice, stream));
                G_TRY(hipStreamSynchronize(stream));
            }
            const int rcw = launch_walk(F, stream);
            if (rcw != WOST_OK) return rcw;
            if (dbg) {
                unsigned long long t[4];
                G_TRY(hipMemcpy(t, g->dbg, sizeof(t), hipMemcpyDeviceToHost));
                std::fprintf(stderr, "[fused sample %d x%d] first wave out of pixels at %.1f us, last at %.1f us, end %.1f us\n", sample, n_run,
                             (double)(t[1] - t[0]) / 100.0, (double)(t[2] - t[0]) / 100.0, (double)(t[3] - t[0]) / 100.0);
            }
            sample += n_run - 1;     // the index of the last sample this launch has run
            d0_valid = true;
        }
        int cur = 0;     // queue holding the evaluation points of this depth
        if (!fused_now) {
        G_TRY(hipMemsetAsync(g->counts + cur, 0, sizeof(uint32_t), stream));
        P.out = g->q[cur]; P.count_out = g->counts + cur;
        hipLaunchKernelGGL(begin_sample_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, P);
        ++launches;
        }
        // The queue only shrinks from depth to depth, so ANY earlier size bounds it.  The sizes come
        // back through pinned memory without the host waiting for them (every kernel reads the exact
        // size on the device); the launches of a sample are issued back to back.
        uint32_t n_cur = (uint32_t)N;
        int polled = -1;                 // last depth whose queue size has arrived
        for (int depth = 0; depth < s.max_depth && !fused_now; ++depth) {
            const int nxt = cur ^ 1;
            P.depth = depth;
            P.guiding = depth < max_guided_depth ? 1 : 0;
            P.last_depth = depth == s.max_depth - 1;
            P.in = g->q[cur]; P.count_in = g->counts + cur;
            P.out = g->q[nxt]; P.count_out = g->counts + nxt;
            const unsigned grid = (n_cur + 255) / 256;
            if (!P.guiding) {
                // no network from here on: every remaining walker runs to its end in one launch
#define LAUNCH_TAIL(E, T)                                                                                            \
    do {                                                                                                             \
        if (v.src.rgb) hipLaunchKernelGGL((tail_kernel<E, T, true>), dim3(grid), dim3(256), lds, stream, P);           \
        else hipLaunchKernelGGL((tail_kernel<E, T, false>), dim3(grid), dim3(256), lds, stream, P);                     \
    } while (0)
                if (emissive) { if (tree) LAUNCH_TAIL(true, true); else LAUNCH_TAIL(true, false); }
                else          { if (tree) LAUNCH_TAIL(false, true); else LAUNCH_TAIL(false, false); }
#undef LAUNCH_TAIL
                ++launches;
                G_TRY(hipGetLastError());
                break;
            }
            G_TRY(hipMemsetAsync(g->counts + nxt, 0, sizeof(uint32_t), stream));
#define LAUNCH_SEP(E, T)                                                                                             \
    do {                                                                                                             \
        if (v.src.rgb) hipLaunchKernelGGL((separate_kernel<E, T, true>), dim3(grid), dim3(256), lds, stream, P);       \
        else hipLaunchKernelGGL((separate_kernel<E, T, false>), dim3(grid), dim3(256), lds, stream, P);                 \
    } while (0)
            if (emissive) { if (tree) LAUNCH_SEP(true, true); else LAUNCH_SEP(true, false); }
            else          { if (tree) LAUNCH_SEP(false, true); else LAUNCH_SEP(false, false); }
#undef LAUNCH_SEP
            ++launches;
            // the out-of-shell queue is the input of the network and of the sampling kernel
            G_TRY(hipMemcpyAsync(g->depth_counts + depth, g->counts + nxt, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            G_TRY(hipEventRecord(g->depth_events[depth], stream));
            while (polled < depth && hipEventQuery(g->depth_events[polled + 1]) == hipSuccess) n_cur = g->depth_counts[++polled];
            if (polled == depth && n_cur == 0) break;      // known to be empty
            const uint32_t n_out = n_cur;                  // an upper bound
            if (P.guiding) {
                const size_t ev = g->net_events.begin(stream);
                int rc = net_inference_dev(g->net, g->net_in, g->counts + nxt, (int)n_out, g->net_out, true, stream, (size_t)N);
                g->net_events.end(ev, stream);
                if (rc != WOST_OK) return rc;      // (counted by the network)
            }
            P.in = g->q[nxt]; P.count_in = g->counts + nxt;
            const unsigned grid2 = (n_out + 255) / 256;
            if (tree) hipLaunchKernelGGL((sample_kernel<true>), dim3(grid2), dim3(256), lds, stream, P);
            else hipLaunchKernelGGL((sample_kernel<false>), dim3(grid2), dim3(256), lds, stream, P);
            ++launches;
            G_TRY(hipGetLastError());
            cur = nxt;
            n_cur = n_out;
        }
        // ---- trainStep (:618-668) ----
        if (training) {
            G_TRY(hipStreamSynchronize(stream));     // the walk phase ends here; train_ms counts the training only
            const auto t0 = std::chrono::high_resolution_clock::now();
            for (int j = 0; j < n_run; ++j) {        // (one record set per sample of the launch; one unless "train_group" is set)
                int rc = enqueue_train_set(stream, j, g->ts);
                if (rc != WOST_OK) return rc;
                G_TRY(hipStreamSynchronize(stream));
                rc = train_passes((size_t)g->host_counts[1 + j], stream, g->ts);
                if (rc != WOST_OK) return rc;
            }
            G_TRY(hipStreamSynchronize(stream));
            train_ms += std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
        }
        // intermediate frames (reference integrator.cu:1049-1081)
        if (g->frame_fn) {
            const bool by_spp = g->frame_spp_every > 0 && sample % g->frame_spp_every == 0 && sample < g->frame_spp_until;
            const bool by_time = g->frame_time_every > 0 && sample % g->frame_time_every == 0;
            if (by_spp || by_time) {
                hipLaunchKernelGGL(resolve_kernel, dim3((3 * N + 255) / 256), dim3(256), 0, stream, g->sol, N, (float)(sample + 1),
                                   g->field);
                std::vector<float> frame((size_t)N * 3);
                G_TRY(hipMemcpyAsync(frame.data(), g->field, frame.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
                G_TRY(hipStreamSynchronize(stream));
                const double ms =
                    std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_start).count();
                if (by_spp && g->frame_fn(g->frame_user, 0, sample, ms, frame.data()) != 0)
                    return set_error(WOST_ERR_INVALID, "frame callback asked to stop");
                if (by_time && g->frame_fn(g->frame_user, 1, sample, ms, frame.data()) != 0)
                    return set_error(WOST_ERR_INVALID, "frame callback asked to stop");
            }
        }
    }
    hipLaunchKernelGGL(resolve_kernel, dim3((3 * N + 255) / 256), dim3(256), 0, stream, g->sol, N, (float)s.spp, g->field);
    G_TRY(hipGetLastError());
    if (field_host) G_TRY(hipMemcpyAsync(field_host, g->field, (size_t)N * 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
    if (field_dev) G_TRY(hipMemcpyAsync(field_dev, g->field, (size_t)N * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream));
    std::vector<GStatsDev> copies(kStatCopies);
    G_TRY(hipMemcpyAsync(copies.data(), g->stats, kStatCopies * sizeof(GStatsDev), hipMemcpyDeviceToHost, stream));
    G_TRY(hipStreamSynchronize(stream));
    GStatsDev hs{};
    for (const GStatsDev &c : copies) {
        hs.steps += c.steps; hs.started += c.started; hs.absorbed += c.absorbed;
        hs.truncated += c.truncated; hs.nhits += c.nhits; hs.guided += c.guided; hs.net_points += c.net_points;
    }
    const double net_infer_ms = g->net_events.drain();
    net_set_gradient_divisor(g->net, 1.0f);      // the rank count belongs to this solve: a later wost_net_train_step on the handle is a plain step
    if (stats) {
        *stats = wost_guided_stats{};
        stats->walk_steps = hs.steps; stats->walks_started = hs.started; stats->walks_absorbed = hs.absorbed;
        stats->walks_truncated = hs.truncated; stats->neumann_hits = hs.nhits; stats->guided_steps = hs.guided;
        stats->train_samples = train_samples;
        stats->optimizer_steps = (uint64_t)(net_optimizer_steps(g->net) - opt_before);
        stats->train_ms = train_ms;
        stats->kernel_launches = launches + (uint32_t)(net_launch_count(g->net) - net_launches_before);
        stats->net_points = hs.net_points;
        stats->reserved = g->last_train_offset;
        stats->net_infer_ms = net_infer_ms;
        stats->solve_ms =
            std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_start).count();
    }
    return WOST_OK;
}

extern "C" {

int wost_guided_solve(wost_guided_handle g, float *field_rgb, wost_guided_stats *stats)
{
    if (!g || !field_rgb) return set_error(WOST_ERR_INVALID, "null argument");
    return run_guided(g, 0, 1, field_rgb, nullptr, stats);
}

int wost_guided_solve_sharded(wost_guided_handle g, int32_t shard_index, int32_t shard_count, float *field_rgb_dev,
                              wost_guided_stats *stats)
{
    if (!g || !field_rgb_dev) return set_error(WOST_ERR_INVALID, "null argument");
    if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count) return set_error(WOST_ERR_INVALID, "bad shard");
    return run_guided(g, shard_index, shard_count, nullptr, field_rgb_dev, stats);
}

}  // extern "C"
