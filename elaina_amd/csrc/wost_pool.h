// wost_pool.h -- the walker-pool walk kernel (gfx950).  Included by wost_hip.hip after the
// queue / Lane / step_finish definitions.
//
// The round kernel keeps one walker per lane in registers; its lanes are in different phases
// of their work (visiting an inner LBVH node, visiting a leaf, finishing a step), so a wave
// pays for every phase on every trip while only part of its lanes use it (measured: 37 %
// VALU lane utilisation, kernel bound by VALU issue).  Here a wave owns a POOL of 64*K
// walkers whose state lives in LDS, and lanes are workers: every trip the wave counts the
// walkers ready for each kind of work with ballots, picks ONE kind, compacts up to 64 ready
// walkers of that kind onto its lanes (ballot + mbcnt prefix, the ids travel through LDS),
// runs the one lean body that kind needs, and writes the walkers back.  Walkers that have
// used up their pixel are replaced on the spot from the global queue (one atomic per wave),
// so a single launch walks the whole frame and there are no end-of-round drains.
#pragma once

namespace wost {

struct PoolParams {
    DevMesh dm, nm;
    DevSettings st;
    WalkQueue q;          // walker records written by init_kernel (global memory)
    uint32_t n_walkers;   // number of records
    uint32_t *next;       // global fetch cursor (zeroed by the host)
    float *field;         // solution/spp written at field[(pix - field_base) * 3]
    int32_t field_base;
    StatsDev *stats;
    int32_t stack_words;  // LDS stack entries per walker (deeper entries spill to `spill`)
    uint32_t *spill;      // [spill_words][gridDim.x * NW] overflow columns in global memory
    int32_t step_weight;  // a step batch is preferred when n_step * step_weight >= 8 * max(n_inner, n_leaf)
};

enum : uint32_t { W_EMPTY = 0, W_INNER = 1, W_LEAF = 2, W_STEP = 3 };

// LDS image of a pool of NW walkers (SoA over the walker index).  Only what the traversal
// trips and the scheduler touch lives here; the rest of a walker (Neumann normal, throughput,
// running solution, evaluation point, cached depth-0 query) stays in its global queue record
// and is gathered by the step trips.
struct PoolLds {
    uint32_t *state;
    float *px, *py;
    uint32_t *lvlpos;   // level << 28 | pos
    int32_t *sp;        // entries on the walker's stack
    float *bd2;         // best squared distance of the running / finished query
    int32_t *bslot;     // its slot; doubles as the temporal hint of the next query
    uint32_t *rlo, *rhi, *meta;
    uint32_t *gslot;    // index of the walker's record in the global queue
    uint32_t *list;     // 64 entries: compaction scratch
    uint32_t *stack;    // stack_words * NW
};

__device__ __forceinline__ PoolLds carve_pool(uint32_t *base, int NW)
{
    PoolLds p;
    uint32_t *c = base;
    auto take = [&](int n) { uint32_t *r = c; c += n; return r; };
    p.state = take(NW);
    p.px = (float *)take(NW); p.py = (float *)take(NW);
    p.lvlpos = take(NW); p.sp = (int32_t *)take(NW);
    p.bd2 = (float *)take(NW); p.bslot = (int32_t *)take(NW);
    p.rlo = take(NW); p.rhi = take(NW); p.meta = take(NW);
    p.gslot = take(NW);
    p.list = take(64);
    p.stack = c;
    return p;
}

static constexpr int kPoolWordsPerWalker = 11;  // + stack_words

template <bool NEUMANN_EMISSIVE, bool NEUMANN_TREE, int K>
__global__ __launch_bounds__(64, 4) void walk_pool_kernel(PoolParams P)
{
    constexpr int NW = 64 * K;
    extern __shared__ uint32_t lds_pool[];
    const PoolLds L = carve_pool(lds_pool, NW);
    const int lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const bool has_d = P.dm.n_segs > 0;
    const int levels = P.dm.levels;
    const uint32_t spp = (uint32_t)P.st.spp;
    // the pool kernel runs the whole solve in one launch: full-width counters
    uint32_t n_steps = 0, n_started = 0, n_absorbed = 0, n_truncated = 0, n_hits = 0, n_inner = 0, n_leaf = 0;
    uint32_t trav_trips = 0, step_trips = 0;
    int max_sp = 0;
    bool exhausted = false;  // wave-uniform: the global queue has no more walkers

    // Start the next walk step of walker w: count it, then either use the cached depth-0
    // answer of the pixel or seed a traversal with the temporal hint.
    auto begin_step = [&](int w, uint32_t depth, float px, float py, int32_t hint, uint32_t g) {
        n_steps++;
        n_started += (depth == 0) ? 1u : 0u;
        if (!has_d || depth == 0) {
            L.bd2[w] = P.q.d0_d2[g];
            L.bslot[w] = P.q.d0_slot[g];
            L.state[w] = W_STEP;
        } else {
            const Closest seed = slot_candidate(P.dm, hint, px, py);
            L.bd2[w] = seed.d2;
            L.bslot[w] = seed.slot;
            L.lvlpos[w] = 0u;
            L.sp[w] = 0;
            L.state[w] = W_INNER;  // the root is an inner node (levels >= 1)
        }
    };

    // Give pool slot w (owned by this lane for the purpose of the request) a fresh walker
    // from the global queue; all lanes call this together, `need` says who wants one.
    auto refill = [&](bool need, int w) {
        const unsigned long long bal = __ballot(need && !exhausted);
        if (bal == 0ull) return;
        const uint32_t cnt = (uint32_t)__popcll(bal);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(P.next, cnt);
        base = __shfl(base, 0);
        if (base + cnt >= P.n_walkers) exhausted = true;
        if (need) {
            const uint32_t g = base + (uint32_t)__popcll(bal & lt_mask);
            if (g < P.n_walkers) {
                const uint64_t r = P.q.rng[g];
                const uint32_t meta = P.q.meta[g];
                const float x = P.q.px[g], y = P.q.py[g];
                L.gslot[w] = g;
                L.rlo[w] = (uint32_t)r; L.rhi[w] = (uint32_t)(r >> 32);
                L.meta[w] = meta;
                L.px[w] = x; L.py[w] = y;
                begin_step(w, META_DEPTH(meta), x, y, P.q.hint[g], g);
            }
        }
    };

    for (int k = 0; k < K; ++k) L.state[lane + 64 * k] = W_EMPTY;
    __syncthreads();
    for (int k = 0; k < K; ++k) refill(true, lane + 64 * k);

    for (;;) {
        __syncthreads();  // pool updates of the previous trip are visible to every lane
        // ---- census: how many walkers are ready for each kind of work ----
        uint32_t own[K];
        int n_inner = 0, n_leaf = 0, n_step = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            own[k] = L.state[lane + 64 * k];
            n_inner += __popcll(__ballot(own[k] == W_INNER));
            n_leaf += __popcll(__ballot(own[k] == W_LEAF));
            n_step += __popcll(__ballot(own[k] == W_STEP));
        }
        if (n_inner + n_leaf + n_step == 0) break;  // pool empty and nothing left to fetch
        uint32_t kind;
        if (n_step * P.step_weight >= 8 * max(n_inner, n_leaf)) kind = W_STEP;
        else kind = (n_inner >= n_leaf) ? W_INNER : W_LEAF;
        // ---- compaction: up to 64 ready walkers of that kind, one per lane ----
        int base = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long m = __ballot(own[k] == kind);
            if (own[k] == kind) {
                const int r = base + __popcll(m & lt_mask);
                if (r < 64) L.list[r] = (uint32_t)(lane + 64 * k);
            }
            base += __popcll(m);
        }
        const int n_sel = min(base, 64);
        __syncthreads();
        const int w = (lane < n_sel) ? (int)L.list[lane] : -1;

        if (kind != W_STEP) {
            ++trav_trips;
            if (w >= 0) {
                const uint32_t lp = L.lvlpos[w];
                Trav T;
                T.level = (int32_t)(lp >> 28);
                T.pos = (int32_t)(lp & 0x0fffffffu);
                T.sp = L.sp[w];
                T.best = Closest{L.bd2[w], L.bslot[w]};
                T.best_orig = -1;
                const float qx = L.px[w], qy = L.py[w];
                const SplitColumn stk{L.stack + w, (uint32_t)NW, P.stack_words,
                                      P.spill + (size_t)blockIdx.x * NW + w, (uint32_t)(gridDim.x * NW)};
                bool more;
                if (kind == W_INNER) {
                    n_inner++;
                    more = trav_visit<false, 1>(P.dm, qx, qy, T, stk);
                } else {
                    n_leaf++;
                    more = trav_visit<false, 2>(P.dm, qx, qy, T, stk);
                }
                L.lvlpos[w] = ((uint32_t)T.level << 28) | (uint32_t)T.pos;
                L.sp[w] = T.sp;
                max_sp = max(max_sp, T.sp);
                L.bd2[w] = T.best.d2;
                L.bslot[w] = T.best.slot;
                L.state[w] = !more ? W_STEP : (T.level == levels ? W_LEAF : W_INNER);
            }
        } else {
            ++step_trips;
            bool want_refill = false;
            if (w >= 0) {
                const uint32_t g = L.gslot[w];
                const uint32_t meta = L.meta[w];
                Lane A;
                A.px = L.px[w]; A.py = L.py[w];
                A.rng.state = ((uint64_t)L.rhi[w] << 32) | L.rlo[w];
                A.rng.inc = 1;
                A.sample = META_SAMPLE(meta); A.depth = META_DEPTH(meta); A.on_n = META_ONN(meta) != 0;
                A.nx = 0.0f; A.ny = 0.0f;
                if (A.on_n) { A.nx = P.q.nx[g]; A.ny = P.q.ny[g]; }
                A.thp = P.q.thp[g];
                const float sr0 = P.q.sr[g], sg0 = P.q.sg[g], sb0 = P.q.sb[g];
                const float thp0 = A.thp;
                A.sr = sr0; A.sg = sg0; A.sb = sb0;
                A.hint = 0;
                const Closest cp{L.bd2[w], L.bslot[w]};
                // the walker's traversal stack is idle between queries: the Neumann-side tree
                // queries of the step logic use it
                const SplitColumn stk{L.stack + w, (uint32_t)NW, P.stack_words,
                                      P.spill + (size_t)blockIdx.x * NW + w, (uint32_t)(gridDim.x * NW)};
                const uint32_t status = step_finish<NEUMANN_EMISSIVE, NEUMANN_TREE>(P.dm, P.nm, P.st, A, cp, stk);
                const bool ended = (status & STEP_ENDED) != 0u;
                n_absorbed += (status >> 1) & 1u;
                n_truncated += (status >> 2) & 1u;
                n_hits += (status >> 3) & 1u;
                bool alive = true;
                if (ended) {
                    // next sample of this pixel starts right away (generateEvaluationPoints,
                    // reference integrator.cu:90-99 + workqueue.h:99-110)
                    A.sample++;
                    A.px = P.q.x0[g]; A.py = P.q.y0[g];
                    A.depth = 0; A.on_n = false; A.nx = 0.0f; A.ny = 0.0f;
                    A.thp = 1.0f;
                    alive = A.sample < spp;
                }
                if (alive) {
                    L.px[w] = A.px; L.py[w] = A.py;
                    L.rlo[w] = (uint32_t)A.rng.state; L.rhi[w] = (uint32_t)(A.rng.state >> 32);
                    L.meta[w] = META_PACK(A.sample, A.depth, A.on_n ? 1 : 0);
                    if (A.on_n) { P.q.nx[g] = A.nx; P.q.ny[g] = A.ny; }
                    if (A.thp != thp0) P.q.thp[g] = A.thp;
                    // bit-wise compare: a contribution of -0.0f changes nothing, NaN must stick
                    if (__float_as_uint(A.sr) != __float_as_uint(sr0) || __float_as_uint(A.sg) != __float_as_uint(sg0) ||
                        __float_as_uint(A.sb) != __float_as_uint(sb0)) {
                        P.q.sr[g] = A.sr; P.q.sg[g] = A.sg; P.q.sb[g] = A.sb;
                    }
                    // step_finish left the finished query's slot in A.hint when there is a
                    // Dirichlet boundary; it seeds the next query
                    begin_step(w, A.depth, A.px, A.py, has_d ? A.hint : 0, g);
                } else {
                    // resolve the finished pixel (reference integrator.cu:616-620)
                    float *f = P.field + 3 * (size_t)((int32_t)P.q.pix[g] - P.field_base);
                    const float fs = (float)P.st.spp;
                    f[0] = A.sr / fs; f[1] = A.sg / fs; f[2] = A.sb / fs;
                    L.state[w] = W_EMPTY;
                    want_refill = true;
                }
            }
            refill(want_refill, w);
        }
    }

    // ---- statistics: wave reduction, one atomic per counter per wave ----
    uint32_t v[7] = {n_steps, n_started, n_absorbed, n_truncated, n_hits, n_inner, n_leaf};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        uint32_t x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        v[k] = x;
    }
    if (lane == 0) {
        if (v[0]) atomicAdd(&my_stats(P.stats)->steps, (unsigned long long)v[0]);
        if (v[1]) atomicAdd(&my_stats(P.stats)->started, (unsigned long long)v[1]);
        if (v[2]) atomicAdd(&my_stats(P.stats)->absorbed, (unsigned long long)v[2]);
        if (v[3]) atomicAdd(&my_stats(P.stats)->truncated, (unsigned long long)v[3]);
        if (v[4]) atomicAdd(&my_stats(P.stats)->nhits, (unsigned long long)v[4]);
        if (v[5]) atomicAdd(&my_stats(P.stats)->inner_visits, (unsigned long long)v[5]);
        if (v[6]) atomicAdd(&my_stats(P.stats)->leaf_visits, (unsigned long long)v[6]);
        atomicAdd(&my_stats(P.stats)->trav_trips, (unsigned long long)trav_trips);
        atomicAdd(&my_stats(P.stats)->step_trips, (unsigned long long)step_trips);
    }
    for (int off = 32; off > 0; off >>= 1) max_sp = max(max_sp, __shfl_down(max_sp, off));
    if (lane == 0) atomicMax(&my_stats(P.stats)->max_stack, (unsigned long long)(max_sp));
}

}  // namespace wost
