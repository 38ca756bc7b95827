// wost_guided3.hip -- GuidedIntegrator<3> on MI355X behind wost3_guided_* of include/wost.h.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/wost.h"
#include "wost_internal3.h"
#include "wost_net_device.h"
#include "wost_vmm3_device.h"

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
    // the launches per depth: the walkers that are still alive, compacted (written by g3_sample_kernel for the next depth's
    // g3_separate_kernel / the tail; l_pid == nullptr: every pixel in pixel order, as at depth 0 and in g3_fused_kernel)
    const uint32_t *l_pid, *l_count;
    uint32_t *l_next_pid, *l_next_count;
    float *net_in, *net_out;
    GStats3Dev *stats;
    int32_t training, train_offset, train_stride, max_train_depth;
    int32_t depth, guiding, first_sample, stack_stride;
    int32_t max_guided_depth;      // (g3_fused_kernel: the depths below it ask the network)
    float uniform_fraction;
    // the tree queries of a wave's walkers through its task pools (closest_triangle_pool & co.): pool_cap tasks per pool and wave,
    // pool_offset words into the block's LDS (behind the stack columns); pool_cap = 0: one descent per thread
    int32_t pool_cap, pool_offset;
    // small frames: one walker per 2^lane_shift lanes of the walk kernels -- a frame of 256^2 fills a quarter of the chip's lanes with
    // one walker per lane, and a wave's 64 queries through its pools take as long as they take; spread over more waves the same
    // queries run on more CUs at once (the lanes without a walker work on the other lanes' tree nodes)
    int32_t lane_shift;
};

// the walker (pixel or queue entry) of this thread, or -1
__device__ __forceinline__ int g3_item(const G3Params &P)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    return (tid & ((1u << P.lane_shift) - 1u)) == 0u ? (int)(tid >> P.lane_shift) : -1;
}

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
    const int i = g3_item(P);
    int p = i;
    bool live;
    if (P.l_pid) {
        live = i >= 0 && (uint32_t)i < *P.l_count;
        p = live ? (int)P.l_pid[i] : 0;
    } else {
        live = p >= 0 && p < P.n_pixels && P.state[p] == 1;
    }
    V3 x;
    const bool keep = g3_separate_body<EMISSIVE, NTREE, SOURCE>(P, P.depth, live ? p : 0, live, W, stk, x);
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
__device__ __forceinline__ bool g3_sample_body(const G3Params &P, int depth, bool guiding, uint32_t pid, bool live, const float *raw, const WavePool3 &W,
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
    return moved;      // the walker goes on to the next depth
}

template <bool NTREE>
__global__ __launch_bounds__(256) void g3_sample_kernel(G3Params P)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk{lds_stack + threadIdx.x, (uint32_t)P.stack_stride};
    const WavePool3 W = g3_pools(P, lds_stack);
    const int i = g3_item(P);
    const bool live = i >= 0 && (uint32_t)i < *P.q_count;
    const uint32_t pid = live ? P.q_pid[i] : 0u;
    const bool moved = g3_sample_body<NTREE>(P, P.depth, P.guiding != 0, pid, live, P.net_out + 41 * (size_t)(live ? i : 0), W, stk);
    if (P.l_next_pid) {
        const uint32_t s = block_push(moved, P.l_next_count);
        if (moved) P.l_next_pid[s] = pid;
    }
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
    const int i = g3_item(P);
    int p = i;
    bool mine = p >= 0 && p < P.n_pixels;
    if (P.l_pid) {      // the walkers that reached this depth, compacted
        mine = i >= 0 && (uint32_t)i < *P.l_count;
        p = mine ? (int)P.l_pid[i] : 0;
    }
    for (int depth = P.depth; depth < P.st.max_depth; ++depth) {
        const bool live = mine && P.state[p] == 1;
        if (!__ballot(live)) break;       // (wave-uniform: the queries are the wave's)
        V3 x;
        const bool keep = g3_separate_body<EMISSIVE, NTREE, SOURCE>(P, depth, mine ? p : 0, live, W, stk, x);
        g3_sample_body<NTREE>(P, depth, false, (uint32_t)(mine ? p : 0), keep, nullptr, W, stk);
    }
}

// ---- a whole sample in ONE launch ------------------------------------------------------------------------------------------------
// The launches per depth above cost a 16-sample solve ~700 launches, each as long as the slowest wave of its depth, and the walkers a
// round trip through the queue and the network's buffers.  Here a wave keeps its walkers from the evaluation point to the end of the
// walk -- g3_tail_kernel's loop from depth 0 -- and evaluates the network ITSELF for those that stay at a guided depth: the walkers
// that need it are ranked (ballot), sixteen of them make a unit of the matrix instructions, lane (i, g) of a unit interpolates the
// levels g and g + 4 of point i (f32_encode_level3, the arithmetic of net_forward_mfma_kernel), the features are turned to the operand
// layout through 2 KB of LDS, and the four matrices run on up to four units at once with every weight fragment fetched once
// (f32_mlp_units: the fragments stay in global memory / L2 -- the block's LDS belongs to the stack columns and the task pools of the
// tree queries).  The 41 outputs go to the pixel's row of net_out; the walker's lane reads them back after a work-group fence (one
// CU, one L1).  Same bodies, same draws, same matrices in the same order: bit-identical to the launches per depth
// (tests/test_guided_3d.py compares both with the oracle and with each other).  fp32 inference on the MFMA path only; any other
// network (half precision, a shape the MFMA kernels do not cover) keeps the launches per depth.
struct G3Net {
    const float *frag, *grid;
    uint32_t w_off[4];
    float scale[8];
    uint32_t res[8], off[9];
    int32_t xch_offset;       // words into the block's LDS: per wave [3][64] inputs, [64] pixel of rank r, [32][16] features of a unit
};
constexpr int kG3XchWords = 3 * 64 + 64 + 32 * 16;

template <int NU>
__device__ __forceinline__ void g3_net_units(const G3Params &P, const G3Net &F, const float *s_scale, const uint32_t *s_res, const uint32_t *s_off, int u0, int n_need,
                                             const float *xin, const uint32_t *xpid, float *ubuf)
{
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    float b[NU][16];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int q = 16 * (u0 + u) + li;
        const bool qv = q < n_need;
        const float qx = qv ? xin[q] : 0.5f, qy = qv ? xin[64 + q] : 0.5f, qz = qv ? xin[128 + q] : 0.5f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int lv = lg + 4 * h;
            float4 f = float4{0.0f, 0.0f, 0.0f, 0.0f};
            if (qv) f = f32_encode_level3(F.grid, s_scale[lv], s_res[lv], s_off[lv], s_off[lv + 1] - s_off[lv], qx, qy, qz);
            ubuf[(4 * lv + 0) * 16 + li] = f.x; ubuf[(4 * lv + 1) * 16 + li] = f.y; ubuf[(4 * lv + 2) * 16 + li] = f.z; ubuf[(4 * lv + 3) * 16 + li] = f.w;
        }
        wave_lds_fence();
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) b[u][s_] = qv ? ubuf[(4 * s_ + lg) * 16 + li] : 0.0f;
        wave_lds_fence();
    }
    // (the fragments do not change inside the launch: without this the compiler loads all 208 words of a lane once, in front of the depth
    // loop, and keeps them in accumulation registers -- one wave per SIMD, where the tree queries want their latency hidden)
    const float *frag = F.frag;
    asm volatile("" : "+s"(frag));
    f32_mlp_units<NU>(frag, F.w_off, lane, b);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int q = 16 * (u0 + u) + li;
        if (q < n_need) {
            float *o = P.net_out + 41 * (size_t)xpid[q];
#pragma unroll
            for (int rt = 0; rt < 3; ++rt)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (16 * rt + 4 * c + lg < 41) o[16 * rt + 4 * c + lg] = b[u][4 * rt + c];      // b[4 rt + c] = output 16 rt + 4 c + g
        }
    }
}

template <bool EMISSIVE, bool NTREE, bool SOURCE>
__global__ __launch_bounds__(256, 2) void g3_fused_kernel(G3Params P, G3Net F)
{
    extern __shared__ uint32_t lds_stack[];
    const LdsColumn stk{lds_stack + threadIdx.x, (uint32_t)P.stack_stride};
    const WavePool3 W = g3_pools(P, lds_stack);
    const int lane = threadIdx.x & 63;
    uint32_t *xw = lds_stack + F.xch_offset + (threadIdx.x >> 6) * kG3XchWords;
    float *xin = reinterpret_cast<float *>(xw);
    uint32_t *xpid = xw + 192;
    float *ubuf = reinterpret_cast<float *>(xw + 256);
    __shared__ float s_scale[8];
    __shared__ uint32_t s_res[8], s_off[9];
    if (threadIdx.x < 9) {
        s_off[threadIdx.x] = F.off[threadIdx.x];
        if (threadIdx.x < 8) {
            s_scale[threadIdx.x] = F.scale[threadIdx.x];
            s_res[threadIdx.x] = F.res[threadIdx.x];
        }
    }
    __syncthreads();
    const int p = g3_item(P);
    const bool mine = p >= 0 && p < P.n_pixels;
    for (int depth = 0; depth < P.st.max_depth; ++depth) {
        const bool live = mine && P.state[p] == 1;
        if (!__ballot(live)) break;       // (wave-uniform: the queries are the wave's)
        V3 x;
        const bool keep = g3_separate_body<EMISSIVE, NTREE, SOURCE>(P, depth, mine ? p : 0, live, W, stk, x);
        const bool guiding = depth < P.max_guided_depth;
        if (guiding) {
            const unsigned long long bal = __ballot(keep);
            if (bal) {
                const int n_need = __popcll(bal);
                if (keep) {
                    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
                    float in3[3];
                    g3_normalize(P.box, x, in3);
                    xin[rank] = in3[0]; xin[64 + rank] = in3[1]; xin[128 + rank] = in3[2];
                    xpid[rank] = (uint32_t)p;
                }
                wave_lds_fence();
                if (n_need > 16) {
                    g3_net_units<4>(P, F, s_scale, s_res, s_off, 0, n_need, xin, xpid, ubuf);
                } else {
                    g3_net_units<1>(P, F, s_scale, s_res, s_off, 0, n_need, xin, xpid, ubuf);
                }
                // the outputs were stored by other lanes of this wave: complete, and visible in this CU's L1, before they are read
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                wave_lds_fence();
            }
        }
        g3_sample_body<NTREE>(P, depth, guiding, (uint32_t)(mine ? p : 0), keep, P.net_out + 41 * (size_t)(mine ? p : 0), W, stk);
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

using namespace wost;

struct wost3_guided {
    int device = 0;
    wost3_handle scene = nullptr;
    wost3_guided_settings s{};
    wost_net_handle net = nullptr;
    G3Box box{};
    std::vector<void *> allocs;
    uint64_t *rng = nullptr;
    float *sol = nullptr, *rec = nullptr, *wx = nullptr, *wn = nullptr, *wthp = nullptr, *wrb = nullptr, *net_in = nullptr, *net_out = nullptr, *field = nullptr;
    uint32_t *cur_depth = nullptr, *q_pid = nullptr, *q_count = nullptr, *block_sums = nullptr, *l_pid = nullptr;
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
    // (counters: [1] the queue, [0] and [2] the live lists written at even / odd depths)
    P.q_pid = g->q_pid; P.q_count = g->q_count + 1; P.net_in = g->net_in; P.net_out = g->net_out; P.stats = g->stats;
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
    // walkers per lane of the walk kernels: spread out while all blocks of the frame are still resident at once (two blocks per CU: the
    // stack columns and the task pools take 46 to 64 KB of LDS).  Round 4 spread as far as 1.5 x three blocks per CU: a frame of 256^2
    // then ran its blocks in two rounds, each as long as its longest walk (the shell scene, 16 samples: 118 -> 89 ms with one round)
    uint64_t resident = 0;
    {
        int n_cus = 256;
        (void)hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, g->device);
        resident = (uint64_t)n_cus * 2 * 256;
        P.lane_shift = 0;
        while (P.pool_cap > 0 && P.lane_shift < 3 && ((uint64_t)N << (P.lane_shift + 1)) <= resident) ++P.lane_shift;
        if (const char *w = std::getenv("WOST3_G_SHIFT")) P.lane_shift = std::min(4, std::max(0, std::atoi(w)));
    }
    const unsigned grid_px = (unsigned)((((uint64_t)N << P.lane_shift) + 255) / 256);      // walk kernels (begin / train-set / resolve: one thread per pixel)
    const unsigned grid_pix = (unsigned)((N + 255) / 256);
    uint32_t launches = 0;
    uint64_t train_samples = 0;
    double train_ms = 0.0;
    const int opt_before = net_optimizer_steps(g->net);
    const uint64_t net_launches_before = net_launch_count(g->net);
    bool training = true;
    float uniform_fraction = s.uniform_fraction_training;
    int max_guided_depth = s.max_guided_depth_training;
    // One launch per sample (g3_fused_kernel) when the network offers its fp32 MFMA fragments and the frame has at most 1.5 walkers per
    // resident lane (196 608 on MI355X): such a solve is bound by its launches -- 30 per sample, each as long as its slowest wave.  A
    // larger frame is bound by throughput, and there the launches per depth win: all their kernels run on compacted lists of the walkers
    // that are left, the fused kernel's waves keep their dead lanes.  8 samples, 4 trained, fused / per depth in ms
    // (tools/probes/g3_forms_by_frame.py): icosphere 256^2 25 / 35, 362^2 30 / 44, 512^2 53 / 58, 724^2 120 / 91, 1024^2 212 / 158; shell
    // 44 / 52, 70 / 82, 132 / 105, 223 / 174, 457 / 301.  WOST3_G_FUSED=0 / 1: never / always.
    G3Net Fn{};
    bool fused = false;
    {
        F32NetView fv{};
        const char *env = std::getenv("WOST3_G_FUSED");
        const bool want = env ? env[0] != '0' : 2 * (uint64_t)N <= 3 * resident;
        if (want && net_f32_view3(g->net, &fv) == WOST_OK && fv.L.n_levels == 8 && fv.L.n_features == 4 && fv.L.n_out == 41) {
            Fn.frag = fv.frag; Fn.grid = fv.grid;
            for (int l = 0; l < 4; ++l) Fn.w_off[l] = fv.L.w_off[l];
            for (int l = 0; l < 8; ++l) { Fn.scale[l] = fv.L.scale[l]; Fn.res[l] = (uint32_t)fv.L.res[l]; }
            for (int l = 0; l <= 8; ++l) Fn.off[l] = fv.L.level_off[l];
            Fn.xch_offset = (int32_t)((lds + 3) / 4);
            fused = true;
        }
    }
    const size_t lds_fused = ((size_t)Fn.xch_offset + 4 * (size_t)kG3XchWords) * sizeof(uint32_t);
    if (fused && lds_fused > 48 * 1024) {
        // the stack columns and pools were sized against 64 KB, the exchange area comes on top (up to 76 KB): ask once, before the
        // sample loop, and take the launches per depth -- which need no more than `lds` -- when the device refuses
        hipError_t e = hipSuccess;
#define G3_ATTR(E, T) (has_src ? hipFuncSetAttribute(reinterpret_cast<const void *>(g3_fused_kernel<E, T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fused) \
                               : hipFuncSetAttribute(reinterpret_cast<const void *>(g3_fused_kernel<E, T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fused))
        if (ntree) e = emissive ? G3_ATTR(true, true) : G3_ATTR(false, true);
        else e = emissive ? G3_ATTR(true, false) : G3_ATTR(false, false);
#undef G3_ATTR
        if (e != hipSuccess) {
            (void)hipGetLastError();
            fused = false;
        }
    }
    for (int sample = 0; sample < s.spp; ++sample) {
        if (sample == s.train_spp_count) {      // :991-996
            training = false;
            uniform_fraction = s.uniform_fraction_guiding;
            max_guided_depth = s.max_guided_depth_guiding;
        }
        P.training = training ? 1 : 0; P.uniform_fraction = uniform_fraction; P.first_sample = sample == 0 ? 1 : 0;
        hipLaunchKernelGGL(g3_begin_kernel, dim3(grid_pix), dim3(256), 0, stream, P);
        ++launches;
        // No host round trip inside a sample: the launches of a depth are sized for the frame (their kernels read the true length of
        // the queue on the device; a block beyond it ends at once), and from the first depth that needs no network on, ONE launch
        // takes every walker that is left to its end (g3_tail_kernel).  A round trip per depth -- later one every fourth depth --
        // and the launch pairs of the late depths, whose few walkers cost a launch what its slowest tree query costs, were most
        // of the solve's wall time (about 2000 launches per 16-sample solve).
        const uint32_t n_upper = (uint32_t)N;
        if (fused) {
            P.max_guided_depth = max_guided_depth;
#define G3_FUSED(E, T)                                                                                                                      \
    do {                                                                                                                                    \
        auto kfn = has_src ? g3_fused_kernel<E, T, true> : g3_fused_kernel<E, T, false>;                                                      \
        hipLaunchKernelGGL(kfn, dim3(grid_px), dim3(256), lds_fused, stream, P, Fn);                                                         \
    } while (0)
            if (ntree) { if (emissive) G3_FUSED(true, true); else G3_FUSED(false, true); }
            else       { if (emissive) G3_FUSED(true, false); else G3_FUSED(false, false); }
#undef G3_FUSED
            ++launches;
        }
        for (int depth = 0; depth < s.max_depth && !fused; ++depth) {
            P.depth = depth; P.guiding = depth < max_guided_depth ? 1 : 0;
            // the walkers that reached this depth: the list the previous depth's sample kernel wrote (depth 0: every pixel)
            P.l_pid = depth > 0 ? g->l_pid + (size_t)((depth - 1) & 1) * N : nullptr;
            P.l_count = depth > 0 ? g->q_count + 2 * ((depth - 1) & 1) : nullptr;
            if (!P.guiding) {
#define G3_LAUNCH(K, E, T)                                                                                              \
    do {                                                                                                                \
        if (has_src) hipLaunchKernelGGL((K<E, T, true>), dim3(grid_px), dim3(256), lds, stream, P);                       \
        else hipLaunchKernelGGL((K<E, T, false>), dim3(grid_px), dim3(256), lds, stream, P);                              \
    } while (0)
                // (the tail starts on the compacted list of the walkers that are left and keeps them to their end: launches of a few depths
                // with the survivors compacted in between were measured and lose -- EXPERIMENTS 23 -- the tail waits for its steps, not for lanes)
                if (ntree) { if (emissive) G3_LAUNCH(g3_tail_kernel, true, true); else G3_LAUNCH(g3_tail_kernel, false, true); }
                else       { if (emissive) G3_LAUNCH(g3_tail_kernel, true, false); else G3_LAUNCH(g3_tail_kernel, false, false); }
                ++launches;
                break;
            }
            // the queue's counter and that of the live list this depth's sample kernel writes: two adjacent words
            W3_TRY(hipMemsetAsync(g->q_count + (depth & 1), 0, 2 * sizeof(uint32_t), stream));
            P.l_next_pid = g->l_pid + (size_t)(depth & 1) * N; P.l_next_count = g->q_count + 2 * (depth & 1);
            if (ntree) { if (emissive) G3_LAUNCH(g3_separate_kernel, true, true); else G3_LAUNCH(g3_separate_kernel, false, true); }
            else       { if (emissive) G3_LAUNCH(g3_separate_kernel, true, false); else G3_LAUNCH(g3_separate_kernel, false, false); }
            ++launches;
            {
                const int rc = net_inference_dev(g->net, g->net_in, P.q_count, (int)n_upper, g->net_out, true, stream, 0);
                if (rc != WOST_OK) return rc;
            }
            const unsigned grid_q = (unsigned)((((uint64_t)n_upper << P.lane_shift) + 255u) / 256u);
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
                launch_vmm3_loss_gradients(stream, out, g->t_dir + 3 * o, g->t_li + o, g->t_pdf + o, g->t_onn + o, g->t_nrm + 3 * o, (int)local, s.loss_scale,
                                           dl, (float *)nullptr);
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
    G3A(wthp, N); G3A(wrb, N); G3A(won, N); G3A(whint, N); G3A(hint0, N); G3A(q_pid, N); G3A(l_pid, 2 * N); G3A(q_count, 4); G3A(net_in, 3 * N); G3A(net_out, 41 * N);
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
