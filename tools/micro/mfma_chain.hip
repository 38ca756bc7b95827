// mfma_chain.hip -- standalone reproducer for EXPERIMENTS 17: the half-precision layer chain of net_forward_h_kernel
// (32 -> 64 -> 64 -> 64 -> 48, v_mfma_f32_16x16x32_f16, weights as fragments in LDS, two 16-point units per wave iteration)
// on inputs that have ONE right answer: inputs in {0, 1}, first-layer weights in {0, +-1/8}, the others in {0, +-1/16}.
// Every partial sum is then exact in fp32 whatever order the matrix instruction adds in (layer 1: multiples of 2^-3 up to 4;
// layer 2: multiples of 2^-7 up to 16, exact in f16 too; layer 3: multiples of 2^-11 up to 64; output: multiples of 2^-15 up to
// 256 = 23 bits), the f16 roundings between the layers are single roundings of exact values, and the host computes the same
// numbers in double precision.  Any word that differs from the host's is a wrong result of the device, not noise.  (A
// calibration launch -- one wave per SIMD, padded -- is compared with the host first and reported.)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/mfma_chain.hip -o tools/micro/mfma_chain
//   ./mfma_chain [launches per configuration = 20000] [points = 524288] [only this MODE = -1: all] [before every launch: 0 nothing, 1 lds, 2 vgpr, 3 code, 4 small, 5 idle, 6 idle + small] [1 = full-mantissa operands]
//
// Configurations: block size 1024 / 768 / 512 / 256 (four / three / two / one wave per SIMD; LDS padded to 150 KB so that one block owns a CU
// like the production kernel) x MODE:
//   0 plain          the chain as the compiler schedules it
//   1 settle         all accumulators of a layer through one asm statement with s_nop 15 behind the layer, operands held (production)
//   2 agpr           accumulators pinned to AGPRs through the layer ("+a"), read back with v_accvgpr_read
//   3 k16            v_mfma_f32_16x16x16_f16, twice as many instructions
//   4 schedbarrier   __builtin_amdgcn_sched_barrier(0) behind each layer, no idle states
//   5 nolds          weight fragments from global memory (L1/L2), nothing in LDS
//   6 tail           MODE 0 plus ~200 VALU instructions of transcendental arithmetic on the outputs (the shape of the fused-loss tail)
//   7 gather         MODE 0 plus sixteen data-dependent 8-byte LDS gathers per lane and tile over the whole 150 KB (the shape of the grid
//                    encoding: bank conflicts, LDS returns of uneven latency next to the matrix instructions)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <unistd.h>

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                                    \
        }                                                                               \
    } while (0)

constexpr int kSub = 2;
constexpr int kEnc = 32, kH = 64, kOutPad = 48;
constexpr int kWOff0 = 0, kWOff1 = kH * kEnc, kWOff2 = kWOff1 + kH * kH, kWOff3 = kWOff2 + kH * kH, kNMlp = kWOff3 + kOutPad * kH;   // 13 312

union Frag {
    uint2 u;
    h4_t h;
};

__device__ __forceinline__ f32x4_t mfma_k32(h4_t a0, h4_t a1, h4_t b0, h4_t b1, f32x4_t acc)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7),
                                                  acc, 0, 0, 0);
}

template <int MODE, int KT, int RT>
__device__ __forceinline__ void layer(const uint2 *wf, int lane, const h4_t (&b)[kSub][4], f32x4_t (&acc)[kSub][4])
{
#pragma unroll
    for (int u = 0; u < kSub; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[u][rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kt = 0; kt < KT; kt += 2)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            Frag a0, a1;
            a0.u = wf[(rt * KT + kt) * 64 + lane];
            a1.u = wf[(rt * KT + kt + 1) * 64 + lane];
#pragma unroll
            for (int u = 0; u < kSub; ++u) {
                if (MODE == 3) {
                    acc[u][rt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0.h, b[u][kt], acc[u][rt], 0, 0, 0);
                    acc[u][rt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1.h, b[u][kt + 1], acc[u][rt], 0, 0, 0);
                } else {
                    acc[u][rt] = mfma_k32(a0.h, a1.h, b[u][kt], b[u][kt + 1], acc[u][rt]);
                }
            }
        }
    if (MODE == 1) {
        if (RT == 4)
            asm volatile("s_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
        else
            asm volatile("s_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]));
        asm volatile("" ::"v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]));
    }
    if (MODE == 2) {
        if (RT == 4)
            asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]));
        else
            asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]));
    }
    if (MODE == 4) __builtin_amdgcn_sched_barrier(0);
}

// in: [unit][2 tiles][64 lanes] f16x4 (chain layout: lane (i, g) holds feature 16 t + 4 g + c of point i)
// expect: [unit][3 tiles][64 lanes] float4 (the D layout of the output layer); bad: count of differing words; first: (launch, unit, tile, lane) of the first few
template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void chain_kernel(const uint2 *fragh, const uint2 *in, const float4 *expect, int n_units, uint32_t launch,
                                                        unsigned long long *bad, uint32_t *first, float4 *out)
{
    extern __shared__ uint2 lds_h[];
    if (MODE != 5) {
        for (uint32_t e = threadIdx.x; e < kNMlp / 4; e += THREADS) lds_h[e] = fragh[e];
        if (MODE == 7)
            for (uint32_t e = kNMlp / 4 + threadIdx.x; e < 150 * 1024 / 8; e += THREADS) lds_h[e] = uint2{0u, 0u};
        __syncthreads();
    }
    const uint2 *img = MODE == 5 ? fragh : lds_h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_tiles = (n_units + kSub - 1) / kSub;
    const uint2 *w0 = img + kWOff0 / 4, *w1 = img + kWOff1 / 4, *w2 = img + kWOff2 / 4, *w3 = img + kWOff3 / 4;
    for (int tile = blockIdx.x * (THREADS / 64) + wave; tile < n_tiles; tile += gridDim.x * (THREADS / 64)) {
        asm volatile("" ::: "memory");
        h4_t b[kSub][4];
        int unit[kSub];
#pragma unroll
        for (int u = 0; u < kSub; ++u) {
            unit[u] = min(tile * kSub + u, n_units - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Frag f;
                f.u = in[((size_t)unit[u] * 2 + t) * 64 + lane];
                b[u][t] = f.h;
            }
        }
        if (MODE == 7) {
            // the gathered words are zeros (the image beyond the weights is cleared by the block before the loop): OR-ed into the inputs
            // they change nothing, but the chain cannot start before they have arrived
            uint32_t z = 0;
            uint32_t a = (uint32_t)lane * 2654435761u + (uint32_t)tile * 40503u;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                a = a * 1664525u + 1013904223u;
                const uint2 w = lds_h[kNMlp / 4 + (a >> 8) % (uint32_t)(150 * 1024 / 8 - kNMlp / 4)];
                z |= w.x | w.y;
            }
#pragma unroll
            for (int u = 0; u < kSub; ++u) {
                Frag f;
                f.h = b[u][0];
                f.u.x |= z;
                b[u][0] = f.h;
            }
        }
        f32x4_t acc[kSub][4];
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            if (l == 0) layer<MODE, 2, 4>(w0, lane, b, acc);
            else layer<MODE, 4, 4>(l == 1 ? w1 : w2, lane, b, acc);
#pragma unroll
            for (int u = 0; u < kSub; ++u)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    b[u][rt] = __builtin_elementwise_max(__builtin_convertvector(acc[u][rt], h4_t), h4_t{(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f});
        }
        layer<MODE, 4, 3>(w3, lane, b, acc);
        float extra = 0.0f;
        if (MODE == 6) {
            // a long VALU tail on the fresh accumulators, the shape of the loss arithmetic (its value is not compared)
#pragma unroll
            for (int u = 0; u < kSub; ++u)
#pragma unroll
                for (int rt = 0; rt < 3; ++rt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = acc[u][rt][c];
                        extra += __expf(v) * __frsqrt_rn(1.0f + v * v) + __shfl(v, (lane + 16) & 63);
                    }
        }
#pragma unroll
        for (int u = 0; u < kSub; ++u)
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) {
                const f32x4_t a = acc[u][rt];
                if (out) {
                    out[((size_t)unit[u] * 3 + rt) * 64 + lane] = float4{a[0], a[1], a[2], a[3]};
                    continue;
                }
                const float4 e = expect[((size_t)unit[u] * 3 + rt) * 64 + lane];
                const int d = (a[0] != e.x) + (a[1] != e.y) + (a[2] != e.z) + (a[3] != e.w);
                if (d) {
                    const unsigned long long k = atomicAdd(bad, (unsigned long long)d);
                    if (k < 64) {
                        first[4 * k + 0] = launch;
                        first[4 * k + 1] = (uint32_t)unit[u];
                        first[4 * k + 2] = (uint32_t)rt;
                        first[4 * k + 3] = (uint32_t)lane;
                    }
                }
            }
        if (MODE == 6 && extra == 123.456f) first[0] = 1;     // keeps the tail alive
    }
}

// ---- what runs BEFORE a launch (EXPERIMENTS 20: the production kernel deviates only in the first tile of a wave, and only when the
// launch follows a different kernel) ----
//   1 lds      every CU's LDS filled with a NaN pattern (what the chain's block finds in LDS before it stages its image)
//   2 vgpr     250 vector registers of every resident wave set to a NaN pattern
//   3 code     a long stretch of other instructions through the instruction cache (a 4096-instruction unrolled VALU chain)
//   4 small    a one-block kernel (the shape of the optimizer step that precedes the production launch: the chip nearly idle)
//   5 idle     nothing on the device for 0.3 ms (host-side wait); 6: that, then the one-block kernel
template <int KIND>
__global__ __launch_bounds__(1024) void before_kernel(uint32_t *sink)
{
    extern __shared__ uint32_t lds_w[];
    if (KIND == 1) {
        for (uint32_t e = threadIdx.x; e < 150 * 1024 / 4; e += blockDim.x) lds_w[e] = 0x7fc0beefu;
        __syncthreads();
        if (lds_w[(threadIdx.x * 97u) % (150 * 1024 / 4)] == 1u) sink[0] = 1u;
    }
    if (KIND == 2) {
        uint32_t v[96];
#pragma unroll
        for (int k = 0; k < 96; ++k) v[k] = 0x7fc00000u + threadIdx.x * 131u + k;
#pragma unroll
        for (int k = 0; k < 96; ++k) asm volatile("" : "+v"(v[k]));
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < 96; ++k) x ^= v[k];
        if (x == 0x12345u) sink[1] = x;
    }
    if (KIND == 3) {
        float a = (float)threadIdx.x, b = 1.0001f;
#pragma unroll
        for (int k = 0; k < 4096; ++k) a = __builtin_fmaf(a, b, (float)k);
        if (a == 123.0f) sink[2] = 1u;
    }
    if (KIND == 4 && threadIdx.x == 0) sink[3] = blockIdx.x;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd()
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}

static double round_h(double v) { return (double)(_Float16)v; }     // one rounding of an exact value (round to nearest even)

static int g_random = 0;      // operands with full mantissas instead of the exact set
static int g_before = 0;       // 0: launches back to back; 1..4: a before_kernel in front of every launch
static uint32_t *g_sink = nullptr;
static void launch_before()
{
    const size_t big = 150 * 1024;
    switch (g_before) {
    case 1: hipLaunchKernelGGL(before_kernel<1>, dim3(256), dim3(1024), big, 0, g_sink); break;
    case 2: hipLaunchKernelGGL(before_kernel<2>, dim3(1024), dim3(256), 0, 0, g_sink); break;
    case 3: hipLaunchKernelGGL(before_kernel<3>, dim3(1024), dim3(256), 0, 0, g_sink); break;
    case 4: hipLaunchKernelGGL(before_kernel<4>, dim3(1), dim3(64), 0, 0, g_sink); break;
    case 5: (void)hipDeviceSynchronize(); usleep(300); break;      // the chip idle for 0.3 ms
    case 6: (void)hipDeviceSynchronize(); usleep(300); hipLaunchKernelGGL(before_kernel<4>, dim3(1), dim3(64), 0, 0, g_sink); break;
    default: break;
    }
}

template <int THREADS, int MODE>
static void run(const char *name, const uint2 *d_frag, const uint2 *d_in, const float4 *d_expect, int n_units, int launches, unsigned long long *d_bad,
                uint32_t *d_first, size_t lds_bytes)
{
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<THREADS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    CHECK(hipMemset(d_bad, 0, sizeof(unsigned long long)));
    const int n_tiles = (n_units + kSub - 1) / kSub;
    const int grid = std::max(1, std::min((n_tiles + THREADS / 64 - 1) / (THREADS / 64), 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    unsigned long long bad_launches = 0, last = 0, bad = 0;
    CHECK(hipEventRecord(e0));
    const int batch = 500;
    for (int l = 0; l < launches; l += batch) {
        // per-launch attribution costs a sync per launch; a batch is read once, the log names the launches
        for (int k = l; k < std::min(launches, l + batch); ++k) {
            launch_before();
            hipLaunchKernelGGL((chain_kernel<THREADS, MODE>), dim3(grid), dim3(THREADS), lds_bytes, 0, d_frag, d_in, d_expect, n_units, (uint32_t)k, d_bad, d_first, (float4 *)nullptr);
        }
        CHECK(hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost));
        if (bad != last) ++bad_launches;
        last = bad;
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    uint32_t first[256];
    CHECK(hipMemcpy(first, d_first, sizeof(first), hipMemcpyDeviceToHost));
    printf("%-13s threads %4d  launches %6d  wrong words %8llu  batches of %d with a wrong word %4llu  (%.1f us per launch)", name, THREADS, launches, bad, batch,
           bad_launches, 1e3 * ms / launches);
    if (bad) {
        // a wave's k-th tile: tile index / (blocks x waves per block)
        size_t in_first = 0, logged = (size_t)std::min<unsigned long long>(bad, 64);
        for (size_t k = 0; k < logged; ++k) in_first += (first[4 * k + 1] / kSub) / (unsigned)(grid * (THREADS / 64)) == 0;
        printf("  %zu of the first %zu wrong words in a wave's FIRST tile; first:", in_first, logged);
        for (unsigned long long k = 0; k < std::min<unsigned long long>(bad, 4); ++k)
            printf(" [launch %u unit %u tile %u lane %u]", first[4 * k], first[4 * k + 1], first[4 * k + 2], first[4 * k + 3]);
    }
    printf("\n");
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 20000;
    const int n = argc > 2 ? atoi(argv[2]) : 524288;
    const int only_mode = argc > 3 ? atoi(argv[3]) : -1;
    g_before = argc > 4 ? atoi(argv[4]) : 0;
    g_random = argc > 5 ? atoi(argv[5]) : 0;
    CHECK(hipMalloc(&g_sink, 64));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(before_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    const int n_units = n / 16;
    // ---- weights, W[layer][row r][column k] ----
    std::vector<double> W(kNMlp);
    const int n_i[4] = {kEnc, kH, kH, kH}, n_o[4] = {kH, kH, kH, kOutPad}, off[4] = {kWOff0, kWOff1, kWOff2, kWOff3};
    for (int l = 0; l < 4; ++l)
        for (int e = 0; e < n_i[l] * n_o[l]; ++e) {
            const uint32_t r = rnd() % 8;
            // slightly more positive than negative weights keep a good share of the hidden units alive behind the ReLU
            const double s = r < 4 ? 1.0 : (r < 7 ? -1.0 : 0.0);
            W[off[l] + e] = s / (l == 0 ? 8.0 : 16.0);
            // "random" data: full-mantissa weights of the size a trained network has (the operands toggle many more bits per
            // instruction than the exact set's do); the expectation is then the calibration launch's result, not the host's
            if (g_random) W[off[l] + e] = (double)(_Float16)(((double)(rnd() & 0xffff) / 32768.0 - 1.0) * (l == 0 ? 0.4 : 0.25));
        }
    std::vector<uint2> frag(kNMlp / 4);
    for (int l = 0; l < 4; ++l) {
        const int KT = n_i[l] / 16;
        for (int f = 0; f < n_i[l] * n_o[l] / 4; ++f) {
            const int ln = f & 63, t = f >> 6, rt = t / KT, kt = t % KT, i = ln & 15, g = ln >> 4;
            Frag v;
            for (int c = 0; c < 4; ++c) v.h[c] = (_Float16)W[off[l] + (16 * rt + i) * n_i[l] + 16 * kt + 4 * g + c];
            frag[off[l] / 4 + f] = v.u;
        }
    }
    // ---- inputs and the expected outputs ----
    std::vector<uint2> in((size_t)n_units * 2 * 64);
    std::vector<float4> expect((size_t)n_units * 3 * 64);
    std::vector<double> x(kH), y(kH);
    size_t nonzero = 0;
    for (int u = 0; u < n_units; ++u)
        for (int i = 0; i < 16; ++i) {
            for (int k = 0; k < kEnc; ++k) x[k] = g_random ? (double)(_Float16)((double)(rnd() & 0xffff) / 65536.0) : ((rnd() & 1) ? 1.0 : 0.0);
            for (int t = 0; t < 2; ++t)
                for (int g = 0; g < 4; ++g) {
                    Frag v;
                    for (int c = 0; c < 4; ++c) v.h[c] = (_Float16)x[16 * t + 4 * g + c];
                    in[((size_t)u * 2 + t) * 64 + 16 * g + i] = v.u;
                }
            for (int l = 0; l < 4; ++l) {
                for (int r = 0; r < n_o[l]; ++r) {
                    double s = 0.0;
                    for (int k = 0; k < n_i[l]; ++k) s += W[off[l] + r * n_i[l] + k] * x[k];
                    y[r] = l < 3 ? std::max(round_h(s), 0.0) : s;
                }
                for (int r = 0; r < n_o[l]; ++r) x[r] = y[r];
            }
            for (int rt = 0; rt < 3; ++rt)
                for (int g = 0; g < 4; ++g) {
                    float4 e = {(float)x[16 * rt + 4 * g + 0], (float)x[16 * rt + 4 * g + 1], (float)x[16 * rt + 4 * g + 2], (float)x[16 * rt + 4 * g + 3]};
                    nonzero += (e.x != 0) + (e.y != 0) + (e.z != 0) + (e.w != 0);
                    expect[((size_t)u * 3 + rt) * 64 + 16 * g + i] = e;
                }
        }
    printf("mfma_chain: %d points (%d units), %d launches per configuration, %.1f %% of the expected outputs non-zero; before every launch: %s\n", n, n_units, launches,
           100.0 * nonzero / ((double)n * 48), g_before == 0 ? "nothing (back to back)" : g_before == 1 ? "LDS filled with NaN patterns" : g_before == 2 ? "vector registers set to NaN patterns" :
           g_before == 3 ? "4096 other instructions through the instruction cache" : g_before == 4 ? "a one-block kernel" : g_before == 5 ? "the chip idle for 0.3 ms" : "0.3 ms idle, then a one-block kernel");
    uint2 *d_frag, *d_in;
    float4 *d_expect;
    unsigned long long *d_bad;
    uint32_t *d_first;
    CHECK(hipMalloc(&d_frag, frag.size() * sizeof(uint2)));
    CHECK(hipMalloc(&d_in, in.size() * sizeof(uint2)));
    CHECK(hipMalloc(&d_expect, expect.size() * sizeof(float4)));
    CHECK(hipMalloc(&d_bad, sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_first, 256 * sizeof(uint32_t)));
    CHECK(hipMemcpy(d_frag, frag.data(), frag.size() * sizeof(uint2), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_in, in.data(), in.size() * sizeof(uint2), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_expect, expect.data(), expect.size() * sizeof(float4), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_first, 0, 256 * sizeof(uint32_t)));
    const size_t lds = 150 * 1024;      // the production kernel's image (weights + grid): one block per CU
    {
        // calibration: the padded chain at one wave per SIMD against the host's exact numbers
        float4 *d_out;
        CHECK(hipMalloc(&d_out, expect.size() * sizeof(float4)));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<256, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((chain_kernel<256, 1>), dim3(256), dim3(256), lds, 0, d_frag, d_in, d_expect, n_units, 0u, d_bad, d_first, d_out);
        std::vector<float4> got(expect.size());
        CHECK(hipMemcpy(got.data(), d_out, got.size() * sizeof(float4), hipMemcpyDeviceToHost));
        const size_t diff = (size_t)(memcmp(got.data(), expect.data(), got.size() * sizeof(float4)) != 0);
        size_t words = 0;
        if (diff)
            for (size_t k = 0; k < got.size() * 4; ++k) words += reinterpret_cast<const float *>(got.data())[k] != reinterpret_cast<const float *>(expect.data())[k];
        printf("calibration (256 threads, settle): %zu of %zu words differ from the host's exact result%s\n", words, got.size() * 4,
               words ? " -- the device's own result is the expectation from here on" : "");
        if (words) {
            // (full-mantissa operands: the matrix instruction's order of additions decides the last bits) the calibration is repeated until
            // two launches in a row agree, and that result is what every launch below must reproduce
            std::vector<float4> again(expect.size());
            for (int t = 0; t < 8; ++t) {
                hipLaunchKernelGGL((chain_kernel<256, 1>), dim3(256), dim3(256), lds, 0, d_frag, d_in, d_expect, n_units, 0u, d_bad, d_first, d_out);
                CHECK(hipMemcpy(again.data(), d_out, again.size() * sizeof(float4), hipMemcpyDeviceToHost));
                const bool same = memcmp(again.data(), got.data(), got.size() * sizeof(float4)) == 0;
                got.swap(again);
                if (same) break;
                printf("calibration: two launches in a row differ, once more\n");
            }
            CHECK(hipMemcpy(d_expect, got.data(), got.size() * sizeof(float4), hipMemcpyHostToDevice));
        }
        CHECK(hipFree(d_out));
    }
#define RUN3(MODE, NAME)                                                                                      \
    if (only_mode < 0 || only_mode == MODE) {                                                                 \
        run<1024, MODE>(NAME, d_frag, d_in, d_expect, n_units, launches, d_bad, d_first, lds);                \
        run<768, MODE>(NAME, d_frag, d_in, d_expect, n_units, launches, d_bad, d_first, lds);                 \
        run<512, MODE>(NAME, d_frag, d_in, d_expect, n_units, launches, d_bad, d_first, lds);                 \
        run<256, MODE>(NAME, d_frag, d_in, d_expect, n_units, launches, d_bad, d_first, lds);                 \
    }
    RUN3(0, "plain")
    RUN3(7, "gather")
    RUN3(6, "tail")
    RUN3(1, "settle")
    RUN3(2, "agpr")
    RUN3(3, "k16")
    RUN3(4, "schedbarrier")
    RUN3(5, "nolds")
    return 0;
}
