// wost_net.hip -- the guiding network of the guided integrator on the device (SURVEY.md 8a rows
// a22/a23 and the optimizer of a27), behind the C-ABI: DenseGrid encoding -> bias-free ReLU MLP
// -> raw mixture parameters; backward pass; Adam nested in a debiased EMA.  gfx950 only.
//
// Replaces (does not port) tiny-cuda-nn as used through the reference adapter
// util/network.h:21-196 with the configuration of data/ladybug/n.json:49-81.  The reference runs
// the MLP in half precision on tensor cores; here everything is fp32 with k-ordered fmaf chains,
// so the result is reproducible and equal to the CPU restatement bit for bit:
//   * the reference's network shape runs on the matrix cores (v_mfma_f32_16x16x4_f32, exact
//     fp32): net_forward_mfma_kernel, net_backward_wgrad_kernel (backward pass + weight gradients
//     of a 1024-point chunk in one block; apart, for WOST_NET_FUSED=0: net_backward_mfma_kernel,
//     weight_grad_mfma_kernel);
//   * other shapes (and WOST_NET_SCALAR=1) use the scalar kernels: one thread per point,
//     activations in a per-lane LDS column, wave-uniform weights through scalar loads;
//   * gradients are sums over the training points and are accumulated in 64-bit fixed point
//     with integer atomics, hence independent of the order in which blocks arrive.
// DESIGN.md section 4.7 has the numbers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/wost.h"
#include "wost_internal.h"
#include "wost_math.h"
#include "wost_net_device.h"
#include "wost_vmm_device.h"

namespace wost {

constexpr int kNetBlock = 64;

// Gradients are SUMS over the training points, and float atomics would make them depend on the
// order in which blocks and lanes arrive.  Every partial sum is therefore converted to 64-bit
// fixed point (2^-36 resolution, +-1.3e8 range) and accumulated with integer atomics: integer
// addition is associative, so the gradient -- and with it the whole training run -- is
// reproducible bit for bit, on this device and against the CPU restatement.
typedef long long fx_t;
constexpr double kFxScale = 68719476736.0;   // 2^36
// round-to-nearest-even of v * 2^36.  The product is exact in fp32 (a power of two), so below 2^31
// the fp32 rounding instruction gives the same integer as the fp64 expression, which stays as
// the path for larger values (and NaN / infinity).
__device__ __forceinline__ fx_t to_fx(float v)
{
    const float f = v * 68719476736.0f;
    if (fabsf(f) < 2147483648.0f) return (fx_t)(int)__builtin_rintf(f);
    return __double2ll_rn((double)v * kFxScale);
}
__device__ __forceinline__ void fx_add(fx_t *p, fx_t v)
{
    atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v);
}

static NetLayout make_layout(const wost_net_config &c, int dims)
{
    NetLayout l{};
    l.dims = dims;
    const float log2s = std::log2(c.per_level_scale);
    uint32_t off = 0;
    for (int i = 0; i < c.n_levels; ++i) {
        l.scale[i] = std::exp2((float)i * log2s) * (float)c.base_resolution - 1.0f;
        l.res[i] = (int)std::ceil(l.scale[i]) + 1;
        uint32_t n = (uint32_t)l.res[i] * (uint32_t)l.res[i];
        if (dims == 3) n *= (uint32_t)l.res[i];
        n = (n + 7u) / 8u * 8u;
        l.level_off[i] = off;
        off += n;
    }
    l.level_off[c.n_levels] = off;
    l.n_levels = c.n_levels;
    l.n_features = c.n_features_per_level;
    l.enc = c.n_levels * c.n_features_per_level;
    l.n_neurons = c.n_neurons;
    l.n_hidden = c.n_hidden_layers;
    l.n_out = c.n_output;
    l.n_out_padded = (c.n_output + 15) / 16 * 16;
    l.n_grid = off * (uint32_t)c.n_features_per_level;
    l.w_off[0] = 0;
    l.w_off[1] = (uint32_t)(l.n_neurons * l.enc);
    for (int i = 2; i <= l.n_hidden; ++i) l.w_off[i] = l.w_off[i - 1] + (uint32_t)(l.n_neurons * l.n_neurons);
    l.n_mlp = l.w_off[l.n_hidden] + (uint32_t)(l.n_out_padded * l.n_neurons);
    return l;
}

// ---- device code --------------------------------------------------------------------------
// per-lane activation column in LDS: element k of this thread at col[k * kNetBlock]
__device__ __forceinline__ float &act(float *col, int k) { return col[k * kNetBlock]; }

// DenseGrid encoding of one point into col[0..enc) (tiny-cuda-nn grid.h: grid_scale,
// grid_resolution, dense grid_index with wrap, linear interpolation)
__device__ __forceinline__ void encode_point(const NetLayout &L, const float *grid, float x, float y, float z, float *col)
{
    const int n_corners = 1 << L.dims;
    for (int lv = 0; lv < L.n_levels; ++lv) {
        const float s = L.scale[lv];
        const uint32_t res = (uint32_t)L.res[lv];
        const uint32_t n_level = L.level_off[lv + 1] - L.level_off[lv];
        float px = __builtin_fmaf(s, x, 0.5f), py = __builtin_fmaf(s, y, 0.5f), pz = __builtin_fmaf(s, z, 0.5f);
        const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
        px -= fx;
        py -= fy;
        pz -= fz;
        const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy, iz = (uint32_t)(int)fz;
        for (int q = 0; q < L.n_features; ++q) act(col, lv * L.n_features + q) = 0.0f;
        for (int k = 0; k < n_corners; ++k) {
            // corner k: bit 0 = +x, bit 1 = +y, bit 2 = +z (three inputs only); weight = product of the axis weights in that order
            const uint32_t cx = ix + (k & 1), cy = iy + ((k >> 1) & 1);
            float w = ((k & 1) ? px : 1.0f - px) * ((k & 2) ? py : 1.0f - py);
            uint32_t idx;
            if (L.dims == 3) {
                // (64-bit like the CPU restatement, so that inputs outside the unit cube -- walkers outside the scene box -- wrap alike)
                w = w * ((k & 4) ? pz : 1.0f - pz);
                const uint32_t cz = iz + ((k >> 2) & 1);
                idx = (uint32_t)(((unsigned long long)cx + (unsigned long long)cy * res + (unsigned long long)cz * res * res) % n_level);
            } else {
                idx = (cx + cy * res) % n_level;
            }
            const float *g = grid + (size_t)(L.level_off[lv] + idx) * L.n_features;
            for (int q = 0; q < L.n_features; ++q) act(col, lv * L.n_features + q) += w * g[q];
        }
    }
}

// out[r] = sum_k W[r][k] * in[k] (k ascending, fma chain), optional ReLU.  The weights are
// wave-uniform and read from the TRANSPOSED copy Wt[k][r], so the 8 weights of one k are one
// s_load_dwordx8.
__device__ __forceinline__ void dense_layer(const float *Wt, int n_out, int n_in, const float *in_col, float *out_col, bool relu)
{
    for (int r0 = 0; r0 < n_out; r0 += 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < n_in; ++k) {
            const float a = in_col[k * kNetBlock];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(Wt[(size_t)k * n_out + r0 + j], a, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) out_col[(r0 + j) * kNetBlock] = relu ? fmaxf(acc[j], 0.0f) : acc[j];
    }
}

// forward; out: n x n_out (unpadded).  acts (optional): n x (enc + n_hidden * n_neurons), the
// activations the backward pass needs.
__global__ __launch_bounds__(kNetBlock) void net_forward_kernel(NetLayout L, const float *params, const float *wt,
                                                                const float *xy, int n, const uint32_t *n_dev, float *out,
                                                                float *acts, size_t out_ldp, size_t out_ldf)
{
    if (n_dev) n = (int)*n_dev;                              // queue size decided on the device
    if ((int)(blockIdx.x * kNetBlock) >= n) return;
    extern __shared__ float lds[];
    float *col_a = lds + threadIdx.x;                       // ping
    float *col_b = lds + 64 * kNetBlock + threadIdx.x;      // pong (widths <= 64)
    const int p = blockIdx.x * kNetBlock + threadIdx.x;
    const bool valid = p < n;
    const size_t xo = (size_t)L.dims * p;        // inputs: dims floats per point
    const float x = valid ? xy[xo] : 0.5f, y = valid ? xy[xo + 1] : 0.5f, z = (valid && L.dims == 3) ? xy[xo + 2] : 0.5f;
    encode_point(L, params + L.n_mlp, x, y, z, col_a);
    const int stride = L.enc + L.n_hidden * L.n_neurons;
    if (acts && valid)
        for (int k = 0; k < L.enc; ++k) acts[(size_t)p * stride + k] = act(col_a, k);
    float *in = col_a, *o = col_b;
    int n_in = L.enc;
    for (int layer = 0; layer < L.n_hidden; ++layer) {
        dense_layer(wt + L.w_off[layer], L.n_neurons, n_in, in, o, true);
        if (acts && valid)
            for (int k = 0; k < L.n_neurons; ++k) acts[(size_t)p * stride + L.enc + layer * L.n_neurons + k] = act(o, k);
        float *t = in; in = o; o = t;
        n_in = L.n_neurons;
    }
    dense_layer(wt + L.w_off[L.n_hidden], L.n_out_padded, n_in, in, o, false);
    if (valid)
        for (int k = 0; k < L.n_out; ++k) out[(size_t)p * out_ldp + k * out_ldf] = act(o, k);
}

// ---- MFMA forward -------------------------------------------------------------------------------
// The MLP is the one dense contraction of the whole path.  v_mfma_f32_16x16x4_f32
// is bit-for-bit a k-ordered fmaf chain, so this kernel returns exactly what net_forward_kernel
// (and the CPU restatement) returns, at MFMA operand bandwidth instead of one LDS read per 8 FMAs.
//
// One wave owns 2 x 16 points.  For out[r][p] = sum_k W[r][k] in[k][p] the A operand is a
// 16-row tile of W, the B operand the activations: lane (i = l&15, g = l>>4) supplies
// in[4s + g][point i] at k-step s and receives D[4g + c][point i] in accumulator register c.
// The rows of every A tile are stored PERMUTED (physical row 4a+b = logical row 4b+a), so that
// register c of row tile rt holds logical feature 16rt + 4c + g: exactly the B operand of k-step
// s = 4rt + c of the next layer.  Layer outputs never leave the registers and the k order stays
// ascending.  Weight fragments (13 312 floats) sit in LDS in lane order: one conflict-free
// ds_read_b32 per MFMA pair.

constexpr int kMfmaSub = 2;   // 16-point subtiles per wave iteration
// staging rows of the encoded features are padded by two floats: the B-operand reads of a k-step
// (16 points x 4 consecutive features) then fall into 32 different LDS banks per half wave
constexpr int kStagePad = 2;
constexpr int kFwdThreads = 1024, kFwdTrainThreads = 256, kBwdThreads = 1024;   // block sizes of the MFMA kernels

// Training layout of the saved activations and deltas (MFMA path).  The weight-gradient kernel
// contracts over points: chain w of a 1024-point chunk takes the 4-point groups w, w+4, w+8, ..
// (net.c of the CPU restatement follows the same order).  A 16-column unit of the forward and
// backward kernels therefore holds the four groups w, w+4, w+8, w+12 of one 64-point span, column
// 4g+m being point 16m + 4w + g of the span, and the unit's values are stored feature-major,
// [unit][feature][16 columns]: the forward / backward kernels write 256 contiguous bytes per
// store, and lane (feature i, k-slot g) of the weight-gradient kernel reads its operands of four
// consecutive MFMAs (columns 4g .. 4g+3 = slot g of the groups m = 0..3) as ONE 16-byte load.
__device__ __forceinline__ int train_point(int unit, int column)
{
    return 64 * (unit >> 2) + 16 * (column & 3) + 4 * (unit & 3) + (column >> 2);
}

// frag[w_off[layer] + (rt * S + s) * 64 + l] = W[16rt + 4((l&15)&3) + ((l&15)>>2)][4s + (l>>4)]
__global__ void fragment_mlp_kernel(NetLayout L, const float *src, float *dst)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.n_mlp) return;
    int layer = 0;
    while (layer < L.n_hidden && e >= L.w_off[layer + 1]) ++layer;
    const int n_i = layer == 0 ? L.enc : L.n_neurons;
    const int S = n_i / 4;
    const uint32_t f = e - L.w_off[layer];
    const uint32_t l = f & 63u, t = f >> 6;
    const uint32_t rt = t / S, s = t % S;
    const uint32_t i = l & 15u, g = l >> 4;
    const uint32_t row = 16 * rt + 4 * (i & 3u) + (i >> 2), k = 4 * s + g;
    dst[e] = src[L.w_off[layer] + row * n_i + k];
}

template <int S, int RT>
__device__ __forceinline__ void mfma_layer(const float *wf, int lane, const float (&b)[kMfmaSub][16], f32x4_t (&acc)[kMfmaSub][4])
{
#pragma unroll
    for (int u = 0; u < kMfmaSub; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[u][rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float a = wf[(rt * S + s) * 64 + lane];
#pragma unroll
            for (int u = 0; u < kMfmaSub; ++u) acc[u][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[u][s], acc[u][rt], 0, 0, 0);
        }
    }
}

// ENC = encoded width, H = neurons, NH = hidden layers (the first takes ENC inputs), NOP = padded outputs,
// NF = features per grid level known at compile time (0: read from the layout)
template <int ENC, int H, int NH, int NOP, bool SAVE, int NF, int NT>
__global__ __launch_bounds__(NT) void net_forward_mfma_kernel(NetLayout L, const float *params, const float *frag,
                                                                  const float *xy, int n, const uint32_t *n_dev, float *out,
                                                                  float *acts, unsigned long long *relu_mask, size_t out_ldp,
                                                                  size_t out_ldf)
{
    static_assert(ENC % 4 == 0 && ENC <= 64 && H % 16 == 0 && H <= 64 && NOP % 16 == 0 && NOP <= 64, "shape");
    extern __shared__ float lds[];
    __shared__ float s_scale[kNetMaxLevels];
    __shared__ uint32_t s_res[kNetMaxLevels], s_off[kNetMaxLevels + 1];
    float *wfrag = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int SS = ENC + kStagePad;
    float *stage = lds + L.n_mlp + wave * (kMfmaSub * 16 * SS);
    for (uint32_t e = threadIdx.x; e < L.n_mlp; e += NT) wfrag[e] = frag[e];
    if (threadIdx.x <= (unsigned)L.n_levels) {
        s_off[threadIdx.x] = L.level_off[threadIdx.x];
        if (threadIdx.x < (unsigned)L.n_levels) {
            s_scale[threadIdx.x] = L.scale[threadIdx.x];
            s_res[threadIdx.x] = (uint32_t)L.res[threadIdx.x];
        }
    }
    __syncthreads();
    if (n_dev) n = (int)*n_dev;
    const int i = lane & 15, g = lane >> 4;
    // 16-column units: consecutive points for inference, the training layout when activations are kept
    const int n_tiles = SAVE ? (n + 63) / 64 * 4 / kMfmaSub : (n + 16 * kMfmaSub - 1) / (16 * kMfmaSub);
    const float *grid = params + L.n_mlp;
    const int astride = ENC + NH * H;
    for (int tile = blockIdx.x * (NT / 64) + wave; tile < n_tiles; tile += gridDim.x * (NT / 64)) {
        int pt[kMfmaSub];
        bool valid[kMfmaSub];
        float *arow[kMfmaSub];      // SAVE: this lane's column of the unit's activation block
        // ---- encoding: lane (i, g) interpolates the levels lv = g, g + 4, ... of its point
#pragma unroll
        for (int u = 0; u < kMfmaSub; ++u) {
            const int unit = tile * kMfmaSub + u;
            pt[u] = SAVE ? train_point(unit, i) : unit * 16 + i;
            valid[u] = pt[u] < n;
            arow[u] = SAVE ? acts + (size_t)unit * 16 * astride + i : nullptr;
            const size_t dims = (size_t)L.dims;      // two inputs, or three (GuidedIntegrator<3>: the NF == 4 branch only)
            const float x = valid[u] ? xy[dims * (size_t)pt[u]] : 0.5f, y = valid[u] ? xy[dims * (size_t)pt[u] + 1] : 0.5f;
            const float z = (valid[u] && dims == 3) ? xy[3 * (size_t)pt[u] + 2] : 0.5f;
            if constexpr (NF == 4) {
                // the reference's grid (4 features per level): one 16-byte gather per corner, all
                // eight of a point's two levels in flight together; the wrap of the dense index
                // is a compare-subtract except for points far outside the unit square
                static_assert(ENC == 32, "8 levels x 4 features");
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int lv = g + 4 * h;
                    const float4 f = dims == 3 ? f32_encode_level3(grid, s_scale[lv], s_res[lv], s_off[lv], s_off[lv + 1] - s_off[lv], x, y, z)
                                               : f32_encode_level(grid, s_scale[lv], s_res[lv], s_off[lv], s_off[lv + 1] - s_off[lv], x, y);
                    float2 *st = reinterpret_cast<float2 *>(stage + (u * 16 + i) * SS + lv * 4);
                    st[0] = float2{f.x, f.y};
                    st[1] = float2{f.z, f.w};
                    if (SAVE && valid[u]) {
                        arow[u][(lv * 4 + 0) * 16] = f.x; arow[u][(lv * 4 + 1) * 16] = f.y;
                        arow[u][(lv * 4 + 2) * 16] = f.z; arow[u][(lv * 4 + 3) * 16] = f.w;
                    }
                }
            } else
            for (int lv = g; lv < L.n_levels; lv += 4) {
                const float sc = s_scale[lv];
                const uint32_t res = s_res[lv], lo = s_off[lv];
                const uint32_t n_level = s_off[lv + 1] - lo;
                float px = __builtin_fmaf(sc, x, 0.5f), py = __builtin_fmaf(sc, y, 0.5f);
                const float fx = floorf(px), fy = floorf(py);
                px -= fx;
                py -= fy;
                const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
                float f[8];
                for (int q = 0; q < L.n_features; ++q) f[q] = 0.0f;
                for (int k = 0; k < 4; ++k) {
                    const uint32_t cx = ix + (k & 1), cy = iy + (k >> 1);
                    const float w = ((k & 1) ? px : 1.0f - px) * ((k >> 1) ? py : 1.0f - py);
                    const uint32_t idx = (cx + cy * res) % n_level;
                    const float *gp = grid + (size_t)(lo + idx) * L.n_features;
                    for (int q = 0; q < L.n_features; ++q) f[q] += w * gp[q];
                }
                for (int q = 0; q < L.n_features; ++q) {
                    stage[(u * 16 + i) * SS + lv * L.n_features + q] = f[q];
                    if (SAVE && valid[u]) arow[u][(lv * L.n_features + q) * 16] = f[q];
                }
            }
        }
        __threadfence_block();      // the wave reads back what its own lanes wrote
        float b[kMfmaSub][16];
#pragma unroll
        for (int u = 0; u < kMfmaSub; ++u)
#pragma unroll
            for (int s = 0; s < ENC / 4; ++s) b[u][s] = stage[(u * 16 + i) * SS + 4 * s + g];
        __threadfence_block();      // before the next iteration overwrites the staging area
        f32x4_t acc[kMfmaSub][4];
        // which hidden activations are positive, bit (layer * 16 + 4 rt + c) of this lane's word: the
        // backward kernel holds the same features in the same lanes and needs nothing else of them
        static_assert(NH * (H / 16) * 4 <= 64, "ReLU mask word");
        unsigned long long mask[kMfmaSub];
#pragma unroll
        for (int u = 0; u < kMfmaSub; ++u) mask[u] = 0ull;
        // ---- hidden layers: ReLU, the accumulators become the next B operands
#pragma unroll
        for (int layer = 0; layer < NH; ++layer) {
            if (layer == 0) mfma_layer<ENC / 4, H / 16>(wfrag + L.w_off[0], lane, b, acc);
            else mfma_layer<H / 4, H / 16>(wfrag + L.w_off[layer], lane, b, acc);
#pragma unroll
            for (int u = 0; u < kMfmaSub; ++u)
#pragma unroll
                for (int rt = 0; rt < H / 16; ++rt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = fmaxf(acc[u][rt][c], 0.0f);
                        b[u][4 * rt + c] = v;
                        if (SAVE) mask[u] |= (unsigned long long)(v > 0.0f) << (layer * 16 + 4 * rt + c);
                        if (SAVE && valid[u]) arow[u][(ENC + layer * H + 16 * rt + 4 * c + g) * 16] = v;
                    }
        }
        if (SAVE) {
#pragma unroll
            for (int u = 0; u < kMfmaSub; ++u)
                if (valid[u]) relu_mask[(size_t)pt[u] * 4 + g] = mask[u];
        }
        // ---- output layer
        mfma_layer<H / 4, NOP / 16>(wfrag + L.w_off[NH], lane, b, acc);
#pragma unroll
        for (int u = 0; u < kMfmaSub; ++u)
#pragma unroll
            for (int rt = 0; rt < NOP / 16; ++rt)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int o = 16 * rt + 4 * c + g;
                    if (valid[u] && o < L.n_out) out[(size_t)pt[u] * out_ldp + o * out_ldf] = acc[u][rt][c];
                }
    }
}

// ---- half-precision inference (the reference's own arithmetic) ---------------------------------------
// tiny-cuda-nn runs the guiding network in half precision (FullyFusedMLP on tensor cores, the grid
// stored in half: util/network.h:21-196, data/ladybug/n.json:49-81).  This is that mode for the
// inference launches of the guided walk: grid values rounded to f16, bilinear interpolation in
// fp32, activations and weights in f16, v_mfma_f32_16x16x16_f16 with fp32 accumulation (16x the
// matrix rate of the fp32 MFMA), ReLU, f16 again between layers, fp32 outputs.  NOT bit-comparable
// with the fp32 path or the oracle -- gated by the tolerance tests of tests/test_guided_network.py;
// fp32 stays the default.  Layout: lane (i, g) of a wave supplies B[k = 16 kt + 4g + c][point i]
// and receives D[r = 16 rt + 4g + c][point i], so the accumulators of row tile rt ARE the B
// operand of k-tile kt = rt of the next layer, and lane (i, g) encodes exactly the grid levels
// g and g + 4 of its point: nothing is exchanged between lanes.
constexpr int kHalfSub = 2;       // 16-point groups per wave iteration
constexpr int kHalfThreads = 256;      // training kernel (wost_net_half.h): one wave per SIMD, the accumulators take the registers
#ifndef WOST_HALF_FWD_THREADS
#define WOST_HALF_FWD_THREADS 512
#endif
constexpr int kHalfFwdThreads = WOST_HALF_FWD_THREADS;  // forward kernel: eight waves share the LDS image, two per SIMD (four per SIMD: EXPERIMENTS 17)

// fragh[(w_off[layer] / 4) + (rt * KT + kt) * 64 + lane] = W[16 rt + i][16 kt + 4g .. 4g + 3], KT = n_i / 16
// followed by the grid, entry by entry (4 features, 8 bytes): the image net_forward_h_kernel keeps in LDS
__global__ void fragment_mlp_h_kernel(NetLayout L, const float *src, uint2 *dst)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.n_mlp / 4) {
        const uint32_t entry = e - L.n_mlp / 4;
        if (entry >= L.level_off[L.n_levels]) return;
        const float4 c = *reinterpret_cast<const float4 *>(src + L.n_mlp + (size_t)entry * 4);
        union { h4_t h; uint2 u; } v;
        v.h = h4_t{(_Float16)c.x, (_Float16)c.y, (_Float16)c.z, (_Float16)c.w};
        dst[e] = v.u;
        return;
    }
    int layer = 0;
    while (layer < L.n_hidden && 4 * e >= L.w_off[layer + 1]) ++layer;
    const int n_i = layer == 0 ? L.enc : L.n_neurons;
    const int KT = n_i / 16;
    const uint32_t f = e - L.w_off[layer] / 4, l = f & 63u, t = f >> 6;
    const uint32_t rt = t / KT, kt = t % KT, i = l & 15u, g = l >> 4;
    const float *w = src + L.w_off[layer] + (size_t)(16 * rt + i) * n_i + 16 * kt + 4 * g;
    union { h4_t h; uint2 u; } v;
    v.h = h4_t{(_Float16)w[0], (_Float16)w[1], (_Float16)w[2], (_Float16)w[3]};
    dst[e] = v.u;
}

template <int KT, int RT>
__device__ __forceinline__ void mfma_layer_h(const uint2 *wf, int lane, const h4_t (&b)[kHalfSub][4], f32x4_t (&acc)[kHalfSub][4])
{
#pragma unroll
    for (int u = 0; u < kHalfSub; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[u][rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    static_assert(KT % 2 == 0, "pairs of 16-deep operands (mfma_k32)");
#pragma unroll
    for (int kt = 0; kt < KT; kt += 2)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            union { uint2 u; h4_t h; } a0, a1;
            a0.u = wf[(rt * KT + kt) * 64 + lane];
            a1.u = wf[(rt * KT + kt + 1) * 64 + lane];
#pragma unroll
            for (int u = 0; u < kHalfSub; ++u) acc[u][rt] = mfma_k32(a0.h, a1.h, b[u][kt], b[u][kt + 1], acc[u][rt]);
        }
}

// the reference's network shape only: 8 levels x 4 features -> 64 -> 64 -> 64 -> 48 (33 used with two inputs, 41 with three).
// DIMS = 2: weights AND grid in LDS; DIMS = 3: the dense 3-D grid is far larger than LDS (a million entries) -- only the weight
// fragments are staged, the eight corner gathers of a level go to L2.

template <int DIMS>
__global__ __launch_bounds__(kHalfFwdThreads) void net_forward_h_kernel(NetLayout L, const float *params, const uint2 *fragh, const float *xy, int n,
                                                                       const uint32_t *n_dev, float *out, size_t out_ldp, size_t out_ldf, uint2 *enc_out)
{
    extern __shared__ uint2 lds_h[];
    __shared__ float s_scale[kNetMaxLevels];
    __shared__ uint32_t s_res[kNetMaxLevels], s_off[kNetMaxLevels + 1];
    // weights and the whole grid (15 384 entries x 8 bytes in half precision): 150 KB of the CU's 160 KB, one block per CU;
    // every gather of the encoding is an LDS read
    const uint32_t n_image = L.n_mlp / 4 + (DIMS == 2 ? L.level_off[L.n_levels] : 0u);
    for (uint32_t e = threadIdx.x; e < n_image; e += kHalfFwdThreads) lds_h[e] = fragh[e];
    if (threadIdx.x <= (unsigned)L.n_levels) {
        s_off[threadIdx.x] = L.level_off[threadIdx.x];
        if (threadIdx.x < (unsigned)L.n_levels) {
            s_scale[threadIdx.x] = L.scale[threadIdx.x];
            s_res[threadIdx.x] = (uint32_t)L.res[threadIdx.x];
        }
    }
    __syncthreads();
    if (n_dev) n = (int)*n_dev;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int n_tiles = (n + 16 * kHalfSub - 1) / (16 * kHalfSub);
    const uint2 *grid = (DIMS == 2 ? lds_h : fragh) + L.n_mlp / 4;
    const uint2 *w0 = lds_h + L.w_off[0] / 4, *w1 = lds_h + L.w_off[1] / 4, *w2 = lds_h + L.w_off[2] / 4, *w3 = lds_h + L.w_off[3] / 4;
    // A wave's FIRST tile is computed twice: first in its turn, then once more behind the wave's last tile, the second result
    // overwriting the first.  Measured (EXPERIMENTS 20): when this kernel follows a DIFFERENT kernel on the device, the first tile a
    // wave computes -- and only that one: 37 of 37 deviating units in 3 840 full-size launches -- now and then comes out with a whole
    // 16-point unit a percent off; the same launch behind a launch of itself never does (0 of 3 840).  What the first pass of a
    // wave through this code finds different (instruction cache, leftover state) is not known; its last pass finds what all the
    // others found, and the stores are idempotent.  One tile in nine at full size.
    const int tile_first = blockIdx.x * (kHalfFwdThreads / 64) + wave, tile_step = gridDim.x * (kHalfFwdThreads / 64);
#ifdef WOST_H_NO_REDO      // (developer builds: the kernel of rounds 1-4)
    const bool redo_first = false;
#else
    const bool redo_first = tile_first < n_tiles;
#endif
    for (int tile_it = tile_first; tile_it < n_tiles || (redo_first && tile_it < n_tiles + tile_step); tile_it += tile_step) {
        const int tile = tile_it < n_tiles ? tile_it : tile_first;
        asm volatile("" ::: "memory");      // the weight fragments are re-read from LDS per tile, not parked in registers
        int pt[kHalfSub];
        bool valid[kHalfSub];
        h4_t b[kHalfSub][4];
#pragma unroll
        for (int u = 0; u < kHalfSub; ++u) {
            pt[u] = (tile * kHalfSub + u) * 16 + i;
            valid[u] = pt[u] < n;
            const float x = valid[u] ? xy[DIMS * (size_t)pt[u]] : 0.5f, y = valid[u] ? xy[DIMS * (size_t)pt[u] + 1] : 0.5f;
            const float z = (DIMS == 3 && valid[u]) ? xy[DIMS * (size_t)pt[u] + 2] : 0.5f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int lv = g + 4 * h;
                b[u][h] = DIMS == 3 ? half_encode_level3(grid, s_scale[lv], s_res[lv], s_off[lv], s_off[lv + 1] - s_off[lv], x, y, z)
                                    : half_encode_level(grid, s_scale[lv], s_res[lv], s_off[lv], s_off[lv + 1] - s_off[lv], x, y);
                if (enc_out) {
                    // training: the encoding goes to the fused backward kernel as it stands (wost_net_half.h)
                    union { h4_t h; uint2 u; } e;
                    e.h = b[u][h];
                    enc_out[((size_t)(tile * kHalfSub + u) * 2 + h) * 64 + lane] = e.u;
                }
            }
        }
        f32x4_t acc[kHalfSub][4];
        // ---- hidden layers: ReLU, the accumulators of row tile rt become the B operand of k-tile rt
#pragma unroll
        for (int layer = 0; layer < 3; ++layer) {
            if (layer == 0) mfma_layer_h<2, 4>(w0, lane, b, acc);
            else mfma_layer_h<4, 4>(layer == 1 ? w1 : w2, lane, b, acc);
            mfma_settle(acc);
            mfma_hold(b);
#pragma unroll
            for (int u = 0; u < kHalfSub; ++u)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    b[u][rt] = __builtin_elementwise_max(__builtin_convertvector(acc[u][rt], h4_t),
                                                         h4_t{(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f});
        }
        mfma_layer_h<4, 3>(w3, lane, b, acc);
        mfma_settle3(acc);
        mfma_hold(b);
#pragma unroll
        for (int u = 0; u < kHalfSub; ++u)
#pragma unroll
            for (int rt = 0; rt < 3; ++rt)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int o = 16 * rt + 4 * g + c;
                    // the network's outputs are half-precision numbers in the reference
                    if (valid[u] && o < L.n_out) out[(size_t)pt[u] * out_ldp + o * out_ldf] = (float)(_Float16)acc[u][rt][c];
                }
    }
}

// ---- MFMA backward ------------------------------------------------------------------------------
// d_in[k][p] = relu'(h[k][p]) * sum_r W[r][k] d_out[r][p]: the same register-resident chain as the
// forward pass, with A = tiles of W^T (rows = k, row-permuted like the forward fragments) and the
// deltas as B operand; sums over r ascending, i.e. bit-identical to net_backward_kernel.
// fragb[w_off[layer] + (kt * S + s) * 64 + l] = W[4s + (l>>4)][16kt + 4((l&15)&3) + ((l&15)>>2)], S = n_o / 4
__global__ void fragment_mlp_t_kernel(NetLayout L, const float *src, float *dst)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.n_mlp) return;
    int layer = 0;
    while (layer < L.n_hidden && e >= L.w_off[layer + 1]) ++layer;
    const int n_i = layer == 0 ? L.enc : L.n_neurons, n_o = layer == L.n_hidden ? L.n_out_padded : L.n_neurons;
    const int S = n_o / 4;
    const uint32_t f = e - L.w_off[layer];
    const uint32_t l = f & 63u, t = f >> 6;
    const uint32_t kt = t / S, s = t % S;
    const uint32_t i = l & 15u, g = l >> 4;
    const uint32_t k = 16 * kt + 4 * (i & 3u) + (i >> 2), r = 4 * s + g;
    dst[e] = src[L.w_off[layer] + r * n_i + k];
}

template <int ENC, int H, int NH, int NOP, int NT>
__global__ __launch_bounds__(NT) void net_backward_mfma_kernel(NetLayout L, const float *fragb, const float *dl_dout,
                                                                   const unsigned long long *relu_mask, int n, float *deltas,
                                                                   float *denc)
{
    extern __shared__ float lds[];
    float *wfrag = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t e = threadIdx.x; e < L.n_mlp; e += NT) wfrag[e] = fragb[e];
    __syncthreads();
    const int i = lane & 15, g = lane >> 4;
    const int n_tiles = (n + 63) / 64 * 4 / kMfmaSub;      // 16-column units of the training layout
    const int dstride = NOP + NH * H;
    for (int tile = blockIdx.x * (NT / 64) + wave; tile < n_tiles; tile += gridDim.x * (NT / 64)) {
        int pt[kMfmaSub];
        bool valid[kMfmaSub];
        float b[kMfmaSub][16];
        unsigned long long mask[kMfmaSub];
        float *drow[kMfmaSub];      // this lane's column of the unit's delta block
#pragma unroll
        for (int u = 0; u < kMfmaSub; ++u) {
            const int unit = tile * kMfmaSub + u;
            pt[u] = train_point(unit, i);
            valid[u] = pt[u] < n;
            drow[u] = deltas + (size_t)unit * 16 * dstride + i;
            mask[u] = valid[u] ? relu_mask[(size_t)pt[u] * 4 + g] : 0ull;
#pragma unroll
            for (int s = 0; s < NOP / 4; ++s) {
                const int r = 4 * s + g;
                b[u][s] = (valid[u] && r < L.n_out) ? dl_dout[(size_t)pt[u] * L.n_out + r] : 0.0f;
            }
        }
        // stores only after every load of the tile has been issued (a load cannot move above a
        // store it might alias: interleaved, each of the 24 loads was its own memory round trip)
#pragma unroll
        for (int u = 0; u < kMfmaSub; ++u) {
#pragma unroll
            for (int s = 0; s < NOP / 4; ++s)
                if (valid[u]) drow[u][(4 * s + g) * 16] = b[u][s];
        }
        f32x4_t acc[kMfmaSub][4];
#pragma unroll
        for (int layer = NH; layer >= 1; --layer) {
            if (layer == NH) mfma_layer<NOP / 4, H / 16>(wfrag + L.w_off[layer], lane, b, acc);
            else mfma_layer<H / 4, H / 16>(wfrag + L.w_off[layer], lane, b, acc);
#pragma unroll
            for (int u = 0; u < kMfmaSub; ++u)
#pragma unroll
                for (int kt = 0; kt < H / 16; ++kt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int k = 16 * kt + 4 * c + g;
                        const bool on = (mask[u] >> ((layer - 1) * 16 + 4 * kt + c)) & 1ull;      // ReLU'
                        const float v = on ? acc[u][kt][c] : 0.0f;
                        b[u][4 * kt + c] = v;
                        if (valid[u]) drow[u][(NOP + (layer - 1) * H + k) * 16] = v;
                    }
        }
        mfma_layer<H / 4, ENC / 16>(wfrag + L.w_off[0], lane, b, acc);
#pragma unroll
        for (int u = 0; u < kMfmaSub; ++u)
#pragma unroll
            for (int kt = 0; kt < ENC / 16; ++kt)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (valid[u]) denc[(size_t)pt[u] * ENC + 16 * kt + 4 * c + g] = acc[u][kt][c];
    }
}

// ---- backward pass and weight gradients in one kernel ------------------------------------------
// The deltas of a layer are needed twice: as B operand of the next backward layer (held in lane
// (column, g)) and as A operand of that layer's weight gradient (lane (feature, k-slot)).  Here
// they never reach global memory: a block owns a 1024-point chunk, wave w is chain w and walks its
// sixteen 16-column units in order; after every backward layer it turns the 16 x 64 delta block
// around through a 5 KB LDS tile and feeds the weight-gradient MFMAs of that layer, whose other
// operand (the layer's input activations) comes from the training layout in global memory with
// one 16-byte load per four MFMAs.  The 13 312 weight-gradient sums of a chain live in registers
// (208 per lane, accumulation registers: one wave per SIMD) for the whole chunk and are combined
// as ((w0 + w1) + w2) + w3 at the end: the same sums in the same order as weight_grad_mfma_kernel.
// The delta tile is written and read by the SAME wave, and a wave's LDS instructions execute in
// order: the compiler only has to keep them in program order.  (A workgroup fence would also wait
// for every global load in flight, i.e. for the prefetched operands of the next unit.)
__device__ __forceinline__ void wave_lds_order() { asm volatile("" ::: "memory"); }

// One wave per SIMD: nothing hides the LDS latency of a weight fragment unless it is requested well
// ahead of its MFMA, so the fragments of four k-steps are read as a group, one group ahead.
template <int S, int RT>
__device__ __forceinline__ void mfma_layer1(const float *wf, int lane, const float (&b)[16], f32x4_t (&acc)[4])
{
    static_assert(S % 4 == 0, "groups of four k-steps");
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    float a[2][4 * RT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[0][q * RT + rt] = wf[(rt * S + q) * 64 + lane];
#pragma unroll
    for (int s0 = 0; s0 < S; s0 += 4) {
        const int cur = (s0 >> 2) & 1;
        if (s0 + 4 < S) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[cur ^ 1][q * RT + rt] = wf[(rt * S + s0 + 4 + q) * 64 + lane];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][q * RT + rt], b[s0 + q], acc[rt], 0, 0, 0);
    }
}

constexpr int kTileStride = 20;   // floats per feature row of the delta tile (16 columns + padding, 16-byte aligned)

// input activations of one layer for the four groups of a unit: lane (feature 16kt + i, k-slot g)
// loads columns 4g .. 4g+3 in one piece.  Nothing touches the values here (that would wait for
// them): columns of points past the end, which were never written, are zeroed where they are used.
template <int KT>
__device__ __forceinline__ void load_inputs(const float *in_unit, int i, int g, float4 (&b)[KT])
{
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) b[kt] = *reinterpret_cast<const float4 *>(in_unit + (size_t)(16 * kt + i) * 16 + 4 * g);
}

// dW[16rt + ..][16kt + ..] += delta^T (from the LDS tile) x input (loaded by load_inputs) for one unit
template <int RT, int KT>
__device__ __forceinline__ void wgrad_unit(const float *tile, const float4 (&bin)[KT], bool ok0, bool ok1, bool ok2, bool ok3, int i,
                                           int g, f32x4_t (&acc)[RT][KT])
{
    float4 a[RT], b[KT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const float4 *>(tile + (16 * rt + i) * kTileStride + 4 * g);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        b[kt].x = ok0 ? bin[kt].x : 0.0f; b[kt].y = ok1 ? bin[kt].y : 0.0f;
        b[kt].z = ok2 ? bin[kt].z : 0.0f; b[kt].w = ok3 ? bin[kt].w : 0.0f;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const float av = m == 0 ? a[rt].x : m == 1 ? a[rt].y : m == 2 ? a[rt].z : a[rt].w;
                const float bv = m == 0 ? b[kt].x : m == 1 ? b[kt].y : m == 2 ? b[kt].z : b[kt].w;
                acc[rt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[rt][kt], 0, 0, 0);
            }
}

// the four chains of a block -> ((w0 + w1) + w2) + w3 -> one fixed-point atomic per weight
template <int RT, int KT>
__device__ __forceinline__ void wgrad_reduce(float *red, int wave, int lane, int i, int g, int n_i, const f32x4_t (&acc)[RT][KT],
                                             fx_t *gW)
{
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int c = 0; c < 4; ++c) red[((wave - 1) * RT * KT * 4 + (rt * KT + kt) * 4 + c) * 64 + lane] = acc[rt][kt][c];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int e = (rt * KT + kt) * 4 + c;
                    const float v = ((acc[rt][kt][c] + red[(0 * RT * KT * 4 + e) * 64 + lane]) + red[(1 * RT * KT * 4 + e) * 64 + lane]) +
                                    red[(2 * RT * KT * 4 + e) * 64 + lane];
                    if (v != 0.0f) fx_add(gW + (size_t)(16 * rt + 4 * g + c) * n_i + 16 * kt + i, to_fx(v));
                }
    }
}

template <int ENC, int H, int NH, int NOP>
__global__ __launch_bounds__(256, 1) void net_backward_wgrad_kernel(NetLayout L, const float *fragb, const float *dl_dout,
                                                                    const unsigned long long *relu_mask, int n, const float *acts,
                                                                    float *denc, fx_t *grad)
{
    static_assert(NH == 3 && H == 64 && ENC == 32 && NOP == 48, "accumulator sets are spelled out for this shape");
    extern __shared__ float lds[];
    float *wfrag = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *tile = lds + L.n_mlp + wave * (64 * kTileStride);
    float *red = lds + L.n_mlp + 4 * (64 * kTileStride);
    for (uint32_t e = threadIdx.x; e < L.n_mlp; e += 256) wfrag[e] = fragb[e];
    __syncthreads();
    const int i = lane & 15, g = lane >> 4;
    const int astride = ENC + NH * H;
    const int p0 = blockIdx.x * 1024, p1 = min(n, p0 + 1024);
    f32x4_t acc3[NOP / 16][H / 16], acc2[H / 16][H / 16], acc1[H / 16][H / 16], acc0[H / 16][ENC / 16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (a < NOP / 16) acc3[a][c] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            acc2[a][c] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            acc1[a][c] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            if (c < ENC / 16) acc0[a][c] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        }
    // Software pipeline over the units of the chain: with one wave per SIMD nothing else hides a
    // load, so every operand of unit u+1 is requested right after its counterpart of unit u has
    // been consumed (the input activations reuse the same registers), a whole unit ahead of its use.
    struct Unit {
        int pt;
        bool valid, ok0, ok1, ok2, ok3, any;
        const float *aunit;
    };
    auto unit_of = [&](int span) {
        Unit u;
        const int pbase = 64 * span + 4 * wave;
        u.any = span * 64 < p1 && pbase < p1;           // wave-uniform: a group of this chain is left
        u.pt = train_point(span * 4 + wave, i);
        u.valid = u.any && u.pt < p1;
        u.ok0 = u.any && pbase + g < p1; u.ok1 = u.any && pbase + g + 16 < p1;
        u.ok2 = u.any && pbase + g + 32 < p1; u.ok3 = u.any && pbase + g + 48 < p1;
        u.aunit = acts + (size_t)(span * 4 + wave) * 16 * astride;
        return u;
    };
    // requests only: a unit without points reads the chunk's first point, and nothing is selected
    // or masked before the values are used one unit later
    auto load_dl = [&](const Unit &u, float (&d)[NOP / 4], unsigned long long &m) {
        const size_t q = (size_t)(u.valid ? u.pt : p0);
        m = relu_mask[q * 4 + g];
#pragma unroll
        for (int s = 0; s < NOP / 4; ++s) d[s] = dl_dout[q * L.n_out + min(4 * s + g, L.n_out - 1)];
    };
    float4 in3[H / 16], in2[H / 16], in1[H / 16], in0[ENC / 16];
    float ndl[NOP / 4];
    unsigned long long nmask;
    Unit cur = unit_of(p0 / 64);
    load_dl(cur, ndl, nmask);
    if (cur.any) {
        load_inputs<H / 16>(cur.aunit + (size_t)(ENC + 2 * H) * 16, i, g, in3);
        load_inputs<H / 16>(cur.aunit + (size_t)(ENC + H) * 16, i, g, in2);
        load_inputs<H / 16>(cur.aunit + (size_t)ENC * 16, i, g, in1);
        load_inputs<ENC / 16>(cur.aunit, i, g, in0);
    }
    f32x4_t st_acc[ENC / 16];          // encoding gradient of the previous unit, stored one unit late
    int st_pt = 0;
    bool st_valid = false;
    for (int span = p0 / 64; cur.any; ++span) {
        const Unit nxt = unit_of(span + 1);
        const int pt = cur.pt;
        const bool valid = cur.valid;
        const unsigned long long mask = valid ? nmask : 0ull;
        float b[16];
#pragma unroll
        for (int s = 0; s < NOP / 4; ++s) b[s] = (valid && 4 * s + g < L.n_out) ? ndl[s] : 0.0f;
        load_dl(nxt, ndl, nmask);
        // The compiler drains every outstanding memory operation at the end of the loop body (it
        // does not track them across the back edge), so nothing young may be in flight there: the
        // encoding gradient of the PREVIOUS unit is stored now, and the encoding of the next unit
        // (the last operand a unit consumes) is requested now, into a second set of registers.
        if (st_valid) {
#pragma unroll
            for (int kt = 0; kt < ENC / 16; ++kt)
#pragma unroll
                for (int c = 0; c < 4; ++c) denc[(size_t)st_pt * ENC + 16 * kt + 4 * c + g] = st_acc[kt][c];
        }
        float4 in0n[ENC / 16];
        if (nxt.any) load_inputs<ENC / 16>(nxt.aunit, i, g, in0n);
        // ---- output layer: dW3 += dl^T x h3
#pragma unroll
        for (int s = 0; s < NOP / 4; ++s) tile[(4 * s + g) * kTileStride + i] = b[s];
        wave_lds_order();
        wgrad_unit<NOP / 16, H / 16>(tile, in3, cur.ok0, cur.ok1, cur.ok2, cur.ok3, i, g, acc3);
        wave_lds_order();
        if (nxt.any) load_inputs<H / 16>(nxt.aunit + (size_t)(ENC + 2 * H) * 16, i, g, in3);
        f32x4_t acc[4];
        // ---- hidden layers, last to first: delta, then the weight gradient that consumes it
#pragma unroll
        for (int layer = NH; layer >= 1; --layer) {
            if (layer == NH) mfma_layer1<NOP / 4, H / 16>(wfrag + L.w_off[layer], lane, b, acc);
            else mfma_layer1<H / 4, H / 16>(wfrag + L.w_off[layer], lane, b, acc);
#pragma unroll
            for (int kt = 0; kt < H / 16; ++kt)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool on = (mask >> ((layer - 1) * 16 + 4 * kt + c)) & 1ull;      // ReLU'
                    const float v = on ? acc[kt][c] : 0.0f;
                    b[4 * kt + c] = v;
                    tile[(16 * kt + 4 * c + g) * kTileStride + i] = v;
                }
            wave_lds_order();
            // delta of hidden layer (layer - 1) x its input: the encoding for the first, else the layer before
            if (layer == 3) {
                wgrad_unit<H / 16, H / 16>(tile, in2, cur.ok0, cur.ok1, cur.ok2, cur.ok3, i, g, acc2);
                if (nxt.any) load_inputs<H / 16>(nxt.aunit + (size_t)(ENC + H) * 16, i, g, in2);
            } else if (layer == 2) {
                wgrad_unit<H / 16, H / 16>(tile, in1, cur.ok0, cur.ok1, cur.ok2, cur.ok3, i, g, acc1);
                if (nxt.any) load_inputs<H / 16>(nxt.aunit + (size_t)ENC * 16, i, g, in1);
            } else {
                wgrad_unit<H / 16, ENC / 16>(tile, in0, cur.ok0, cur.ok1, cur.ok2, cur.ok3, i, g, acc0);
            }
            wave_lds_order();
        }
        // ---- gradient of the encoding, for grid_grad_kernel
        mfma_layer1<H / 4, ENC / 16>(wfrag + L.w_off[0], lane, b, acc);
#pragma unroll
        for (int kt = 0; kt < ENC / 16; ++kt) {
            st_acc[kt] = acc[kt];
            in0[kt] = in0n[kt];
        }
        st_pt = pt;
        st_valid = valid;
        cur = nxt;
    }
    if (st_valid) {
#pragma unroll
        for (int kt = 0; kt < ENC / 16; ++kt)
#pragma unroll
            for (int c = 0; c < 4; ++c) denc[(size_t)st_pt * ENC + 16 * kt + 4 * c + g] = st_acc[kt][c];
    }
    wgrad_reduce<NOP / 16, H / 16>(red, wave, lane, i, g, H, acc3, grad + L.w_off[3]);
    wgrad_reduce<H / 16, H / 16>(red, wave, lane, i, g, H, acc2, grad + L.w_off[2]);
    wgrad_reduce<H / 16, H / 16>(red, wave, lane, i, g, H, acc1, grad + L.w_off[1]);
    wgrad_reduce<H / 16, ENC / 16>(red, wave, lane, i, g, ENC, acc0, grad + L.w_off[0]);
}

// dW[r][k] += sum_p delta[p][r] * input[p][k] as MFMA over the point index: A = delta^T tile
// (16 rows r x 4 points), B = input tile (4 points x 16 columns k).  One BLOCK owns a chunk of
// 1024 consecutive points; its four waves take the 4-point groups round robin (wave w: groups
// w, w+4, ..), each wave owns all (row tile, column tile) pairs: 4 + 4 operand loads feed 16 MFMAs
// per group and 16 independent accumulators hide the MFMA latency.  The four partial sums are
// added in the fixed order ((w0 + w1) + w2) + w3 through LDS and leave through one fixed-point
// atomic per weight -- the summation order every implementation of this gradient follows.
template <int N_O, int N_I>
__global__ __launch_bounds__(256) void weight_grad_mfma_kernel(const float *delta, int dstride, int doff, const float *input,
                                                               int istride, int ioff, int n, int chunk, fx_t *gW)
{
    constexpr int RT = N_O / 16, KT = N_I / 16;
    __shared__ float red[3][RT * KT * 4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int p0 = blockIdx.x * chunk, p1 = min(n, p0 + chunk);
    f32x4_t acc[RT][KT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc[rt][kt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    // wave w = chain w: per 64-point span one 16-column unit, i.e. the groups w, w+4, w+8, w+12
    for (int span = p0 / 64; span * 64 < p1; ++span) {
        const int unit = span * 4 + wave;
        const float *dp = delta + (size_t)unit * 16 * dstride + (size_t)(doff + i) * 16 + 4 * g;
        const float *ip = input + (size_t)unit * 16 * istride + (size_t)(ioff + i) * 16 + 4 * g;
        float4 a[RT], b[KT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const float4 *>(dp + 16 * 16 * rt);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) b[kt] = *reinterpret_cast<const float4 *>(ip + 16 * 16 * kt);
        // columns of points past the end were never written
        const int pbase = 64 * span + 4 * wave + g;
        const bool ok0 = pbase < p1, ok1 = pbase + 16 < p1, ok2 = pbase + 32 < p1, ok3 = pbase + 48 < p1;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            a[rt].x = ok0 ? a[rt].x : 0.0f; a[rt].y = ok1 ? a[rt].y : 0.0f;
            a[rt].z = ok2 ? a[rt].z : 0.0f; a[rt].w = ok3 ? a[rt].w : 0.0f;
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            b[kt].x = ok0 ? b[kt].x : 0.0f; b[kt].y = ok1 ? b[kt].y : 0.0f;
            b[kt].z = ok2 ? b[kt].z : 0.0f; b[kt].w = ok3 ? b[kt].w : 0.0f;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (pbase - g + 16 * m >= p1) break;       // wave-uniform: no group left in this span
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const float av = m == 0 ? a[rt].x : m == 1 ? a[rt].y : m == 2 ? a[rt].z : a[rt].w;
                    const float bv = m == 0 ? b[kt].x : m == 1 ? b[kt].y : m == 2 ? b[kt].z : b[kt].w;
                    acc[rt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[rt][kt], 0, 0, 0);
                }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int c = 0; c < 4; ++c) red[wave - 1][(rt * KT + kt) * 4 + c][lane] = acc[rt][kt][c];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = (rt * KT + kt) * 4 + c;
                const float v = ((acc[rt][kt][c] + red[0][e][lane]) + red[1][e][lane]) + red[2][e][lane];
                if (v != 0.0f) fx_add(gW + (size_t)(16 * rt + 4 * g + c) * N_I + 16 * kt + i, to_fx(v));
            }
}

// backward, one thread per point: propagates dL/dout to every layer input (kept per point in
// `deltas`: n x (n_out_padded + n_hidden * n_neurons)) and to the encoding (`denc`: n x enc).
// Weight and grid gradients are formed afterwards by weight_grad_kernel / grid_grad_kernel.
__global__ __launch_bounds__(kNetBlock) void net_backward_kernel(NetLayout L, const float *params, const float *xy,
                                                                 const float *dl_dout, const float *acts, int n, float *deltas,
                                                                 float *denc)
{
    extern __shared__ float lds[];
    float *col_a = lds + threadIdx.x;
    float *col_b = lds + 64 * kNetBlock + threadIdx.x;
    const int p = blockIdx.x * kNetBlock + threadIdx.x;
    const bool valid = p < n;
    const int astride = L.enc + L.n_hidden * L.n_neurons;
    const int dstride = L.n_out_padded + L.n_hidden * L.n_neurons;
    // delta of the output layer
    for (int k = 0; k < L.n_out_padded; ++k) {
        const float g = (valid && k < L.n_out) ? dl_dout[(size_t)p * L.n_out + k] : 0.0f;
        act(col_a, k) = g;
        if (valid) deltas[(size_t)p * dstride + k] = g;
    }
    float *d_out = col_a, *d_in = col_b;
    int n_o = L.n_out_padded;
    for (int layer = L.n_hidden; layer >= 1; --layer) {
        // d_in[k] = relu'(h[k]) * sum_r W[r][k] d_out[r], h = output of hidden layer (layer-1)
        const float *W = params + L.w_off[layer];
        const int n_i = L.n_neurons;
        for (int k0 = 0; k0 < n_i; k0 += 8) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < n_o; ++r) {
                const float g = d_out[r * kNetBlock];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(W[(size_t)r * n_i + k0 + j], g, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float h = valid ? acts[(size_t)p * astride + L.enc + (layer - 1) * L.n_neurons + k0 + j] : 0.0f;
                const float v = h > 0.0f ? acc[j] : 0.0f;
                d_in[(k0 + j) * kNetBlock] = v;
                if (valid) deltas[(size_t)p * dstride + L.n_out_padded + (layer - 1) * L.n_neurons + k0 + j] = v;
            }
        }
        float *t = d_out; d_out = d_in; d_in = t;
        n_o = L.n_neurons;
    }
    // encoding gradient: d_enc[k] = sum_r W1[r][k] d_out[r]
    {
        const float *W = params + L.w_off[0];
        for (int k0 = 0; k0 < L.enc; k0 += 8) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < L.n_neurons; ++r) {
                const float g = d_out[r * kNetBlock];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(W[(size_t)r * L.enc + k0 + j], g, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) d_in[(k0 + j) * kNetBlock] = acc[j];
        }
    }
    if (!valid) return;
    for (int k = 0; k < L.enc; ++k) denc[(size_t)p * L.enc + k] = d_in[k * kNetBlock];
}

// Gradient of the grid: every point adds w_corner * d_enc to the 4 corners of its cell on every
// level.  Training points arrive in pixel order, so neighbouring lanes hit the SAME entries:
// direct global float atomics cost 39 ms per 524 288-sample batch.  Instead a launch takes a
// group of consecutive levels [lv0, lv1) whose accumulators fit into LDS, a block accumulates its
// chunk of points there (64-bit fixed point, ds_add_u64) and flushes each touched entry with ONE
// global atomic; a level too large for LDS is split by feature range [q0, q1).
// Work item = (point, level): neighbouring lanes read consecutive float4s of d_enc and hit
// different levels.  use_lds = 0 (a level too large for LDS) scatters straight to global memory.
constexpr int kGridGradBlock = 1024;
// denc: dL/d(encoding) of point p, level lv at denc + p * ld_point + lv * ld_level (rows of the encoding per point in the
// fp32 path; level-major in the half-precision path, where every launch then reads only the levels it works on)
// the work of one block: the points [p0, p0 + chunk) against the levels [lv0, lv1), features [q0, q1)
// `replicas` (a power of two, LDS accumulators only): that many copies of the accumulators, a lane adds to copy lane % replicas --
// on the coarse levels the 64 lanes of a wave fall into a few dozen cells, and LDS atomics on one address are served one by one
__device__ __forceinline__ void grid_grad_block(const NetLayout &L, const float *xy, const float *denc, size_t ld_point, size_t ld_level, int n, int p0,
                                                int chunk, int lv0, int lv1, int q0, int q1, int use_lds, fx_t *grad, fx_t *acc, int replicas = 1)
{
    __shared__ float s_scale[kNetMaxLevels];
    __shared__ uint32_t s_res[kNetMaxLevels], s_off[kNetMaxLevels + 1];
    if (threadIdx.x <= (unsigned)L.n_levels) {
        s_off[threadIdx.x] = L.level_off[threadIdx.x];
        if (threadIdx.x < (unsigned)L.n_levels) {
            s_scale[threadIdx.x] = L.scale[threadIdx.x];
            s_res[threadIdx.x] = (uint32_t)L.res[threadIdx.x];
        }
    }
    const int nf = L.n_features, nq = q1 - q0;                          // features [q0, q1) of every entry
    const uint32_t base = L.level_off[lv0];                             // first entry of the group
    // accumulators feature-major: [feature][entry] -- entry-major (four 8-byte words per entry) put the lanes of every
    // atomic instruction 32 bytes apart, on four of the LDS banks
    const int n_ent = (int)(L.level_off[lv1] - base);
    const int n_acc = use_lds ? n_ent * nq : 0;
    for (int e = threadIdx.x; e < n_acc * replicas; e += kGridGradBlock) acc[e] = 0;
    acc += (size_t)(threadIdx.x & (unsigned)(replicas - 1)) * n_acc;      // this lane's copy
    __syncthreads();
    const int p1 = min(n, p0 + chunk);
    fx_t *gG = grad + L.n_mlp;
    const int n_lv = lv1 - lv0;
    // The sums are integers, so the order of the additions is free: consecutive work items take
    // points that lie chunk / 64 apart (chunk is a multiple of 64).  Neighbouring training points
    // are neighbouring pixels and fall into the same grid cells; spread out, the lanes of a wave
    // mostly hit different accumulators instead of queueing on one.
    const int spread = chunk >> 6;
    const int n_items = chunk * n_lv;
    for (int item = threadIdx.x; item < n_items; item += kGridGradBlock) {
        const int pi = item / n_lv, lv = lv0 + item % n_lv;
        const int p = p0 + (pi & 63) * spread + (pi >> 6);
        if (p >= p1) continue;
        const float s = s_scale[lv];
        const uint32_t res = s_res[lv], lo = s_off[lv], n_level = s_off[lv + 1] - lo;
        if (L.dims == 3) {
            // three inputs: the eight corners of the cell, weights as in encode_point; these levels are far larger than LDS
            // (res^3 entries), so use_lds is 0 for all but the coarsest and the sums go straight to global memory
            const float *q3 = xy + 3 * (size_t)p;
            float pf[3];
            uint32_t pi[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = __builtin_fmaf(s, q3[a], 0.5f), fl = floorf(v);
                pf[a] = v - fl;
                pi[a] = (uint32_t)(int)fl;
            }
            const float *d = denc + (size_t)p * ld_point + (size_t)lv * ld_level;
            float dv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) dv[q] = q < nf ? d[q] : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float w = (((k & 1) ? pf[0] : 1.0f - pf[0]) * ((k & 2) ? pf[1] : 1.0f - pf[1])) * ((k & 4) ? pf[2] : 1.0f - pf[2]);
                const uint32_t idx = (uint32_t)(((unsigned long long)(pi[0] + (k & 1)) + (unsigned long long)(pi[1] + ((k >> 1) & 1)) * res +
                                                 (unsigned long long)(pi[2] + ((k >> 2) & 1)) * res * res) % n_level);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (q < q0 || q >= q1) continue;
                    const fx_t v = to_fx(w * dv[q]);
                    if (use_lds) fx_add(&acc[(q - q0) * n_ent + (int)(lo + idx - base)], v);
                    else fx_add(gG + (size_t)(lo + idx) * nf + q, v);
                }
            }
            continue;
        }
        float px = __builtin_fmaf(s, xy[2 * p], 0.5f), py = __builtin_fmaf(s, xy[2 * p + 1], 0.5f);
        const float fx = floorf(px), fy = floorf(py);
        px -= fx;
        py -= fy;
        const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
        // the point's gradient values first: loaded inside the corner loop, each load would wait
        // for the atomics before it (they may alias) -- 16 dependent memory round trips per item
        const float *d = denc + (size_t)p * ld_point + (size_t)lv * ld_level;
        float dv[8];
        if (nf == 4) {
            const float4 d4 = *reinterpret_cast<const float4 *>(d);
            dv[0] = d4.x; dv[1] = d4.y; dv[2] = d4.z; dv[3] = d4.w;
            dv[4] = dv[5] = dv[6] = dv[7] = 0.0f;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) dv[q] = q < nf ? d[q] : 0.0f;
        }
        uint32_t entry[4];
        float w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t cx = ix + (k & 1), cy = iy + (k >> 1);
            w[k] = ((k & 1) ? px : 1.0f - px) * ((k >> 1) ? py : 1.0f - py);
            uint32_t idx = cx + cy * res;              // % n_level, a compare-subtract inside the unit square
            if (idx >= n_level) {
                idx -= n_level;
                if (idx >= n_level) idx %= n_level;
            }
            entry[k] = lo + idx;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q < q0 || q >= q1) continue;
                const fx_t v = to_fx(w[k] * dv[q]);
                if (use_lds) fx_add(&acc[(q - q0) * n_ent + (int)(entry[k] - base)], v);
                else fx_add(gG + (size_t)entry[k] * nf + q, v);
            }
        }
    }
    __syncthreads();
    acc -= (size_t)(threadIdx.x & (unsigned)(replicas - 1)) * n_acc;
    for (int e = threadIdx.x; e < n_acc; e += kGridGradBlock) {
        fx_t v = acc[e];
        for (int r = 1; r < replicas; ++r) v += acc[e + (size_t)r * n_acc];
        if (v != 0) fx_add(gG + (size_t)(base + e % n_ent) * nf + q0 + e / n_ent, v);
    }
}

__global__ __launch_bounds__(kGridGradBlock) void grid_grad_kernel(NetLayout L, const float *xy, const float *denc, size_t ld_point, size_t ld_level,
                                                                   int n, int chunk, int lv0, int lv1, int q0, int q1, int use_lds, fx_t *grad)
{
    extern __shared__ fx_t acc[];
    grid_grad_block(L, xy, denc, ld_point, ld_level, n, (int)blockIdx.x * chunk, chunk, lv0, lv1, q0, q1, use_lds, grad, acc);
}

// ALL level groups of a training step in ONE launch.  The five launches above ran one after the other, each a single wave of
// blocks (one per CU, its accumulators up to 150 KB of LDS) that spent as long flushing its accumulators as filling them:
// 137 us of a 400-us step.  Here every group's accumulators fit 72 KB (a larger level is cut into feature slices), so two
// blocks share a CU, and a block of group g takes chunk[g] points with chunk[g] x (atomics per point of g) about equal for
// all groups: the whole grid is resident at once, every block flushes once.  The sums are integers: any split is bit-exact.
struct GridGradPlan {
    int32_t n_groups;
    int32_t lv0[16], lv1[16], q0[16], q1[16], chunk[16], replicas[16], first_block[17];
};
__global__ __launch_bounds__(kGridGradBlock, 2) void grid_grad_plan_kernel(NetLayout L, const float *xy, const float *denc, size_t ld_point, size_t ld_level,
                                                                           int n, GridGradPlan plan, fx_t *grad)
{
    extern __shared__ fx_t acc[];
    int g = 0;
    while (g + 1 < plan.n_groups && (int)blockIdx.x >= plan.first_block[g + 1]) ++g;
    const int chunk = plan.chunk[g];
    grid_grad_block(L, xy, denc, ld_point, ld_level, n, ((int)blockIdx.x - plan.first_block[g]) * chunk, chunk, plan.lv0[g], plan.lv1[g], plan.q0[g],
                    plan.q1[g], 1, grad, acc, plan.replicas[g]);
}

// ---- three inputs: ALL levels of a step's grid gradient in one launch, through spatial bins ------------------------------------
// The trilinear grids of GuidedIntegrator<3> do not fit LDS (up to 2^19 entries a level), and round 4 / 5 added their upper levels
// straight to global memory: 128 sixty-four-bit atomics per point that queue on the cells the walkers share -- 1.5 ms of a 2-ms
// training step at 2 x 10^5 points, 7 to 10 ms at 5 x 10^5 (profiles/r05_ab_grid_grad_launches.txt).  Here the points are first
// put into bins^3 boxes of the unit cube (count, table, scatter: any order inside a box, the sums are integers).  All points of a
// box touch, on level l, a sub-grid of at most floor(scale_l / bins) + 3 cells per axis: with 8 boxes per axis the sub-grids of all
// eight levels of the reference's network are 4 205 entries x 4 features = 135 KB of accumulators.  A block takes up to `chunk`
// points of ONE box, adds them up in LDS and flushes what is not zero with one global atomic per entry and feature.
struct GridBin3Plan {
    int32_t bins, chunk, n_acc;            // boxes per axis (a power of two); points per block; accumulator entries (x n_features)
    int32_t ext[kNetMaxLevels], acc_off[kNetMaxLevels + 1];
};
// counts[0 .. nb): points per box; then [nb .. 2 nb): scatter cursors; [2 nb]: ticket of the count kernel; after that, written by
// the last block of the count kernel: start[nb + 1] (first point of a box in `order`) and first_block[nb + 1]
__device__ __forceinline__ int grid_bin3_of(const float *q3, int bins)
{
    int b[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) b[a] = max(0, min(bins - 1, (int)(q3[a] * (float)bins)));
    return b[0] + bins * (b[1] + bins * b[2]);
}
constexpr int kBin3CountPoints = 4096;      // points of a count / scatter block
__global__ __launch_bounds__(1024) void grid_bin3_count_kernel(const float *xyz, int n, int bins, int chunk, uint32_t *tab)
{
    extern __shared__ uint32_t s_cnt[];
    const int nb = bins * bins * bins;
    for (int b = threadIdx.x; b < nb; b += 1024) s_cnt[b] = 0;
    __syncthreads();
    const int p0 = blockIdx.x * kBin3CountPoints;
    for (int p = p0 + threadIdx.x; p < min(n, p0 + kBin3CountPoints); p += 1024) atomicAdd(&s_cnt[grid_bin3_of(xyz + 3 * (size_t)p, bins)], 1u);
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += 1024)
        if (s_cnt[b]) atomicAdd(&tab[b], s_cnt[b]);
    __threadfence();
    __syncthreads();
    __shared__ uint32_t s_last;
    if (threadIdx.x == 0) s_last = atomicAdd(&tab[2 * nb], 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // the table, by the block that finished last: one thread per box group, two running sums (nb <= 4096: four boxes a thread)
    uint32_t *start = tab + 2 * nb + 1, *first_block = start + nb + 1;
    const int per = (nb + 1023) / 1024;
    uint32_t pts = 0, blks = 0;
    for (int k = 0; k < per; ++k) {
        const int b = threadIdx.x * per + k;
        if (b < nb) {
            const uint32_t c = __hip_atomic_load(&tab[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pts += c;
            blks += (c + chunk - 1) / chunk;
        }
    }
    uint32_t *s_p = s_cnt, *s_b = s_cnt + 1024;      // (nb >= 512 words of LDS are there: the launch asks for max(nb, 2048))
    s_p[threadIdx.x] = pts;
    s_b[threadIdx.x] = blks;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t a = threadIdx.x >= (unsigned)d ? s_p[threadIdx.x - d] : 0u, c = threadIdx.x >= (unsigned)d ? s_b[threadIdx.x - d] : 0u;
        __syncthreads();
        s_p[threadIdx.x] += a;
        s_b[threadIdx.x] += c;
        __syncthreads();
    }
    uint32_t run_p = s_p[threadIdx.x] - pts, run_b = s_b[threadIdx.x] - blks;
    for (int k = 0; k < per; ++k) {
        const int b = threadIdx.x * per + k;
        if (b < nb) {
            const uint32_t c = __hip_atomic_load(&tab[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            start[b] = run_p;
            first_block[b] = run_b;
            run_p += c;
            run_b += (c + chunk - 1) / chunk;
        }
    }
    if (threadIdx.x == 1023) {
        start[nb] = s_p[1023];
        first_block[nb] = s_b[1023];
    }
}
__global__ __launch_bounds__(1024) void grid_bin3_scatter_kernel(const float *xyz, int n, int bins, uint32_t *tab, uint32_t *order)
{
    extern __shared__ uint32_t s_cnt[];        // [nb] counts of this block's points, then [nb] their first slot
    const int nb = bins * bins * bins;
    uint32_t *s_at = s_cnt + nb;
    for (int b = threadIdx.x; b < nb; b += 1024) s_cnt[b] = 0;
    __syncthreads();
    const int p0 = blockIdx.x * kBin3CountPoints, p1 = min(n, p0 + kBin3CountPoints);
    int mine[kBin3CountPoints / 1024];
#pragma unroll
    for (int k = 0; k < kBin3CountPoints / 1024; ++k) {
        const int p = p0 + threadIdx.x + 1024 * k;
        mine[k] = p < p1 ? grid_bin3_of(xyz + 3 * (size_t)p, bins) : -1;
        if (mine[k] >= 0) atomicAdd(&s_cnt[mine[k]], 1u);
    }
    __syncthreads();
    const uint32_t *start = tab + 2 * nb + 1;
    for (int b = threadIdx.x; b < nb; b += 1024) s_at[b] = s_cnt[b] ? start[b] + atomicAdd(&tab[nb + b], s_cnt[b]) : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kBin3CountPoints / 1024; ++k)
        if (mine[k] >= 0) order[atomicAdd(&s_at[mine[k]], 1u)] = (uint32_t)(p0 + threadIdx.x + 1024 * k);
}
__global__ __launch_bounds__(kGridGradBlock) void grid_bin3_accumulate_kernel(NetLayout L, const float *xyz, const float *denc, size_t ld_point, size_t ld_level,
                                                                               GridBin3Plan plan, const uint32_t *tab, const uint32_t *order, fx_t *grad)
{
    extern __shared__ fx_t acc[];
    __shared__ float s_scale[kNetMaxLevels];
    __shared__ uint32_t s_res[kNetMaxLevels], s_off[kNetMaxLevels + 1];
    __shared__ int s_lo[kNetMaxLevels][3];
    __shared__ int s_box, s_from, s_to;
    const int bins = plan.bins, nb = bins * bins * bins, nf = L.n_features, n_lv = L.n_levels;
    const uint32_t *start = tab + 2 * nb + 1, *first_block = start + nb + 1;
    if ((uint32_t)blockIdx.x >= first_block[nb]) return;
    if (threadIdx.x == 0) {
        int lo = 0, hi = nb;                   // the last box whose first block is <= this one (empty boxes share a first block with
        while (hi - lo > 1) {                  // the next box: the last of them is the one that has blocks)
            const int mid = (lo + hi) >> 1;
            if (first_block[mid] <= (uint32_t)blockIdx.x) lo = mid;
            else hi = mid;
        }
        s_box = lo;
        s_from = (int)(start[lo] + ((uint32_t)blockIdx.x - first_block[lo]) * (uint32_t)plan.chunk);
        s_to = (int)min(start[lo + 1], (uint32_t)s_from + (uint32_t)plan.chunk);
    }
    if (threadIdx.x <= (unsigned)n_lv) {
        s_off[threadIdx.x] = L.level_off[threadIdx.x];
        if (threadIdx.x < (unsigned)n_lv) {
            s_scale[threadIdx.x] = L.scale[threadIdx.x];
            s_res[threadIdx.x] = (uint32_t)L.res[threadIdx.x];
        }
    }
    for (int e = threadIdx.x; e < plan.n_acc * nf; e += kGridGradBlock) acc[e] = 0;
    __syncthreads();
    if (threadIdx.x < (unsigned)(3 * n_lv)) {
        // the first cell of the box on every level: positions of the box are >= b / bins (exact: bins is a power of two), the cell
        // index floor(fma(scale, x, 0.5)) is monotone in x
        const int lv = threadIdx.x / 3, a = threadIdx.x % 3;
        const int b = a == 0 ? s_box % bins : (a == 1 ? (s_box / bins) % bins : s_box / (bins * bins));
        s_lo[lv][a] = (int)floorf(__builtin_fmaf(s_scale[lv], (float)b / (float)bins, 0.5f));
    }
    __syncthreads();
    fx_t *gG = grad + L.n_mlp;
    const int from = s_from, n_pts = s_to - s_from;
    for (int item = threadIdx.x; item < n_pts * n_lv; item += kGridGradBlock) {
        const int lv = item % n_lv;
        const uint32_t p = order[from + item / n_lv];
        const float s = s_scale[lv];
        const uint32_t res = s_res[lv], lo = s_off[lv], n_level = s_off[lv + 1] - lo;
        const float *q3 = xyz + 3 * (size_t)p;
        float pf[3];
        int pi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = __builtin_fmaf(s, q3[a], 0.5f), fl = floorf(v);
            pf[a] = v - fl;
            pi[a] = (int)fl;
        }
        const float *d = denc + (size_t)p * ld_point + (size_t)lv * ld_level;
        float dv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) dv[q] = q < nf ? d[q] : 0.0f;
        const int ext = plan.ext[lv], cells = ext * ext * ext;
        const int lx = pi[0] - s_lo[lv][0], ly = pi[1] - s_lo[lv][1], lz = pi[2] - s_lo[lv][2];
        // (a point outside its box's sub-grid -- not met; it would take rounding the table does not allow for -- adds to global memory)
        const bool inside = lx >= 0 && ly >= 0 && lz >= 0 && lx + 1 < ext && ly + 1 < ext && lz + 1 < ext;
        fx_t *a_lv = acc + (size_t)plan.acc_off[lv] * nf;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float w = (((k & 1) ? pf[0] : 1.0f - pf[0]) * ((k & 2) ? pf[1] : 1.0f - pf[1])) * ((k & 4) ? pf[2] : 1.0f - pf[2]);
            if (inside) {
                const int cell = (lx + (k & 1)) + ext * ((ly + ((k >> 1) & 1)) + ext * (lz + ((k >> 2) & 1)));
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < nf) fx_add(a_lv + q * cells + cell, to_fx(w * dv[q]));
            } else {
                const uint32_t idx = (uint32_t)(((unsigned long long)((uint32_t)pi[0] + (k & 1)) + (unsigned long long)((uint32_t)pi[1] + ((k >> 1) & 1)) * res +
                                                 (unsigned long long)((uint32_t)pi[2] + ((k >> 2) & 1)) * res * res) % n_level);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < nf) fx_add(gG + (size_t)(lo + idx) * nf + q, to_fx(w * dv[q]));
            }
        }
    }
    __syncthreads();
    for (int lv = 0; lv < n_lv; ++lv) {
        const int ext = plan.ext[lv], cells = ext * ext * ext;
        const uint32_t res = s_res[lv], lo = s_off[lv], n_level = s_off[lv + 1] - lo;
        const fx_t *a_lv = acc + (size_t)plan.acc_off[lv] * nf;
        for (int e = threadIdx.x; e < cells * nf; e += kGridGradBlock) {
            const fx_t v = a_lv[e];
            if (v == 0) continue;
            const int q = e / cells, cell = e % cells;
            const uint32_t cx = (uint32_t)(s_lo[lv][0] + cell % ext), cy = (uint32_t)(s_lo[lv][1] + (cell / ext) % ext), cz = (uint32_t)(s_lo[lv][2] + cell / (ext * ext));
            const uint32_t idx = (uint32_t)(((unsigned long long)cx + (unsigned long long)cy * res + (unsigned long long)cz * res * res) % n_level);
            fx_add(gG + (size_t)(lo + idx) * nf + q, v);
        }
    }
}

// dW[r][k] += sum_p delta[p][r] * input[p][k] for one layer: a block owns a chunk of points,
// thread t owns the (r, k) pairs {t, t + 256, ...}; per pair four fmaf chains over the 4-point
// groups taken round robin, added as ((c0 + c1) + c2) + c3 -- the order of the MFMA kernel.
__global__ __launch_bounds__(256) void weight_grad_kernel(const float *delta, int dstride, int doff, const float *input,
                                                          int istride, int ioff, int n_o, int n_i, int n, int chunk,
                                                          fx_t *gW)
{
    extern __shared__ float lds[];
    float *sd = lds;                 // [32][n_o]
    float *si = lds + 32 * 64;       // [32][n_i]
    const int p0 = blockIdx.x * chunk;
    const int p1 = min(n, p0 + chunk);
    const int n_pairs = n_o * n_i;
    float acc[16][4];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int w = 0; w < 4; ++w) acc[j][w] = 0.0f;
    for (int pb = p0; pb < p1; pb += 32) {
        const int cnt = min(32, p1 - pb);
        __syncthreads();
        for (int e = threadIdx.x; e < 32 * n_o; e += 256) {
            const int q = e / n_o, r = e % n_o;
            sd[q * 64 + r] = q < cnt ? delta[(size_t)(pb + q) * dstride + doff + r] : 0.0f;
        }
        for (int e = threadIdx.x; e < 32 * n_i; e += 256) {
            const int q = e / n_i, k = e % n_i;
            si[q * 64 + k] = q < cnt ? input[(size_t)(pb + q) * istride + ioff + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int pair = threadIdx.x + 256 * j;
            if (pair < n_pairs) {
                const int r = pair / n_i, k = pair % n_i;
#pragma unroll
                for (int q = 0; q < 32; ++q) acc[j][(q >> 2) & 3] = __builtin_fmaf(sd[q * 64 + r], si[q * 64 + k], acc[j][(q >> 2) & 3]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int pair = threadIdx.x + 256 * j;
        const float v = ((acc[j][0] + acc[j][1]) + acc[j][2]) + acc[j][3];
        if (pair < n_pairs && v != 0.0f) fx_add(gW + pair, to_fx(v));
    }
}

// transposed copies of the MLP matrices for the forward pass: dst[k][r] = src[r][k] per layer
__global__ void transpose_mlp_kernel(NetLayout L, const float *src, float *dst)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n_mlp) return;
    int layer = 0;
    while (layer < L.n_hidden && i >= L.w_off[layer + 1]) ++layer;
    const int n_i = layer == 0 ? L.enc : L.n_neurons, n_o = layer == L.n_hidden ? L.n_out_padded : L.n_neurons;
    const uint32_t e = i - L.w_off[layer];
    const uint32_t r = e / n_i, k = e % n_i;
    dst[L.w_off[layer] + k * n_o + r] = src[i];
}

// tiny-cuda-nn's adam_step (optimizers/adam.h) nested in its EMA optimizer (optimizers/ema.h), which is
// what the reference trains with (data/ladybug/n.json:69-81), step counts from 1: L2 regularisation
// only on the matrix weights; an encoding parameter whose gradient is exactly zero is not touched (no
// moment decay, no step); every parameter debiases with ITS OWN step count (lr_table[s] =
// lr sqrt(1 - b2^s) / (1 - b1^s), tabulated on the host so that the oracle's powf and this kernel agree
// bit for bit); the EMA runs over all parameters.  grad_div: number of ranks whose gradients were
// summed (shared-network mode).
// The copies of the MLP matrices in other orders (transposed for the scalar forward pass, MFMA
// fragments for the forward and backward kernels) are written here too: the inverse of the index
// maps of transpose_mlp_kernel / fragment_mlp_kernel / fragment_mlp_t_kernel, five launches less
// per optimizer step.
struct DerivedLayouts {
    float *params_t, *inference_t, *params_f, *inference_f, *params_fb;   // any may be null
    // half-precision images (wost_net_half.h), as arrays of halves: fragments of the matrices then the grid (inference
    // and training weights), fragments of the transposed training matrices
    _Float16 *inference_h, *params_h, *params_hb;
};

__global__ void optimizer_kernel(NetLayout L, uint32_t n, float *params, float *m1, float *m2, float *ema_raw, float *inference,
                                 const fx_t *grad, const float *lr_table, uint32_t *param_steps, float beta1, float beta2,
                                 float eps, float l2, float decay, float debias, float loss_scale, float grad_div,
                                 DerivedLayouts D)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float w = params[i];
    float g = (float)((double)grad[i] / kFxScale) / loss_scale;
    if (grad_div != 1.0f) g = g / grad_div;
    float nw = w;
    if (i < L.n_mlp || g != 0.0f) {
        if (i < L.n_mlp) g += l2 * w;
        const float a = m1[i] = beta1 * m1[i] + (1.0f - beta1) * g;
        const float b = m2[i] = beta2 * m2[i] + (1.0f - beta2) * (g * g);
        const uint32_t s = ++param_steps[i];
        nw = w - (lr_table[s] / (sqrtf(b) + eps)) * a;
        params[i] = nw;
    }
    const float e = ema_raw[i] = decay * ema_raw[i] + (1.0f - decay) * nw;
    const float inf = e * debias;
    inference[i] = inf;
    if (i >= L.n_mlp) {
        // grid entry (i - n_mlp) / 4, feature (i - n_mlp) % 4 of the images: the same index as the parameter
        if (D.inference_h) D.inference_h[i] = (_Float16)inf;
        if (D.params_h) D.params_h[i] = (_Float16)nw;
        return;
    }
    int layer = 0;
    while (layer < L.n_hidden && i >= L.w_off[layer + 1]) ++layer;
    const uint32_t n_i = layer == 0 ? L.enc : L.n_neurons, n_o = layer == L.n_hidden ? L.n_out_padded : L.n_neurons;
    const uint32_t off = L.w_off[layer], q = i - off, r = q / n_i, k = q % n_i;
    if (D.params_t) {
        D.params_t[off + k * n_o + r] = nw;
        D.inference_t[off + k * n_o + r] = inf;
    }
    if (D.params_f) {
        // forward fragments: row 16rt + 4(i&3) + (i>>2) of lane i, k = 4s + g
        const uint32_t rem = r & 15u, ii = (rem & 3u) * 4u + (rem >> 2), S = n_i / 4;
        const uint32_t f = off + ((r >> 4) * S + (k >> 2)) * 64u + (k & 3u) * 16u + ii;
        D.params_f[f] = nw;
        D.inference_f[f] = inf;
        // backward fragments (transposed matrix): k = 16kt + 4(i&3) + (i>>2), r = 4s + g
        const uint32_t remk = k & 15u, ik = (remk & 3u) * 4u + (remk >> 2), Sb = n_o / 4;
        D.params_fb[off + ((k >> 4) * Sb + (r >> 2)) * 64u + (r & 3u) * 16u + ik] = nw;
    }
    if (D.inference_h || D.params_h) {
        // fragment_mlp_h_kernel: entry w_off / 4 + (rt KT + kt) 64 + lane holds W[16 rt + i][16 kt + 4 g .. + 3], lane = (i, g)
        const uint32_t KT = n_i / 16, e4 = off / 4 + ((r >> 4) * KT + (k >> 4)) * 64u + ((k & 15u) >> 2) * 16u + (r & 15u);
        if (D.inference_h) D.inference_h[4u * e4 + (k & 3u)] = (_Float16)inf;
        if (D.params_h) D.params_h[4u * e4 + (k & 3u)] = (_Float16)nw;
        // fragment_mlp_hb_kernel: entry w_off / 4 + (kt RT + rt) 64 + lane holds W[16 rt + 4 g .. + 3][16 kt + i]
        if (D.params_hb) {
            const uint32_t RT = n_o / 16, b4 = off / 4 + ((k >> 4) * RT + (r >> 4)) * 64u + ((r & 15u) >> 2) * 16u + (k & 15u);
            D.params_hb[4u * b4 + (r & 3u)] = (_Float16)nw;
        }
    }
}

}  // namespace wost

#include "wost_net_half.h"

using namespace wost;

struct wost_net {
    int device = 0;
    wost_net_config cfg{};
    NetLayout L{};
    uint32_t n_params = 0;
    float *params = nullptr, *inference = nullptr, *m1 = nullptr, *m2 = nullptr, *ema_raw = nullptr;
    fx_t *grad = nullptr;   // fixed-point gradient sums of the last training step
    bool grad_owned = true; // false: the caller's buffer (wost_net_set_gradient_buffer)
    float *params_t = nullptr, *inference_t = nullptr;   // transposed MLP matrices (scalar forward pass)
    float *params_f = nullptr, *inference_f = nullptr;   // MFMA A-fragment order (MFMA forward pass)
    float *params_fb = nullptr;                          // MFMA fragments of the transposed matrices (backward pass)
    bool use_mfma = false;
    int precision = 32;              // 32 = fp32 everywhere (bit-exact mode, default), 16 = half-precision inference
    uint2 *inference_h = nullptr;    // f16 MFMA fragments of the inference (EMA) weights, precision 16 only
    int train_precision = 32;        // 16 = forward / backward / weight gradients of a training step on f16 MFMAs (wost_net_half.h)
    uint2 *params_h = nullptr, *params_hb = nullptr;   // f16 fragments of the training weights and of their transposes
    float *train_partial = nullptr;                    // per-block sums of the matrix gradients (net_train_h_kernel), 256 rows
    bool fused_backward = true;      // net_backward_wgrad_kernel (WOST_NET_FUSED=0: backward and weight gradients apart)
    int step = 0;
    uint64_t n_launches = 0;           // kernels and fills issued by the *_dev entry points (wost_guided_stats.kernel_launches)
    uint32_t *param_steps = nullptr;   // Adam steps taken by each parameter (tiny-cuda-nn adam_step)
    float *lr_table = nullptr;         // debiased learning rate of step s at [s], s = 1 .. lr_cap
    int lr_cap = 0;
    float grad_div = 1.0f;             // shared-network mode: ranks whose gradients are summed
    // scratch (grown on demand)
    float *d_xy = nullptr, *d_out = nullptr, *d_dl = nullptr, *d_acts = nullptr, *d_deltas = nullptr, *d_denc = nullptr;
    unsigned long long *d_mask = nullptr;   // MFMA path: signs of the hidden activations, 4 words per point
    uint32_t *d_bin_order = nullptr, *d_bin_tab = nullptr;   // three inputs: the points of a training step by spatial box (grid_bin3_*)
    size_t cap_points = 0;
};

#define NET_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return set_error(WOST_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// entries of the half-precision image of one parameter set: MFMA fragments of the matrices, then the grid
static size_t half_image_entries(const NetLayout &L) { return (size_t)L.n_mlp / 4 + L.level_off[L.n_levels]; }

static int launch_forward_h(wost_net *h, const float *p, const uint2 *image, const float *xy_dev, int n, const uint32_t *n_dev, float *out_dev,
                            size_t ldp, size_t ldf, uint2 *enc_out, hipStream_t stream)
{
    const NetLayout &L = h->L;
    const size_t lds = (L.dims == 2 ? half_image_entries(L) : (size_t)L.n_mlp / 4) * sizeof(uint2);
    const int n_tiles = (n + 16 * kHalfSub - 1) / (16 * kHalfSub);
    // one block per CU (two inputs: the image takes 150 KB of LDS), walking over the tiles
    const unsigned grid = (unsigned)std::max(1, std::min((n_tiles + kHalfFwdThreads / 64 - 1) / (kHalfFwdThreads / 64), 256));
    auto kfn = L.dims == 2 ? net_forward_h_kernel<2> : net_forward_h_kernel<3>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(kHalfFwdThreads), lds, stream, L, p, image, xy_dev, n, n_dev, out_dev, ldp, ldf, enc_out);
    ++h->n_launches;
    NET_TRY(hipGetLastError());
    return WOST_OK;
}

static void refresh_half(wost_net *h, hipStream_t stream)
{
    const unsigned gh = (unsigned)((half_image_entries(h->L) + 255) / 256);
    if (h->precision == 16 && h->inference_h) hipLaunchKernelGGL(fragment_mlp_h_kernel, dim3(gh), dim3(256), 0, stream, h->L, h->inference, h->inference_h);
    if (h->train_precision == 16 && h->params_h) {
        hipLaunchKernelGGL(fragment_mlp_h_kernel, dim3(gh), dim3(256), 0, stream, h->L, h->params, h->params_h);
        hipLaunchKernelGGL(fragment_mlp_hb_kernel, dim3(gh), dim3(256), 0, stream, h->L, h->params, h->params_hb);
    }
}

static int refresh_transposed(wost_net *h, hipStream_t stream)
{
    const unsigned g = (h->L.n_mlp + 255) / 256;
    hipLaunchKernelGGL(transpose_mlp_kernel, dim3(g), dim3(256), 0, stream, h->L, h->params, h->params_t);
    hipLaunchKernelGGL(transpose_mlp_kernel, dim3(g), dim3(256), 0, stream, h->L, h->inference, h->inference_t);
    if (h->use_mfma) {
        hipLaunchKernelGGL(fragment_mlp_kernel, dim3(g), dim3(256), 0, stream, h->L, h->params, h->params_f);
        hipLaunchKernelGGL(fragment_mlp_kernel, dim3(g), dim3(256), 0, stream, h->L, h->inference, h->inference_f);
        hipLaunchKernelGGL(fragment_mlp_t_kernel, dim3(g), dim3(256), 0, stream, h->L, h->params, h->params_fb);
    }
    refresh_half(h, stream);
    NET_TRY(hipGetLastError());
    return WOST_OK;
}

// forward pass on device pointers: MFMA kernel for the reference's network shape, the scalar
// kernel otherwise (or when WOST_NET_SCALAR=1 asks for the comparison path)
static int launch_forward(wost_net *h, bool use_inference_params, const float *xy_dev, int n, const uint32_t *n_dev,
                          float *out_dev, float *acts_dev, hipStream_t stream, size_t feature_stride = 0)
{
    if (n <= 0) return WOST_OK;
    const NetLayout &L = h->L;
    // outputs: point-major rows of n_out floats, or (feature_stride > 0) one array per output
    const size_t ldp = feature_stride ? 1 : (size_t)L.n_out, ldf = feature_stride ? feature_stride : 1;
    const float *p = use_inference_params ? h->inference : h->params;
    if (h->precision == 16 && use_inference_params && !acts_dev && h->inference_h) {
        // half-precision inference (the reference's network precision)
        return launch_forward_h(h, p, h->inference_h, xy_dev, n, n_dev, out_dev, ldp, ldf, nullptr, stream);
    }
    if (h->use_mfma) {
        const float *f = use_inference_params ? h->inference_f : h->params_f;
        // inference: one block of 16 waves per CU shares one copy of the weight fragments (53 KB,
        // 120 registers); with the activations kept the kernel needs 168 registers: two blocks of 4 waves
        const int nt = acts_dev ? kFwdTrainThreads : kFwdThreads;
        const size_t lds = ((size_t)L.n_mlp + (nt / 64) * kMfmaSub * 16 * (32 + kStagePad)) * sizeof(float);
        const int n_tiles = (n + 16 * kMfmaSub - 1) / (16 * kMfmaSub);
        const unsigned grid = (unsigned)std::min((n_tiles + nt / 64 - 1) / (nt / 64), nt >= 512 ? 256 : 512);
        unsigned long long *mask = acts_dev ? h->d_mask : nullptr;
#define WOST_FWD(SAVE, NF, NT)                                                                                          \
    do {                                                                                                                \
        auto kfn = net_forward_mfma_kernel<32, 64, 3, 48, SAVE, NF, NT>;                                                \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(NT), lds, stream, L, p, f, xy_dev, n, n_dev, out_dev, acts_dev, mask, ldp, ldf); \
    } while (0)
        if (L.n_features == 4) {
            if (acts_dev) WOST_FWD(true, 4, kFwdTrainThreads); else WOST_FWD(false, 4, kFwdThreads);
        } else {
            if (acts_dev) WOST_FWD(true, 0, kFwdTrainThreads); else WOST_FWD(false, 0, kFwdThreads);
        }
#undef WOST_FWD
    } else {
        const float *t = use_inference_params ? h->inference_t : h->params_t;
        const size_t lds = 2 * 64 * kNetBlock * sizeof(float);
        hipLaunchKernelGGL(net_forward_kernel, dim3((n + kNetBlock - 1) / kNetBlock), dim3(kNetBlock), lds, stream, L, p, t, xy_dev,
                           n, n_dev, out_dev, acts_dev, ldp, ldf);
    }
    ++h->n_launches;
    NET_TRY(hipGetLastError());
    return WOST_OK;
}

static int ensure_points(wost_net *h, size_t n)
{
    if (n <= h->cap_points) return WOST_OK;
    for (float **p : {&h->d_xy, &h->d_out, &h->d_dl, &h->d_acts, &h->d_deltas, &h->d_denc})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    h->cap_points = 0;
    const NetLayout &L = h->L;
    NET_TRY(hipMalloc((void **)&h->d_xy, n * 3 * sizeof(float)));
    NET_TRY(hipMalloc((void **)&h->d_out, n * L.n_out * sizeof(float)));
    NET_TRY(hipMalloc((void **)&h->d_dl, n * L.n_out * sizeof(float)));
    const size_t n64 = (n + 63) / 64 * 64;     // the MFMA training layout works in whole 64-point spans
    NET_TRY(hipMalloc((void **)&h->d_acts, n64 * (size_t)(L.enc + L.n_hidden * L.n_neurons) * sizeof(float)));
    NET_TRY(hipMalloc((void **)&h->d_deltas, n64 * (size_t)(L.n_out_padded + L.n_hidden * L.n_neurons) * sizeof(float)));
    NET_TRY(hipMalloc((void **)&h->d_denc, n * (size_t)L.enc * sizeof(float)));
    if (h->d_mask) { (void)hipFree(h->d_mask); h->d_mask = nullptr; }
    NET_TRY(hipMalloc((void **)&h->d_mask, n * 4 * sizeof(unsigned long long)));
    if (h->d_bin_order) { (void)hipFree(h->d_bin_order); h->d_bin_order = nullptr; }
    if (L.dims == 3) NET_TRY(hipMalloc((void **)&h->d_bin_order, n * sizeof(uint32_t)));
    h->cap_points = n;
    return WOST_OK;
}

// WOST_NET_CHECK3=1 (developer self check, EXPERIMENTS 20): the half-precision training kernels of every Adam step launched three times
// on the same inputs, the three results compared word by word on the device; the counts go to stderr when the network is destroyed
__global__ void check3_kernel(const uint32_t *a, const uint32_t *b, const uint32_t *c, size_t n, unsigned long long *out, uint32_t *log)
{
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = a[i], y = b[i], z = c[i];
    if (x == y && y == z) return;
    const unsigned long long k = atomicAdd(&out[0], 1ull);
    atomicAdd(&out[1 + (y == z ? 0 : x == z ? 1 : x == y ? 2 : 3)], 1ull);
    if (log && k < 65536ull) log[k] = (uint32_t)i;
}
struct Check3 {
    bool on = false, asked = false;
    unsigned long long *dev = nullptr;       // [kernel 0 forward / 1 train][words differing, odd launch 0 1 2, all differ] + launches with a difference
    void *scratch[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes[4] = {0, 0, 0, 0};
    unsigned long long steps = 0;
    uint32_t *log = nullptr;                 // word indices of the forward kernel's first 65 536 differing words
    int n_out = 33;
};
static Check3 g_check3;

// ---- what runs in FRONT of the three launches of the self check (WOST_NET_CHECK3_PRE, EXPERIMENTS 20 / 26) ----------------------
// The first-tile deviation of net_forward_h_kernel appears when the kernel follows a different kernel.  Two readings: a transient of
// the matrix pipe at the start of matrix work, or state the previous launch left behind (LDS is never cleared between launches, the
// instruction cache holds the previous kernel).  These kernels set up one or the other in front of chosen launches:
//   "lds"      every LDS byte of every CU filled with half-precision NaNs in front of ALL three launches
//   "lds23"    ... in front of launches 2 and 3 only (which never deviate as things stand)
//   "burn1"    a heavy matrix kernel of another kind in front of launch 1 (a warm matrix pipe, foreign LDS / instruction cache)
//   "icache23" the instruction caches invalidated in front of launches 2 and 3
__global__ __launch_bounds__(1024) void check3_lds_fill_kernel(uint32_t pattern, uint32_t n_words, uint32_t *sink)
{
    extern __shared__ uint32_t s_fill[];
    for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x) s_fill[i] = pattern;
    __syncthreads();
    if (s_fill[(threadIdx.x * 977u) % n_words] != pattern) atomicAdd(sink, 1u);
}
__global__ void check3_icache_kernel()
{
    asm volatile("s_icache_inv\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}
__global__ __launch_bounds__(1024) void check3_burn_kernel(int iters, float *sink)
{
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 16; ++j) acc[k][j] = 0.0f;
    const h4 a = {(_Float16)(threadIdx.x & 7), (_Float16)1.0f, (_Float16)0.5f, (_Float16)0.25f}, b = {(_Float16)1.0f, (_Float16)(threadIdx.x & 3), (_Float16)2.0f, (_Float16)0.125f};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, acc[k], 0, 0, 0);
    float t = 0.0f;
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 16; ++j) t += acc[k][j];
    if (t == 12345.678f) sink[0] = t;
}
static uint32_t check3_pre_mask()
{
    // bit k (0..2): lds in front of launch k; bit 3: burn in front of launch 0; bits 4, 5: icache in front of launches 1 and 2
    static int mask = -1;
    if (mask < 0) {
        mask = 0;
        const char *e = std::getenv("WOST_NET_CHECK3_PRE");
        const std::string v = e ? e : "";
        if (v.find("lds23") != std::string::npos) mask |= 6;
        else if (v.find("lds") != std::string::npos) mask |= 7;
        if (v.find("burn1") != std::string::npos) mask |= 8;
        if (v.find("icache23") != std::string::npos) mask |= 48;
    }
    return (uint32_t)mask;
}
static void check3_pre(int launch, hipStream_t stream)
{
    const uint32_t m = check3_pre_mask();
    if (!m || !g_check3.dev) return;
    uint32_t *sink = reinterpret_cast<uint32_t *>(g_check3.dev + 15);
    if (m & (1u << launch)) {
        const uint32_t bytes = 156u * 1024u;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(check3_lds_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        hipLaunchKernelGGL(check3_lds_fill_kernel, dim3(512), dim3(1024), bytes, stream, 0x7e007e00u, bytes / 4u, sink);
    }
    if (launch == 0 && (m & 8u)) hipLaunchKernelGGL(check3_burn_kernel, dim3(512), dim3(1024), 0, stream, 400, reinterpret_cast<float *>(sink));
    if ((launch == 1 && (m & 16u)) || (launch == 2 && (m & 32u))) hipLaunchKernelGGL(check3_icache_kernel, dim3(4096), dim3(64), 0, stream);
}
static bool check3_on()
{
    if (!g_check3.asked) {
        g_check3.asked = true;
        const char *e = std::getenv("WOST_NET_CHECK3");
        g_check3.on = e && std::atoi(e) != 0;
        if (g_check3.on && (hipMalloc((void **)&g_check3.log, 65536 * 4) != hipSuccess || hipMalloc((void **)&g_check3.dev, 16 * sizeof(unsigned long long)) != hipSuccess ||
                            hipMemset(g_check3.dev, 0, 16 * sizeof(unsigned long long)) != hipSuccess))
            g_check3.on = false;
    }
    return g_check3.on;
}
static void *check3_scratch(int k, size_t bytes)
{
    if (g_check3.scratch_bytes[k] < bytes) {
        if (g_check3.scratch[k]) (void)hipFree(g_check3.scratch[k]);
        g_check3.scratch[k] = nullptr;
        if (hipMalloc(&g_check3.scratch[k], bytes) != hipSuccess) return nullptr;
        g_check3.scratch_bytes[k] = bytes;
    }
    return g_check3.scratch[k];
}
static void check3_compare(int kernel, const void *a, const void *b, const void *c, size_t bytes, hipStream_t stream)
{
    const size_t words = bytes / 4;
    hipLaunchKernelGGL(check3_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const uint32_t *>(a),
                       reinterpret_cast<const uint32_t *>(b), reinterpret_cast<const uint32_t *>(c), words, g_check3.dev + 8 * kernel, kernel == 0 ? g_check3.log : nullptr);
}
static void check3_report()
{
    if (!g_check3.on || !g_check3.dev) return;
    unsigned long long v[16];
    if (hipMemcpy(v, g_check3.dev, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return;
#ifdef WOST_H_NO_REDO
    std::fprintf(stderr, "CHECK3 build: net_forward_h_kernel WITHOUT the recomputed first tile (WOST_H_NO_REDO)\n");
#else
    std::fprintf(stderr, "CHECK3 build: net_forward_h_kernel recomputes a wave's first tile\n");
#endif
    std::fprintf(stderr, "CHECK3 after %llu training steps: forward words differing %llu (odd launch 0/1/2/all: %llu %llu %llu %llu); train kernel words differing %llu (%llu %llu %llu %llu)\n",
                 g_check3.steps, v[0], v[1], v[2], v[3], v[4], v[8], v[9], v[10], v[11], v[12]);
    // where in the launch the forward kernel's differing units lie: a wave takes the tiles blockIdx * waves + wave + k * (256 * waves), k = 0, 1, ...
    const size_t n_log = (size_t)std::min<unsigned long long>(v[0], 65536ull);
    if (n_log) {
        std::vector<uint32_t> idx(n_log);
        if (hipMemcpy(idx.data(), g_check3.log, n_log * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
        const int waves = kHalfFwdThreads / 64;
        std::vector<char> seen;
        size_t hist[16] = {0}, units = 0;
        for (uint32_t w : idx) {
            const size_t unit = (size_t)w / (size_t)g_check3.n_out / 16;
            if (seen.size() <= unit) seen.resize(unit + 1, 0);
            if (seen[unit]) continue;
            seen[unit] = 1;      // (a unit that fails in two different steps is counted once: rare)
            ++units;
            hist[std::min<size_t>(15, unit / 2 / (size_t)(256 * waves))]++;
        }
        std::fprintf(stderr, "CHECK3 forward: %zu distinct units among the first %zu differing words; by the wave's iteration k: ", units, n_log);
        for (int k = 0; k < 16; ++k) std::fprintf(stderr, "%zu ", hist[k]);
        std::fprintf(stderr, "\n");
    }
}

static void net_free(wost_net *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    check3_report();
    for (float *p : {h->params, h->inference, h->params_t, h->inference_t, h->params_f, h->inference_f, h->params_fb, h->m1, h->m2, h->ema_raw, h->d_xy, h->d_out, h->d_dl, h->d_acts, h->d_deltas, h->d_denc})
        if (p) (void)hipFree(p);
    if (h->grad && h->grad_owned) (void)hipFree(h->grad);
    if (h->d_mask) (void)hipFree(h->d_mask);
    if (h->d_bin_order) (void)hipFree(h->d_bin_order);
    if (h->d_bin_tab) (void)hipFree(h->d_bin_tab);
    if (h->param_steps) (void)hipFree(h->param_steps);
    if (h->lr_table) (void)hipFree(h->lr_table);
    if (h->inference_h) (void)hipFree(h->inference_h);
    if (h->params_h) (void)hipFree(h->params_h);
    if (h->params_hb) (void)hipFree(h->params_hb);
    if (h->train_partial) (void)hipFree(h->train_partial);
    delete h;
}

namespace wost {

// ---- device-pointer entry points used by the guided integrator (wost_guided.hip) -----------
int net_inference_dev(wost_net *h, const float *xy_dev, const uint32_t *count_dev, int max_n, float *out_dev,
                      bool use_inference_params, hipStream_t stream, size_t feature_stride)
{
    return launch_forward(h, use_inference_params, xy_dev, max_n, count_dev, out_dev, nullptr, stream, feature_stride);
}

// forward with the training parameters, keeping activations; *out_dev = raw outputs
// (n x n_output), *dl_dev = where the caller writes dL/dout before net_backward_update_dev
int net_forward_train_dev(wost_net *h, const float *xy_dev, int n, hipStream_t stream, float **out_dev, float **dl_dev)
{
    int rc = ensure_points(h, (size_t)n);
    if (rc != WOST_OK) return rc;
    if (h->train_precision == 16) {
        // the inference kernel on the training weights; the f16 encoding of every point (64 bytes) is kept
        if (check3_on() && h->L.dims == 2 && std::atoi(std::getenv("WOST_NET_CHECK3")) == 2) {
            // variant 2: a discarded launch in front (is it the launch that follows OTHER kernels that differs, whatever it computes?)
            const size_t ob = (size_t)n * h->L.n_out * 4, eb = (size_t)((n + 31) / 32 * 2) * 2 * 64 * 8;
            float *o3 = (float *)check3_scratch(3, ob);
            uint2 *e1 = (uint2 *)check3_scratch(2, eb);
            if (o3 && e1) (void)launch_forward_h(h, h->params, h->params_h, xy_dev, n, nullptr, o3, (size_t)h->L.n_out, 1, e1, stream);
        }
        if (check3_on()) check3_pre(0, stream);
        rc = launch_forward_h(h, h->params, h->params_h, xy_dev, n, nullptr, h->d_out, (size_t)h->L.n_out, 1, reinterpret_cast<uint2 *>(h->d_acts), stream);
        if (rc != WOST_OK) return rc;
        if (check3_on()) {
            g_check3.n_out = h->L.n_out;
            const size_t ob = (size_t)n * h->L.n_out * 4, eb = (size_t)((n + 31) / 32 * 2) * 2 * 64 * 8;
            float *o1 = (float *)check3_scratch(0, ob), *o2 = (float *)check3_scratch(1, ob);
            uint2 *e1 = (uint2 *)check3_scratch(2, eb);
            if (o1 && o2 && e1) {
                check3_pre(1, stream);
                (void)launch_forward_h(h, h->params, h->params_h, xy_dev, n, nullptr, o1, (size_t)h->L.n_out, 1, e1, stream);
                check3_pre(2, stream);
                (void)launch_forward_h(h, h->params, h->params_h, xy_dev, n, nullptr, o2, (size_t)h->L.n_out, 1, e1, stream);
                check3_compare(0, h->d_out, o1, o2, ob, stream);
                ++g_check3.steps;
            }
        }
    } else {
        rc = launch_forward(h, false, xy_dev, n, nullptr, h->d_out, h->d_acts, stream);
        if (rc != WOST_OK) return rc;
    }
    *out_dev = h->d_out;
    *dl_dev = h->d_dl;
    return WOST_OK;
}

int net_apply_update_dev(wost_net *h, float loss_scale, hipStream_t stream);

int net_backward_update_dev(wost_net *h, const float *xy_dev, int n, float loss_scale, int apply_update, hipStream_t stream)
{
    const NetLayout &L = h->L;
    NET_TRY(hipMemsetAsync(h->grad, 0, (size_t)h->n_params * sizeof(fx_t), stream));
    ++h->n_launches;
    const bool half = h->train_precision == 16;
    const bool fused = h->fused_backward || half;
    if (half) {
        // recompute the hidden layers from the stored encoding, backward pass, all weight gradients in registers
        const size_t lds = (size_t)L.n_mlp / 4 * sizeof(uint2) * 2;
        const int n_units = (n + 15) / 16;
        const unsigned gridb = (unsigned)std::min((n_units + kHalfThreads / 64 - 1) / (kHalfThreads / 64), 256);
        int k = 0;
        while (k < 10 && (n >> (k + 10)) > 0) ++k;          // 2^k ~ n / 512, between 1 and 1024
        hipLaunchKernelGGL(net_train_h_kernel, dim3(gridb), dim3(kHalfThreads), lds, stream, L, h->params_h, h->params_hb,
                           reinterpret_cast<const uint2 *>(h->d_acts), h->d_dl, n, (float)(1 << k), h->d_denc, h->train_partial);
        if (check3_on() && L.dims == 2) {
            const size_t db = (size_t)n * L.enc * 4, pb = (size_t)gridb * L.n_mlp * 4;
            float *d1 = (float *)check3_scratch(0, db), *d2 = (float *)check3_scratch(1, db), *p1 = (float *)check3_scratch(2, pb), *p2 = (float *)check3_scratch(3, pb);
            if (d1 && d2 && p1 && p2) {
                for (int rep = 0; rep < 2; ++rep)
                    hipLaunchKernelGGL(net_train_h_kernel, dim3(gridb), dim3(kHalfThreads), lds, stream, L, h->params_h, h->params_hb,
                                       reinterpret_cast<const uint2 *>(h->d_acts), h->d_dl, n, (float)(1 << k), rep ? d2 : d1, rep ? p2 : p1);
                check3_compare(1, h->d_denc, d1, d2, db, stream);
                check3_compare(1, h->train_partial, p1, p2, pb, stream);
            }
        }
        hipLaunchKernelGGL(net_train_h_reduce_kernel, dim3((L.n_mlp + 255) / 256, (gridb + 15) / 16), dim3(256), 0, stream, L, h->train_partial, (int)gridb,
                           h->grad);
        h->n_launches += 2;
    } else if (h->use_mfma && fused) {
        // backward pass and weight gradients of a 1024-point chunk in one block (deltas stay on chip)
        const size_t lds = ((size_t)L.n_mlp + 4 * 64 * kTileStride + 3 * 64 * 64) * sizeof(float);
        auto kfn = net_backward_wgrad_kernel<32, 64, 3, 48>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kfn, dim3((unsigned)((n + 1023) / 1024)), dim3(256), lds, stream, L, h->params_fb, h->d_dl, h->d_mask, n,
                           h->d_acts, h->d_denc, h->grad);
        ++h->n_launches;
    } else if (h->use_mfma) {
        const size_t lds = (size_t)L.n_mlp * sizeof(float);
        const int n_tiles = (n + 63) / 64 * 4 / kMfmaSub;
        const unsigned gridb = (unsigned)std::min((n_tiles + kBwdThreads / 64 - 1) / (kBwdThreads / 64), 256);
        hipLaunchKernelGGL((net_backward_mfma_kernel<32, 64, 3, 48, kBwdThreads>), dim3(gridb), dim3(kBwdThreads), lds, stream, L,
                           h->params_fb, h->d_dl, h->d_mask, n, h->d_deltas, h->d_denc);
        ++h->n_launches;
    } else {
        const size_t lds = 2 * 64 * kNetBlock * sizeof(float);
        const unsigned gridp = (unsigned)((n + kNetBlock - 1) / kNetBlock);
        hipLaunchKernelGGL(net_backward_kernel, dim3(gridp), dim3(kNetBlock), lds, stream, L, h->params, xy_dev, h->d_dl, h->d_acts,
                           n, h->d_deltas, h->d_denc);
        ++h->n_launches;
    }
    NET_TRY(hipGetLastError());
    {
        // group consecutive levels so that each group's 8-byte accumulators fit into LDS: up to
        // 72 KB per group (two blocks per CU), a single larger level alone up to 150 KB (one block
        // per CU), and a level larger than that in feature slices
        const size_t small = 72 * 1024, big = 150 * 1024;
        auto level_bytes = [&](int l, int nq) { return (size_t)(L.level_off[l + 1] - L.level_off[l]) * nq * sizeof(fx_t); };
        auto launch = [&](int lv0, int lv1, int q0, int q1, size_t bytes, int use_lds) {
            const int gchunk = bytes > small ? 4096 : 2048;
            if (bytes > 48 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(grid_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)big);
            const size_t ld_point = half ? (size_t)L.n_features : (size_t)L.enc, ld_level = half ? (size_t)n * L.n_features : (size_t)L.n_features;
            hipLaunchKernelGGL(grid_grad_kernel, dim3((unsigned)((n + gchunk - 1) / gchunk)), dim3(kGridGradBlock), bytes, stream, L, xy_dev,
                               h->d_denc, ld_point, ld_level, n, gchunk, lv0, lv1, q0, q1, use_lds, h->grad);
            ++h->n_launches;
        };
        // one launch for all groups when every group fits the budget of two blocks per CU (two-input networks: the three-input
        // grids are far larger than LDS); WOST_GRID_GRAD_PLAN=0: the launches per group below
        GridGradPlan plan{};
        size_t plan_bytes = 0;
        bool planned = L.dims == 2 && !(std::getenv("WOST_GRID_GRAD_PLAN") && std::atoi(std::getenv("WOST_GRID_GRAD_PLAN")) == 0);
        int coarse_copies = 4;       // copies of the accumulators of levels below 1024 cells (WOST_GRID_GRAD_COPIES)
        if (const char *w = std::getenv("WOST_GRID_GRAD_COPIES")) coarse_copies = std::max(1, std::min(8, std::atoi(w)));
        while (coarse_copies & (coarse_copies - 1)) --coarse_copies;
        for (int lv = 0; planned && lv < L.n_levels;) {
            int end = lv;
            size_t bytes = 0;
            const bool coarse = L.level_off[lv + 1] - L.level_off[lv] < 1024u;
            const size_t budget = coarse ? small / (size_t)coarse_copies : small;
            while (end < L.n_levels && (L.level_off[end + 1] - L.level_off[end] < 1024u) == coarse && bytes + level_bytes(end, L.n_features) <= budget)
                bytes += level_bytes(end++, L.n_features);
            if (end == lv && coarse && level_bytes(lv, L.n_features) <= small) bytes = level_bytes(end++, L.n_features);      // alone, fewer copies
            if (end > lv) {
                if (plan.n_groups >= 16) { planned = false; break; }
                const int g = plan.n_groups++;
                plan.lv0[g] = lv; plan.lv1[g] = end; plan.q0[g] = 0; plan.q1[g] = L.n_features;
                int copies = 1;
                while (copies < 8 && (size_t)(2 * copies) * bytes <= small) copies *= 2;
                plan.replicas[g] = copies;
                plan_bytes = std::max(plan_bytes, bytes * (size_t)copies);
                lv = end;
                continue;
            }
            int slices = 1;
            while (slices < L.n_features && level_bytes(lv, (L.n_features + slices - 1) / slices) > small) ++slices;
            const int per = (L.n_features + slices - 1) / slices;
            if (level_bytes(lv, per) > small) { planned = false; break; }
            for (int q0 = 0; q0 < L.n_features; q0 += per) {
                if (plan.n_groups >= 16) { planned = false; break; }
                const int g = plan.n_groups++;
                plan.lv0[g] = lv; plan.lv1[g] = lv + 1; plan.q0[g] = q0; plan.q1[g] = std::min(L.n_features, q0 + per);
                plan.replicas[g] = 1;
                plan_bytes = std::max(plan_bytes, level_bytes(lv, plan.q1[g] - plan.q0[g]));
            }
            ++lv;
        }
        if (planned && plan.n_groups > 0) {
            // equal work per block: chunk[g] x (levels x features of g) about the same for every group, ~ 448 blocks in all (two per CU)
            double total = 0.0;
            for (int g = 0; g < plan.n_groups; ++g) total += (double)(plan.lv1[g] - plan.lv0[g]) * (plan.q1[g] - plan.q0[g]);
            const double per_block = total * (double)n / 448.0;
            int blocks = 0;
            for (int g = 0; g < plan.n_groups; ++g) {
                const double w = (double)(plan.lv1[g] - plan.lv0[g]) * (plan.q1[g] - plan.q0[g]);
                long long chunk_g = (long long)(per_block / w);
                chunk_g = std::max<long long>(1024, (chunk_g + 63) / 64 * 64);
                chunk_g = std::min<long long>(chunk_g, ((long long)n + 63) / 64 * 64);
                plan.chunk[g] = (int32_t)chunk_g;
                plan.first_block[g] = blocks;
                blocks += (int)(((long long)n + chunk_g - 1) / chunk_g);
            }
            plan.first_block[plan.n_groups] = blocks;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(grid_grad_plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small);
            const size_t ld_point = half ? (size_t)L.n_features : (size_t)L.enc, ld_level = half ? (size_t)n * L.n_features : (size_t)L.n_features;
            hipLaunchKernelGGL(grid_grad_plan_kernel, dim3((unsigned)blocks), dim3(kGridGradBlock), plan_bytes, stream, L, xy_dev, h->d_denc, ld_point, ld_level, n,
                               plan, h->grad);
            ++h->n_launches;
        }
        int lv = planned && plan.n_groups > 0 ? L.n_levels : 0;
        // three inputs: every level in one launch through spatial boxes (WOST_GRID_GRAD_BINS=0: the launches per level group below)
        if (L.dims == 3 && !(std::getenv("WOST_GRID_GRAD_BINS") && std::atoi(std::getenv("WOST_GRID_GRAD_BINS")) == 0) && h->d_bin_order &&
            (size_t)n <= h->cap_points) {
            GridBin3Plan bp{};
            const size_t lds_cap = 152 * 1024;
            for (int bins : {8, 16}) {
                int total = 0;
                for (int l = 0; l < L.n_levels; ++l) {
                    bp.ext[l] = (int)std::floor(L.scale[l] / (float)bins) + 3;
                    bp.acc_off[l] = total;
                    total += bp.ext[l] * bp.ext[l] * bp.ext[l];
                }
                bp.acc_off[L.n_levels] = bp.n_acc = total;
                bp.bins = (size_t)total * L.n_features * sizeof(fx_t) <= lds_cap ? bins : 0;
                if (bp.bins) break;
            }
            if (bp.bins) {
                const int nb = bp.bins * bp.bins * bp.bins;
                bp.chunk = 2048;
                if (const char *w = std::getenv("WOST_GRID_GRAD_BIN_CHUNK")) bp.chunk = std::max(64, std::atoi(w));
                if (!h->d_bin_tab) NET_TRY(hipMalloc((void **)&h->d_bin_tab, (2 * 4096 + 1 + 2 * 4097) * sizeof(uint32_t)));
                NET_TRY(hipMemsetAsync(h->d_bin_tab, 0, (2 * (size_t)nb + 1) * sizeof(uint32_t), stream));
                const unsigned gc = (unsigned)((n + kBin3CountPoints - 1) / kBin3CountPoints);
                hipLaunchKernelGGL(grid_bin3_count_kernel, dim3(gc), dim3(1024), (size_t)std::max(nb, 2048) * sizeof(uint32_t), stream, xy_dev, n, bp.bins,
                                   bp.chunk, h->d_bin_tab);
                hipLaunchKernelGGL(grid_bin3_scatter_kernel, dim3(gc), dim3(1024), 2 * (size_t)nb * sizeof(uint32_t), stream, xy_dev, n, bp.bins, h->d_bin_tab,
                                   h->d_bin_order);
                const size_t acc_bytes = (size_t)bp.n_acc * L.n_features * sizeof(fx_t);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(grid_bin3_accumulate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
                const size_t ld_point = half ? (size_t)L.n_features : (size_t)L.enc, ld_level = half ? (size_t)n * L.n_features : (size_t)L.n_features;
                hipLaunchKernelGGL(grid_bin3_accumulate_kernel, dim3((unsigned)(n / bp.chunk + nb + 1)), dim3(kGridGradBlock), acc_bytes, stream, L, xy_dev, h->d_denc,
                                   ld_point, ld_level, bp, h->d_bin_tab, h->d_bin_order, h->grad);
                h->n_launches += 4;
                lv = L.n_levels;
            }
        }
        while (lv < L.n_levels) {
            int end = lv;
            size_t bytes = 0;
            while (end < L.n_levels && bytes + level_bytes(end, L.n_features) <= small) bytes += level_bytes(end++, L.n_features);
            if (end > lv) {
                launch(lv, end, 0, L.n_features, bytes, 1);
                lv = end;
                continue;
            }
            // levels of which not even ONE feature fits a block's LDS (three inputs: 32^3 cells and up) add straight to global memory: all
            // of them and all their features in ONE launch -- a launch per level and feature slice (round 4: sixteen of a step's twenty-odd
            // launches) walks all n points again each time for the same atomics
            if (level_bytes(lv, 1) > big) {
                int end_g = lv;
                while (end_g < L.n_levels && level_bytes(end_g, 1) > big) ++end_g;
                launch(lv, end_g, 0, L.n_features, 0, 0);
                lv = end_g;
                continue;
            }
            // this level alone exceeds the group budget: whole if it fits one block per CU, else feature slices
            int slices = 1;
            while (slices < L.n_features && level_bytes(lv, (L.n_features + slices - 1) / slices) > big) ++slices;
            const int per = (L.n_features + slices - 1) / slices;
            for (int q0 = 0; q0 < L.n_features; q0 += per) {
                const int q1 = std::min(L.n_features, q0 + per);
                const size_t sb = level_bytes(lv, q1 - q0);
                launch(lv, lv + 1, q0, q1, sb <= big ? sb : 0, sb <= big ? 1 : 0);
            }
            ++lv;
        }
        NET_TRY(hipGetLastError());
    }
    const int astride = L.enc + L.n_hidden * L.n_neurons, dstride = L.n_out_padded + L.n_hidden * L.n_neurons;
    const int chunk = 1024;
    const unsigned gridc = (unsigned)((n + chunk - 1) / chunk);
    const size_t lds_w = 2 * 32 * 64 * sizeof(float);
    for (int layer = 0; layer <= L.n_hidden && !(h->use_mfma && fused) && !half; ++layer) {
        const int n_i = layer == 0 ? L.enc : L.n_neurons, n_o = layer == L.n_hidden ? L.n_out_padded : L.n_neurons;
        const int doff = layer == L.n_hidden ? 0 : L.n_out_padded + layer * L.n_neurons;   // delta of this layer's output
        const int ioff = layer == 0 ? 0 : L.enc + (layer - 1) * L.n_neurons;               // this layer's input
        fx_t *gW = h->grad + L.w_off[layer];
        ++h->n_launches;
        if (h->use_mfma) {
#define WG(NO, NI) hipLaunchKernelGGL((weight_grad_mfma_kernel<NO, NI>), dim3(gridc), dim3(256), 0, stream, h->d_deltas, dstride, \
                                      doff, h->d_acts, astride, ioff, n, chunk, gW)
            if (layer == 0) WG(64, 32);
            else if (layer == L.n_hidden) WG(48, 64);
            else WG(64, 64);
#undef WG
        } else {
            hipLaunchKernelGGL(weight_grad_kernel, dim3(gridc), dim3(256), lds_w, stream, h->d_deltas, dstride, doff, h->d_acts,
                               astride, ioff, n_o, n_i, n, chunk, gW);
        }
    }
    NET_TRY(hipGetLastError());
    if (apply_update) return net_apply_update_dev(h, loss_scale, stream);
    return WOST_OK;
}

// one Adam + EMA step on the accumulated gradient (kept apart from the backward pass so that a
// multi-GPU caller can sum the fixed-point gradients of all ranks in between)
int net_apply_update_dev(wost_net *h, float loss_scale, hipStream_t stream)
{
    h->step += 1;
    const wost_net_config &c = h->cfg;
    if (h->step > h->lr_cap) {
        // grow the table of debiased learning rates (same expression as the oracle, evaluated by the
        // host's libm on both sides)
        const int cap = std::max(4096, 2 * h->step);
        std::vector<float> tab((size_t)cap + 1, 0.0f);
        for (int s_ = 1; s_ <= cap; ++s_)
            tab[s_] = c.learning_rate * std::sqrt(1.0f - std::pow(c.beta2, (float)s_)) / (1.0f - std::pow(c.beta1, (float)s_));
        NET_TRY(hipStreamSynchronize(stream));
        if (h->lr_table) (void)hipFree(h->lr_table);
        h->lr_table = nullptr;
        NET_TRY(hipMalloc((void **)&h->lr_table, tab.size() * sizeof(float)));
        NET_TRY(hipMemcpy(h->lr_table, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
        h->lr_cap = cap;
    }
    const float debias = 1.0f / (1.0f - std::pow(c.ema_decay, (float)h->step));
    DerivedLayouts D{h->params_t, h->inference_t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (h->use_mfma) { D.params_f = h->params_f; D.inference_f = h->inference_f; D.params_fb = h->params_fb; }
    // the half-precision images are kept up to date by the same kernel (three launches less per step)
    if (h->precision == 16 && h->inference_h) D.inference_h = reinterpret_cast<_Float16 *>(h->inference_h);
    if (h->train_precision == 16 && h->params_h) {
        D.params_h = reinterpret_cast<_Float16 *>(h->params_h);
        D.params_hb = reinterpret_cast<_Float16 *>(h->params_hb);
    }
    hipLaunchKernelGGL(optimizer_kernel, dim3((h->n_params + 255) / 256), dim3(256), 0, stream, h->L, h->n_params, h->params, h->m1,
                       h->m2, h->ema_raw, h->inference, h->grad, h->lr_table, h->param_steps, c.beta1, c.beta2, c.epsilon,
                       c.l2_reg, c.ema_decay, debias, loss_scale, h->grad_div, D);
    ++h->n_launches;
    NET_TRY(hipGetLastError());
    return WOST_OK;
}

void *net_gradient_buffer(wost_net *h, uint64_t *count)
{
    if (count) *count = h->n_params;
    return h->grad;
}

int net_f32_view(wost_net *h, F32NetView *out)
{
    if (!h || !out) return set_error(WOST_ERR_INVALID, "null argument");
    if (!h->use_mfma || !h->inference_f || h->L.dims != 2) return WOST_ERR_UNSUPPORTED;      // a probe: the caller falls back, no error is recorded
    out->L = h->L;
    out->frag = h->inference_f;
    out->grid = h->inference + h->L.n_mlp;
    return WOST_OK;
}

int net_f32_view3(wost_net *h, F32NetView *out)
{
    if (!h || !out) return set_error(WOST_ERR_INVALID, "null argument");
    if (!h->use_mfma || !h->inference_f || h->L.dims != 3 || h->precision != 32) return WOST_ERR_UNSUPPORTED;      // a probe, as above
    out->L = h->L;
    out->frag = h->inference_f;
    out->grid = h->inference + h->L.n_mlp;
    return WOST_OK;
}

int net_half_view(wost_net *h, HalfNetView *out)
{
    if (!h || !out) return set_error(WOST_ERR_INVALID, "null argument");
    if (h->precision != 16 || !h->inference_h || h->L.dims != 2) return WOST_ERR_UNSUPPORTED;      // a probe, as above
    out->L = h->L;
    out->image = h->inference_h;
    return WOST_OK;
}

size_t net_snapshot_bytes(wost_net *h)
{
    if (!h) return 0;
    if (h->precision == 16 && h->inference_h) return half_image_entries(h->L) * sizeof(uint2);
    if (h->use_mfma && h->inference_f && h->L.dims == 2) return (size_t)h->n_params * sizeof(float);    // fragments, then the grid
    return 0;
}

int net_snapshot_dev(wost_net *h, void *dst, hipStream_t stream)
{
    if (!h || !dst) return set_error(WOST_ERR_INVALID, "null argument");
    if (h->precision == 16 && h->inference_h) {
        NET_TRY(hipMemcpyAsync(dst, h->inference_h, half_image_entries(h->L) * sizeof(uint2), hipMemcpyDeviceToDevice, stream));
    } else if (h->use_mfma && h->inference_f && h->L.dims == 2) {
        float *d = reinterpret_cast<float *>(dst);
        NET_TRY(hipMemcpyAsync(d, h->inference_f, (size_t)h->L.n_mlp * sizeof(float), hipMemcpyDeviceToDevice, stream));
        NET_TRY(hipMemcpyAsync(d + h->L.n_mlp, h->inference + h->L.n_mlp, (size_t)(h->n_params - h->L.n_mlp) * sizeof(float), hipMemcpyDeviceToDevice, stream));
    } else {
        return set_error(WOST_ERR_UNSUPPORTED, "this network has no image to copy");
    }
    h->n_launches += 1;
    return WOST_OK;
}

int net_snapshot_views(wost_net *h, const void *snap, bool *half, HalfNetView *hv, F32NetView *fv)
{
    if (!h || !snap || !half || !hv || !fv) return set_error(WOST_ERR_INVALID, "null argument");
    if (h->precision == 16 && h->inference_h) {
        *half = true;
        hv->L = h->L;
        hv->image = reinterpret_cast<const uint2 *>(snap);
        return WOST_OK;
    }
    if (h->use_mfma && h->inference_f && h->L.dims == 2) {
        *half = false;
        fv->L = h->L;
        fv->frag = reinterpret_cast<const float *>(snap);
        fv->grid = fv->frag + h->L.n_mlp;
        return WOST_OK;
    }
    return WOST_ERR_UNSUPPORTED;
}

int net_optimizer_steps(const wost_net *h) { return h->step; }
uint64_t net_launch_count(const wost_net *h) { return h->n_launches; }
void net_set_gradient_divisor(wost_net *h, float ranks) { h->grad_div = ranks > 0.0f ? ranks : 1.0f; }
int net_n_output(const wost_net *h) { return h->L.n_out; }

}  // namespace wost

extern "C" {

static int net_create_dims(int device, const wost_net_config *cfg, uint64_t seed, int dims, wost_net_handle *out)
{
    if (!cfg || !out) return set_error(WOST_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->n_levels < 1 || cfg->n_levels > kNetMaxLevels || cfg->n_features_per_level < 1 || cfg->n_features_per_level > 8 ||
        cfg->n_levels * cfg->n_features_per_level > 64 || (cfg->n_levels * cfg->n_features_per_level) % 8 != 0 ||
        cfg->n_neurons < 8 || cfg->n_neurons > 64 || cfg->n_neurons % 8 != 0 || cfg->n_hidden_layers < 1 ||
        cfg->n_hidden_layers >= kNetMaxLevels || cfg->n_output < 1 || cfg->n_output > 64 || cfg->base_resolution < 1 ||
        !(cfg->per_level_scale >= 1.0f))
        return set_error(WOST_ERR_UNSUPPORTED, "network shape outside this build (widths <= 64, multiples of 8)");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return set_error(WOST_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n_dev) return set_error(WOST_ERR_INVALID, "device index out of range");
    NET_TRY(hipSetDevice(device));
    wost_net *h = new (std::nothrow) wost_net();
    if (!h) return set_error(WOST_ERR_NOMEM, "out of host memory");
    h->device = device;
    h->cfg = *cfg;
    {
        // (offsets are 32-bit: a grid of three inputs has res^3 entries per level)
        double entries = 0.0;
        for (int i = 0; i < cfg->n_levels; ++i) {
            const double r = std::ceil(std::exp2((double)i * std::log2((double)cfg->per_level_scale)) * cfg->base_resolution - 1.0) + 1.0;
            entries += dims == 3 ? r * r * r : r * r;
        }
        if (entries * cfg->n_features_per_level > 1.0e9) {
            delete h;
            return set_error(WOST_ERR_UNSUPPORTED, "grid encoding too large");
        }
    }
    h->L = make_layout(*cfg, dims);
    h->n_params = h->L.n_mlp + h->L.n_grid;
    {
        // the MFMA forward kernel is instantiated for the reference's network shape
        const char *unfused = getenv("WOST_NET_FUSED");
        h->fused_backward = !(unfused && atoi(unfused) == 0);
        const char *scalar = getenv("WOST_NET_SCALAR");
        // (three inputs -- GuidedIntegrator<3> -- only with the reference's four features per level: f32_encode_level3)
        h->use_mfma = (dims == 2 || (dims == 3 && h->L.n_features == 4)) && h->L.enc == 32 && h->L.n_neurons == 64 && h->L.n_hidden == 3 &&
                      h->L.n_out_padded == 48 && h->L.n_features <= 8 && !(scalar && atoi(scalar) != 0);
    }
    // initialisation (tiny-cuda-nn defaults): MLP xavier uniform, grid uniform(-1e-4, 1e-4)
    std::vector<float> init(h->n_params);
    uint64_t state = 0, inc = (54u << 1u) | 1u;
    auto next_u32 = [&]() {
        const uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        const uint32_t x = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
        return (x >> rot) | (x << ((~rot + 1u) & 31));
    };
    next_u32(); state += seed; next_u32();
    auto uniform = [&]() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); };
    const NetLayout &L = h->L;
    for (int layer = 0; layer <= L.n_hidden; ++layer) {
        const int n_i = layer == 0 ? L.enc : L.n_neurons, n_o = layer == L.n_hidden ? L.n_out_padded : L.n_neurons;
        const float s = std::sqrt(6.0f / (float)(n_i + n_o));
        for (int e = 0; e < n_i * n_o; ++e) init[L.w_off[layer] + e] = (uniform() * 2.0f - 1.0f) * s;
    }
    for (uint32_t e = 0; e < L.n_grid; ++e) init[L.n_mlp + e] = (uniform() * 2.0f - 1.0f) * 1e-4f;
    const size_t bytes = (size_t)h->n_params * sizeof(float);
    hipError_t e = hipSuccess;
    for (float **p : {&h->params, &h->inference, &h->m1, &h->m2, &h->ema_raw, &h->params_t, &h->inference_t, &h->params_f, &h->inference_f, &h->params_fb})
        if (e == hipSuccess) e = hipMalloc((void **)p, bytes);
    if (e == hipSuccess) e = hipMemcpy(h->params, init.data(), bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->inference, init.data(), bytes, hipMemcpyHostToDevice);
    for (float *p : {h->m1, h->m2, h->ema_raw})
        if (e == hipSuccess) e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&h->grad, (size_t)h->n_params * sizeof(fx_t));
    if (e == hipSuccess) e = hipMemset(h->grad, 0, (size_t)h->n_params * sizeof(fx_t));
    if (e == hipSuccess) e = hipMalloc((void **)&h->param_steps, (size_t)h->n_params * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(h->param_steps, 0, (size_t)h->n_params * sizeof(uint32_t));
    if (e != hipSuccess) {
        net_free(h);
        return set_error(WOST_ERR_DEVICE, std::string("network allocation: ") + hipGetErrorString(e));
    }
    if (refresh_transposed(h, nullptr) != WOST_OK || hipDeviceSynchronize() != hipSuccess) {
        net_free(h);
        return set_error(WOST_ERR_DEVICE, "network initialisation failed");
    }
    *out = h;
    return WOST_OK;
}

int wost_net_create(int device, const wost_net_config *cfg, uint64_t seed, wost_net_handle *out)
{
    return net_create_dims(device, cfg, seed, 2, out);
}

int wost3_net_create(int device, const wost_net_config *cfg, uint64_t seed, wost_net_handle *out)
{
    return net_create_dims(device, cfg, seed, 3, out);
}

int wost_net_destroy(wost_net_handle h)
{
    net_free(h);
    return WOST_OK;
}

int wost_net_n_params(wost_net_handle h, uint64_t *n_total, uint64_t *n_mlp)
{
    if (!h || !n_total) return set_error(WOST_ERR_INVALID, "null argument");
    *n_total = h->n_params;
    if (n_mlp) *n_mlp = h->L.n_mlp;
    return WOST_OK;
}

int wost_net_get_params(wost_net_handle h, int which, float *host)
{
    if (!h || !host || which < 0 || which > 2) return set_error(WOST_ERR_INVALID, "bad argument");
    NET_TRY(hipSetDevice(h->device));
    if (which == 2) {
        std::vector<fx_t> fx(h->n_params);
        NET_TRY(hipMemcpy(fx.data(), h->grad, (size_t)h->n_params * sizeof(fx_t), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < h->n_params; ++i) host[i] = (float)((double)fx[i] / kFxScale);
        return WOST_OK;
    }
    const float *src = which == 0 ? h->params : h->inference;
    NET_TRY(hipMemcpy(host, src, (size_t)h->n_params * sizeof(float), hipMemcpyDeviceToHost));
    return WOST_OK;
}

int wost_net_set_gradient_buffer(wost_net_handle h, void *dev_int64)
{
    if (!h) return set_error(WOST_ERR_INVALID, "null argument");
    NET_TRY(hipSetDevice(h->device));
    if (!dev_int64) {                     // back to an internal buffer
        if (!h->grad_owned) {
            h->grad = nullptr;
            NET_TRY(hipMalloc((void **)&h->grad, (size_t)h->n_params * sizeof(fx_t)));
            h->grad_owned = true;
        }
        return WOST_OK;
    }
    if (h->grad && h->grad_owned) (void)hipFree(h->grad);
    h->grad = static_cast<fx_t *>(dev_int64);
    h->grad_owned = false;
    return WOST_OK;
}

int wost_net_set_params(wost_net_handle h, const float *host)
{
    if (!h || !host) return set_error(WOST_ERR_INVALID, "null argument");
    NET_TRY(hipSetDevice(h->device));
    const size_t bytes = (size_t)h->n_params * sizeof(float);
    NET_TRY(hipMemcpy(h->params, host, bytes, hipMemcpyHostToDevice));
    NET_TRY(hipMemcpy(h->inference, host, bytes, hipMemcpyHostToDevice));
    for (float *p : {h->m1, h->m2, h->ema_raw}) NET_TRY(hipMemset(p, 0, bytes));
    NET_TRY(hipMemset(h->param_steps, 0, (size_t)h->n_params * sizeof(uint32_t)));
    h->step = 0;
    int rc = refresh_transposed(h, nullptr);
    if (rc != WOST_OK) return rc;
    NET_TRY(hipDeviceSynchronize());
    return WOST_OK;
}

int wost_net_set_option(wost_net_handle h, const char *key, double value)
{
    if (!h || !key) return set_error(WOST_ERR_INVALID, "null argument");
    const std::string k(key);
    if (k == "precision") {
        if (value != 16 && value != 32) return set_error(WOST_ERR_INVALID, "precision must be 32 (fp32, default) or 16 (half-precision inference)");
        if (value == 16) {
            const NetLayout &L = h->L;
            if (!(L.enc == 32 && L.n_neurons == 64 && L.n_hidden == 3 && L.n_out_padded == 48 && L.n_features == 4 && L.n_levels == 8))
                return set_error(WOST_ERR_UNSUPPORTED, "half-precision inference is built for the reference's network shape only (8 levels x 4 features, 3 x 64, two or three inputs)");
            NET_TRY(hipSetDevice(h->device));
            if (L.dims == 2 && half_image_entries(L) * sizeof(uint2) > 158 * 1024)
                return set_error(WOST_ERR_UNSUPPORTED, "half-precision network: weights and grid must fit into 158 KB of LDS");
            if (!h->inference_h) NET_TRY(hipMalloc((void **)&h->inference_h, half_image_entries(L) * sizeof(uint2)));
            h->precision = 16;
            refresh_half(h, nullptr);
            NET_TRY(hipGetLastError());
            NET_TRY(hipDeviceSynchronize());
        } else {
            h->precision = 32;
        }
        return WOST_OK;
    }
    if (k == "train_precision") {
        if (value != 16 && value != 32) return set_error(WOST_ERR_INVALID, "train_precision must be 32 (fp32, default) or 16 (half-precision training passes)");
        if (value == 16) {
            const NetLayout &L = h->L;
            if (!(L.enc == 32 && L.n_neurons == 64 && L.n_hidden == 3 && L.n_out_padded == 48 && L.n_features == 4 && L.n_levels == 8))
                return set_error(WOST_ERR_UNSUPPORTED, "half-precision training is built for the reference's network shape only (8 levels x 4 features, 3 x 64, two or three inputs)");
            NET_TRY(hipSetDevice(h->device));
            if (L.dims == 2 && half_image_entries(L) * sizeof(uint2) > 158 * 1024)
                return set_error(WOST_ERR_UNSUPPORTED, "half-precision network: weights and grid must fit into 158 KB of LDS");
            if (!h->params_h) NET_TRY(hipMalloc((void **)&h->params_h, half_image_entries(L) * sizeof(uint2)));
            if (!h->params_hb) NET_TRY(hipMalloc((void **)&h->params_hb, (size_t)L.n_mlp / 4 * sizeof(uint2)));
            if (!h->train_partial) NET_TRY(hipMalloc((void **)&h->train_partial, (size_t)256 * L.n_mlp * sizeof(float)));
            h->train_precision = 16;
            refresh_half(h, nullptr);
            NET_TRY(hipGetLastError());
            NET_TRY(hipDeviceSynchronize());
        } else {
            h->train_precision = 32;
        }
        return WOST_OK;
    }
    return set_error(WOST_ERR_INVALID, "unknown option: " + k);
}

int wost_net_inference(wost_net_handle h, const float *xy, int32_t n, float *out, int use_inference_params)
{
    if (!h || !xy || !out || n < 0) return set_error(WOST_ERR_INVALID, "bad argument");
    if (n == 0) return WOST_OK;
    NET_TRY(hipSetDevice(h->device));
    int rc = ensure_points(h, (size_t)n);
    if (rc != WOST_OK) return rc;
    NET_TRY(hipMemcpy(h->d_xy, xy, (size_t)n * h->L.dims * sizeof(float), hipMemcpyHostToDevice));
    if (!use_inference_params && h->train_precision == 16) {
        // the training weights as a half-precision training step evaluates them
        float *o = nullptr, *dl = nullptr;
        rc = wost::net_forward_train_dev(h, h->d_xy, n, nullptr, &o, &dl);
    } else {
        rc = launch_forward(h, use_inference_params != 0, h->d_xy, n, nullptr, h->d_out, nullptr, nullptr);
    }
    if (rc != WOST_OK) return rc;
    NET_TRY(hipMemcpy(out, h->d_out, (size_t)n * h->L.n_out * sizeof(float), hipMemcpyDeviceToHost));
    return WOST_OK;
}

int wost_net_train_step(wost_net_handle h, const float *xy, const float *dl_dout, int32_t n, float loss_scale,
                        int apply_update)
{
    if (!h || !xy || !dl_dout || n <= 0 || !(loss_scale > 0.0f)) return set_error(WOST_ERR_INVALID, "bad argument");
    NET_TRY(hipSetDevice(h->device));
    int rc = ensure_points(h, (size_t)n);
    if (rc != WOST_OK) return rc;
    NET_TRY(hipMemcpy(h->d_xy, xy, (size_t)n * h->L.dims * sizeof(float), hipMemcpyHostToDevice));
    float *out = nullptr, *dl = nullptr;
    rc = wost::net_forward_train_dev(h, h->d_xy, n, nullptr, &out, &dl);
    if (rc != WOST_OK) return rc;
    NET_TRY(hipMemcpy(dl, dl_dout, (size_t)n * h->L.n_out * sizeof(float), hipMemcpyHostToDevice));
    rc = wost::net_backward_update_dev(h, h->d_xy, n, loss_scale, apply_update, nullptr);
    if (rc != WOST_OK) return rc;
    NET_TRY(hipDeviceSynchronize());
    return WOST_OK;
}

}  // extern "C"
