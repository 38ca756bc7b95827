// wost_net_device.h -- what other kernels of the library need from the guiding network (wost_net.hip):
// its layout, and the half-precision forward pass of ONE 16-point unit as a device function, so that a walk
// kernel can evaluate the network for the walkers of its own wave (wost_guided.hip, the fused sample kernel)
// with exactly the arithmetic of net_forward_h_kernel.  Not part of the C-ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/wost.h"

namespace wost {

constexpr int kNetMaxLevels = 16;

struct NetLayout {
    int32_t res[kNetMaxLevels];
    float scale[kNetMaxLevels];
    uint32_t level_off[kNetMaxLevels + 1];  // entries (x n_features floats)
    int32_t n_levels, n_features, enc, n_neurons, n_hidden, n_out, n_out_padded;
    uint32_t n_mlp, n_grid;
    uint32_t w_off[kNetMaxLevels];  // offset of every weight matrix in the parameter vector
    int32_t dims;                   // input dimensions: 2 (GuidedIntegrator<2>) or 3 (GuidedIntegrator<3>: trilinear grid, res^3 entries per level)
};

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

// gfx950's v_mfma_f32_16x16x32_f16 on TWO of the 16-deep operand pairs this library keeps (fragments of four halves per lane, the
// "chain" layout of the activations): lane (i, g) owns k-slots 8g .. 8g + 7 of the 32-deep product; filling slots 0..3 with the
// features 16 t0 + 4g + c and slots 4..7 with 16 t1 + 4g + c on BOTH operands is a permutation of the summation index, so
// acc += A[.][t0] B[t0][.] + A[.][t1] B[t1][.] in one issue -- no relayout of weights or activations, twice the K per instruction.
__device__ __forceinline__ f32x4_t mfma_k32(h4_t a0, h4_t a1, h4_t b0, h4_t b1, f32x4_t acc)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7),
                                                  acc, 0, 0, 0);
}

// Every accumulator of a layer passes through one of these before anything reads it: all matrix instructions of the layer issued,
// then sixteen idle states, the layer's operands held until then (mfma_hold).  hipcc (ROCm 7.2) otherwise schedules the conversions
// of the first accumulators between the last v_mfma_f32_16x16x32_f16 of the layer, and on MI355X, with four waves of a SIMD issuing
// such layers side by side, a whole 16-point unit then comes out a percent different now and then: three launches of one kernel on
// the same inputs, any one of them the odd one out (EXPERIMENTS 17: what made config 4 in half precision differ between two runs).
// Measured: idle states BETWEEN the matrix instructions or in front of them make it far worse, this statement behind them 30 x better,
// two waves per SIMD instead of four (net_forward_h_kernel's block of 512) remove what was left; one wave per SIMD
// (net_train_h_kernel) never showed it.
#define WOST_NOP16 "s_nop 15\n\t"
#ifdef WOST_MFMA_SETTLE_OFF     // (developer builds: the unpadded kernels of EXPERIMENTS 17)
#define WOST_MFMA_SETTLE ""
#elif defined(WOST_MFMA_SETTLE_64)
#define WOST_MFMA_SETTLE WOST_NOP16 WOST_NOP16 WOST_NOP16 "s_nop 15"
#elif defined(WOST_MFMA_SETTLE_128)
#define WOST_MFMA_SETTLE WOST_NOP16 WOST_NOP16 WOST_NOP16 WOST_NOP16 WOST_NOP16 WOST_NOP16 WOST_NOP16 "s_nop 15"
#else
#define WOST_MFMA_SETTLE "s_nop 15"
#endif
__device__ __forceinline__ void mfma_settle(f32x4_t &a) { asm volatile(WOST_MFMA_SETTLE : "+v"(a)); }
__device__ __forceinline__ void mfma_settle(f32x4_t (&a)[2]) { asm volatile(WOST_MFMA_SETTLE : "+v"(a[0]), "+v"(a[1])); }
__device__ __forceinline__ void mfma_settle(f32x4_t (&a)[3]) { asm volatile(WOST_MFMA_SETTLE : "+v"(a[0]), "+v"(a[1]), "+v"(a[2])); }
__device__ __forceinline__ void mfma_settle(f32x4_t (&a)[4]) { asm volatile(WOST_MFMA_SETTLE : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }
__device__ __forceinline__ void mfma_settle(f32x4_t (&a)[2][4])
{
    asm volatile(WOST_MFMA_SETTLE : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]));
}
__device__ __forceinline__ void mfma_settle3(f32x4_t (&a)[4]) { asm volatile(WOST_MFMA_SETTLE : "+v"(a[0]), "+v"(a[1]), "+v"(a[2])); }
__device__ __forceinline__ void mfma_settle3(f32x4_t (&a)[2][4])
{
    asm volatile(WOST_MFMA_SETTLE : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]));
}
// the operands of a layer held until the same point (nothing may overwrite one while a matrix instruction of the layer can still read it)
__device__ __forceinline__ void mfma_hold(const h4_t (&b)[2]) { asm volatile("" :: "v"(b[0]), "v"(b[1])); }
__device__ __forceinline__ void mfma_hold(const h4_t (&b)[3]) { asm volatile("" :: "v"(b[0]), "v"(b[1]), "v"(b[2])); }
__device__ __forceinline__ void mfma_hold(const h4_t (&b)[4]) { asm volatile("" :: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])); }
__device__ __forceinline__ void mfma_hold(const h4_t (&b)[2][4])
{
    asm volatile("" :: "v"(b[0][0]), "v"(b[0][1]), "v"(b[0][2]), "v"(b[0][3]), "v"(b[1][0]), "v"(b[1][1]), "v"(b[1][2]), "v"(b[1][3]));
}

// The half-precision image of the inference weights ("precision" 16): MFMA fragments of the four matrices
// (n_mlp / 4 entries of 8 bytes, layout of fragment_mlp_h_kernel), then the grid entry by entry (4 features in f16).
// Refreshed by the network after every optimizer step.
struct HalfNetView {
    NetLayout L;
    const uint2 *image;
};
// WOST_ERR_UNSUPPORTED unless the network runs its inference in half precision
int net_half_view(wost_net_handle h, HalfNetView *out);

// The fp32 network (the bit-exact mode) for kernels that evaluate it themselves: `frag` = the MFMA fragments of the
// inference matrices (n_mlp floats, layout of fragment_mlp_kernel), `grid` = the fp32 grid of the inference weights.
struct F32NetView {
    NetLayout L;
    const float *frag, *grid;
};
// WOST_ERR_UNSUPPORTED unless the network has the reference's shape (the MFMA kernels); net_f32_view: two inputs, net_f32_view3: three
int net_f32_view(wost_net_handle h, F32NetView *out);
int net_f32_view3(wost_net_handle h, F32NetView *out);

// A frozen copy of what the views above point at (the pipelined training order of the guided solve: a sample walks with the
// weights of an earlier training pass while the optimizer rewrites the network's own images on another stream).
// net_snapshot_bytes: size of one copy (0: the network offers neither view); net_snapshot_dev copies the current inference
// weights into `dst` on `stream`; net_snapshot_views points the views at a copy (the layout members are filled as by the
// view calls; *half tells which of the two is valid).
size_t net_snapshot_bytes(wost_net_handle h);
int net_snapshot_dev(wost_net_handle h, void *dst, hipStream_t stream);
int net_snapshot_views(wost_net_handle h, const void *snap, bool *half, HalfNetView *hv, F32NetView *fv);

// one level of the DenseGrid encoding of (x, y), fp32 (tiny-cuda-nn grid.h semantics as restated by the oracle): the
// arithmetic of net_forward_mfma_kernel, which calls this function
__device__ __forceinline__ float4 f32_encode_level(const float *grid, float sc, uint32_t res, uint32_t lo, uint32_t n_level, float x, float y)
{
    float px = __builtin_fmaf(sc, x, 0.5f), py = __builtin_fmaf(sc, y, 0.5f);
    const float fx = floorf(px), fy = floorf(py);
    px -= fx;
    py -= fy;
    const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
    float4 c[4];
    float w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t cx = ix + (k & 1), cy = iy + (k >> 1);
        w[k] = ((k & 1) ? px : 1.0f - px) * ((k >> 1) ? py : 1.0f - py);
        uint32_t idx = cx + cy * res;
        if (idx >= n_level) {
            idx -= n_level;
            if (idx >= n_level) idx %= n_level;
        }
        c[k] = *reinterpret_cast<const float4 *>(grid + (size_t)(lo + idx) * 4);
    }
    float4 f = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f.x += w[k] * c[k].x; f.y += w[k] * c[k].y; f.z += w[k] * c[k].z; f.w += w[k] * c[k].w;
    }
    return f;
}

// the same for three inputs (GuidedIntegrator<3>): the eight corners of the cell, the weight of corner k = (wx wy) wz with bit 0 = +x,
// bit 1 = +y, bit 2 = +z, the dense index in 64 bits -- the arithmetic of encode_point (wost_net.hip)
__device__ __forceinline__ float4 f32_encode_level3(const float *grid, float sc, uint32_t res, uint32_t lo, uint32_t n_level, float x, float y, float z)
{
    float px = __builtin_fmaf(sc, x, 0.5f), py = __builtin_fmaf(sc, y, 0.5f), pz = __builtin_fmaf(sc, z, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    px -= fx;
    py -= fy;
    pz -= fz;
    const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy, iz = (uint32_t)(int)fz;
    float4 c[8];
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t cx = ix + (k & 1), cy = iy + ((k >> 1) & 1), cz = iz + ((k >> 2) & 1);
        w[k] = (((k & 1) ? px : 1.0f - px) * ((k & 2) ? py : 1.0f - py)) * ((k & 4) ? pz : 1.0f - pz);
        const uint32_t idx = (uint32_t)(((unsigned long long)cx + (unsigned long long)cy * res + (unsigned long long)cz * res * res) % n_level);
        c[k] = *reinterpret_cast<const float4 *>(grid + (size_t)(lo + idx) * 4);
    }
    float4 f = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f.x += w[k] * c[k].x; f.y += w[k] * c[k].y; f.z += w[k] * c[k].z; f.w += w[k] * c[k].w;
    }
    return f;
}

// The four matrices on one 16-point unit in fp32 (v_mfma_f32_16x16x4_f32, k ascending: bit-identical to the scalar
// fmaf chains of the oracle), reference network shape.  Lane (i, g) supplies in[s] = encoded feature 4 s + g of point i
// (s = 0 .. 7) and receives out[4 rt + c] = output 16 rt + 4 c + g of point i.  `wf` = the fragments in LDS.
__device__ __forceinline__ void f32_mlp_unit(const float *wf, const uint32_t (&w_off)[4], int lane, const float (&in)[8], float (&out)[12])
{
    float b[16];
#pragma unroll
    for (int s = 0; s < 8; ++s) b[s] = in[s];
    f32x4_t acc[4];
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
        const int S = layer == 0 ? 8 : 16;
        const float *w = wf + w_off[layer];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s < S) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[(rt * S + s) * 64 + lane], b[s], acc[rt], 0, 0, 0);
            }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int c = 0; c < 4; ++c) b[4 * rt + c] = fmaxf(acc[rt][c], 0.0f);
    }
    const float *w3 = wf + w_off[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) acc[rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w3[(rt * 16 + s) * 64 + lane], b[s], acc[rt], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[4 * rt + c] = acc[rt][c];
}

// The same for NU units at once with the fragments in GLOBAL memory (the fused GuidedIntegrator<3> kernel, whose LDS belongs to the tree
// queries): every fragment is fetched once for all units -- a wave of 64 points reads the 53 KB of matrices once, not four times --
// and D k-steps ahead of the matrix instructions that use it (left to itself the compiler issues all 208 loads of the pass first:
// 256 + 256 registers and scratch).  Per unit the instructions and their order are those of f32_mlp_unit: same bits.
template <int NU, int D, int S, int RT>
__device__ __forceinline__ void f32_mlp_layer(const float *w, int lane, const float (&b)[NU][16], f32x4_t (&acc)[NU][4])
{
    float ring[D][RT];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) ring[d][rt] = d < S ? w[(rt * S + d) * 64 + lane] : 0.0f;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[u][rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        float cur[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) cur[rt] = ring[s % D][rt];
        if (s + D < S) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) ring[s % D][rt] = w[(rt * S + s + D) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[rt], b[u][s], acc[u][rt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// b[u][0 .. 7]: in, the features 4 s + g of point i of unit u in lane (i, g); out: b[u][4 rt + c] = output 16 rt + 4 c + g, rt < 3
template <int NU>
__device__ __forceinline__ void f32_mlp_units(const float *wf, const uint32_t (&w_off)[4], int lane, float (&b)[NU][16])
{
    constexpr int D = NU >= 4 ? 2 : 4;
    f32x4_t acc[NU][4];
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
        if (layer == 0) f32_mlp_layer<NU, D, 8, 4>(wf + w_off[0], lane, b, acc);
        else f32_mlp_layer<NU, D, 16, 4>(wf + w_off[layer], lane, b, acc);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int c = 0; c < 4; ++c) b[u][4 * rt + c] = fmaxf(acc[u][rt][c], 0.0f);
    }
    f32_mlp_layer<NU, D, 16, 3>(wf + w_off[3], lane, b, acc);
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int rt = 0; rt < 3; ++rt)
#pragma unroll
            for (int c = 0; c < 4; ++c) b[u][4 * rt + c] = acc[u][rt][c];
}

// one level of the DenseGrid encoding of (x, y) in the half-precision network: grid values as stored (f16),
// bilinear interpolation in fp32, result rounded to f16.  `grid` = the entries of the image (LDS or global).
__device__ __forceinline__ h4_t half_encode_level(const uint2 *grid, float sc, uint32_t res, uint32_t lo, uint32_t n_level, float x, float y)
{
    float px = __builtin_fmaf(sc, x, 0.5f), py = __builtin_fmaf(sc, y, 0.5f);
    const float fx = floorf(px), fy = floorf(py);
    px -= fx;
    py -= fy;
    const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
    union { uint2 u; h4_t h; } c[4];
    float w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t cx = ix + (k & 1), cy = iy + (k >> 1);
        w[k] = ((k & 1) ? px : 1.0f - px) * ((k >> 1) ? py : 1.0f - py);
        uint32_t idx = cx + cy * res;
        if (idx >= n_level) {
            idx -= n_level;
            if (idx >= n_level) idx %= n_level;
        }
        c[k].u = grid[lo + idx];      // the grid as the half-precision network holds it
    }
    float4 f = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f.x += w[k] * (float)c[k].h[0]; f.y += w[k] * (float)c[k].h[1];
        f.z += w[k] * (float)c[k].h[2]; f.w += w[k] * (float)c[k].h[3];
    }
    return h4_t{(_Float16)f.x, (_Float16)f.y, (_Float16)f.z, (_Float16)f.w};
}

// the same for three inputs (GuidedIntegrator<3> in the reference's network precision, integrator/guided/integrator.h:54): the eight
// corners of the cell as in f32_encode_level3 -- weight of corner k = (wx wy) wz, dense index in 64 bits -- on the f16 grid
__device__ __forceinline__ h4_t half_encode_level3(const uint2 *grid, float sc, uint32_t res, uint32_t lo, uint32_t n_level, float x, float y, float z)
{
    float px = __builtin_fmaf(sc, x, 0.5f), py = __builtin_fmaf(sc, y, 0.5f), pz = __builtin_fmaf(sc, z, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    px -= fx;
    py -= fy;
    pz -= fz;
    const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy, iz = (uint32_t)(int)fz;
    union { uint2 u; h4_t h; } c[8];
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t cx = ix + (k & 1), cy = iy + ((k >> 1) & 1), cz = iz + ((k >> 2) & 1);
        w[k] = (((k & 1) ? px : 1.0f - px) * ((k & 2) ? py : 1.0f - py)) * ((k & 4) ? pz : 1.0f - pz);
        const uint32_t idx = (uint32_t)(((unsigned long long)cx + (unsigned long long)cy * res + (unsigned long long)cz * res * res) % n_level);
        c[k].u = grid[lo + idx];
    }
    float4 f = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f.x += w[k] * (float)c[k].h[0]; f.y += w[k] * (float)c[k].h[1];
        f.z += w[k] * (float)c[k].h[2]; f.w += w[k] * (float)c[k].h[3];
    }
    return h4_t{(_Float16)f.x, (_Float16)f.y, (_Float16)f.z, (_Float16)f.w};
}

// The four matrices on one 16-point unit, reference network shape (32 -> 64 -> 64 -> 64 -> 48).  Lane l = (i = l & 15,
// g = l >> 4) supplies in[h] = the encoding of levels g + 4h of point i and receives out[rt][c] = output
// 16 rt + 4 g + c of point i, rounded to f16 like the network's outputs.  `wf` = the fragment part of the image
// (LDS), w_off4[l] = w_off[l] / 4.  Must be called by all 64 lanes together.
__device__ __forceinline__ void half_mlp_unit(const uint2 *wf, const uint32_t (&w_off4)[4], int lane, const h4_t (&in)[2], h4_t (&out)[3])
{
    const h4_t zero = h4_t{(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
    h4_t b[4] = {in[0], in[1], zero, zero};
    f32x4_t acc[4];
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
        const int KT = layer == 0 ? 2 : 4;
        const uint2 *w = wf + w_off4[layer];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kt = 0; kt < 4; kt += 2)
            if (kt < KT) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    union { uint2 u; h4_t h; } a0, a1;
                    a0.u = w[(rt * KT + kt) * 64 + lane];
                    a1.u = w[(rt * KT + kt + 1) * 64 + lane];
                    acc[rt] = mfma_k32(a0.h, a1.h, b[kt], b[kt + 1], acc[rt]);
                }
            }
        mfma_settle(acc);
        mfma_hold(b);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) b[rt] = __builtin_elementwise_max(__builtin_convertvector(acc[rt], h4_t), zero);
    }
    const uint2 *w3 = wf + w_off4[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) acc[rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kt = 0; kt < 4; kt += 2)
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) {
            union { uint2 u; h4_t h; } a0, a1;
            a0.u = w3[(rt * 4 + kt) * 64 + lane];
            a1.u = w3[(rt * 4 + kt + 1) * 64 + lane];
            acc[rt] = mfma_k32(a0.h, a1.h, b[kt], b[kt + 1], acc[rt]);
        }
    mfma_settle3(acc);
    mfma_hold(b);
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) out[rt] = __builtin_convertvector(acc[rt], h4_t);
}

}  // namespace wost
