// wost_net_half.h -- half-precision TRAINING kernel of the guiding network (included by wost_net.hip
// after its layout / fixed-point definitions and the half-precision inference kernel).
//
// "train_precision" 16 of wost_net_set_option: the forward pass, the backward pass and the weight
// gradients of a training step run on v_mfma_f32_16x16x32_f16 (layers, backward products: mfma_k32) and v_mfma_f32_16x16x16_f16
// (weight gradients: K = the 16 points of a unit) with f16 operands and fp32
// accumulation -- the arithmetic the reference's tiny-cuda-nn network trains in (util/network.h:21-196,
// half precision with loss scale 128, guided/parameters.h:13).  Master weights, Adam and the EMA stay
// fp32; the gradient sums stay 64-bit fixed point, so a step is still reproducible bit for bit (the
// matrix instruction is deterministic, the sums across waves are integers).  Not bit-comparable with
// the fp32 path or the oracle: gated by the tolerance tests of tests/test_guided_network.py.
//
// A training step is three launches instead of the fp32 path's activations-through-HBM pipeline:
//   1. net_forward_h_kernel (the inference kernel, training weights) -> raw outputs, and the f16
//      encoding of every point (64 bytes) so that nothing is gathered twice;
//   2. the loss-gradient kernel of the caller (dL/dout);
//   3. net_train_h_kernel: ONE kernel that recomputes the hidden layers from the stored encoding
//      (f16 MFMAs are 16x cheaper than the fp32 ones -- recomputing costs less than reading the
//      activations back), runs the backward pass and accumulates ALL weight gradients in registers
//      (13 312 weights = 52 tiles of 16 x 16 = 208 accumulator registers per lane, one wave per SIMD),
//      and writes dL/d(encoding) for the grid-gradient kernel.  No activation or delta ever leaves the CU.
//
// Two operand layouts of a 16-point unit u (points 16u .. 16u+15), lane l = (i = l & 15, g = l >> 4):
//   chain  C(X, t): lane holds X[feature 16t + 4g + c][point i],      c = 0..3  -- what a layer hands to
//                   the next one (it is both the D layout of mfma(W, X) and its B operand layout);
//   turned T(X, t): lane holds X[feature 16t + i][point 4g + c],      c = 0..3  -- what the weight
//                   gradient dW[r][k] = sum_p delta[r][p] a[k][p] needs for BOTH operands (the sum over
//                   the 16 points of the unit is the K dimension of one instruction).
// Turning costs no shuffle and no LDS: the registers of C(X, t) are also a valid A operand
// [point][feature], and one product with the 16 x 16 identity delivers it transposed, exactly
// (a single non-zero product per element).
#pragma once

namespace wost {

union HalfFrag {
    uint2 u;
    h4_t h;
};

// backward fragments: fragb[(w_off[layer] / 4) + (kt * RT + rt) * 64 + lane] = W[16 rt + 4g .. 4g + 3][16 kt + i]
// i.e. the A operand [M = k][K = r] of delta_in = W^T delta_out, RT = n_o / 16
__global__ void fragment_mlp_hb_kernel(NetLayout L, const float *src, uint2 *dst)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.n_mlp / 4) return;
    int layer = 0;
    while (layer < L.n_hidden && 4 * e >= L.w_off[layer + 1]) ++layer;
    const int n_i = layer == 0 ? L.enc : L.n_neurons, n_o = layer == L.n_hidden ? L.n_out_padded : L.n_neurons;
    const int RT = n_o / 16;
    const uint32_t f = e - L.w_off[layer] / 4, l = f & 63u, t = f >> 6;
    const uint32_t kt = t / RT, rt = t % RT, i = l & 15u, g = l >> 4;
    const float *w = src + L.w_off[layer];
    HalfFrag v;
    const uint32_t k = 16 * kt + i, r0 = 16 * rt + 4 * g;
    v.h = h4_t{(_Float16)w[(size_t)(r0 + 0) * n_i + k], (_Float16)w[(size_t)(r0 + 1) * n_i + k], (_Float16)w[(size_t)(r0 + 2) * n_i + k],
               (_Float16)w[(size_t)(r0 + 3) * n_i + k]};
    dst[e] = v.u;
}

__device__ __forceinline__ h4_t identity_frag(int i, int g)
{
    // B[K = 4g + c][N = i] (or A[M = i][K = 4g + c]) of the 16 x 16 identity
    return h4_t{(_Float16)(4 * g + 0 == i ? 1.0f : 0.0f), (_Float16)(4 * g + 1 == i ? 1.0f : 0.0f), (_Float16)(4 * g + 2 == i ? 1.0f : 0.0f),
                (_Float16)(4 * g + 3 == i ? 1.0f : 0.0f)};
}

__device__ __forceinline__ h4_t relu_pack(f32x4_t v)
{
    // round, then clamp: two packed instructions per pair (rounding is monotonic, the result is the same)
    return __builtin_elementwise_max(__builtin_convertvector(v, h4_t), h4_t{(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f});
}

__device__ __forceinline__ _Float16 sat_h(float v) { return (_Float16)__builtin_fminf(__builtin_fmaxf(v, -65504.0f), 65504.0f); }

__device__ __forceinline__ h4_t turn_h(h4_t x, h4_t ident)
{
    f32x4_t z = __builtin_amdgcn_mfma_f32_16x16x16f16(x, ident, f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
    mfma_settle(z);
    return h4_t{(_Float16)z[0], (_Float16)z[1], (_Float16)z[2], (_Float16)z[3]};
}

// hidden layer, chain layout: out[rt] = relu(sum_kt W[rt][kt] in[kt]) rounded to f16
template <int KT>
__device__ __forceinline__ void hidden_h(const uint2 *wf, int lane, const h4_t (&in)[KT], h4_t (&out)[4])
{
    f32x4_t acc[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    static_assert(KT % 2 == 0, "pairs of 16-deep operands (mfma_k32)");
#pragma unroll
    for (int kt = 0; kt < KT; kt += 2)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            HalfFrag a0, a1;
            a0.u = wf[(rt * KT + kt) * 64 + lane];
            a1.u = wf[(rt * KT + kt + 1) * 64 + lane];
            acc[rt] = mfma_k32(a0.h, a1.h, in[kt], in[kt + 1], acc[rt]);
        }
    mfma_settle(acc);
    mfma_hold(in);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) out[rt] = relu_pack(acc[rt]);
}

// weight gradient of one layer over the 16 points of a unit: g[rt][kt] += T(delta, rt) x T(a, kt)
template <int RT, int KT>
__device__ __forceinline__ void wgrad_h(const h4_t (&d)[4], const h4_t (&a)[KT], h4_t ident, f32x4_t (&gacc)[RT][KT])
{
    // the transposes: all their matrix instructions, then the conversions (mfma_settle)
    h4_t td[RT], ta[KT];
    f32x4_t zd[RT], za[KT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) zd[rt] = __builtin_amdgcn_mfma_f32_16x16x16f16(d[rt], ident, f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) za[kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a[kt], ident, f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
    mfma_settle(zd);
    mfma_settle(za);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) td[rt] = h4_t{(_Float16)zd[rt][0], (_Float16)zd[rt][1], (_Float16)zd[rt][2], (_Float16)zd[rt][3]};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) ta[kt] = h4_t{(_Float16)za[kt][0], (_Float16)za[kt][1], (_Float16)za[kt][2], (_Float16)za[kt][3]};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) gacc[rt][kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(td[rt], ta[kt], gacc[rt][kt], 0, 0, 0);
}

// backward through one layer, chain layout: acc[kt] = sum_rt W^T[kt][rt] d[rt]
template <int RT, int KT>
__device__ __forceinline__ void back_h(const uint2 *wb, int lane, const h4_t (&d)[4], f32x4_t (&acc)[KT])
{
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[kt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    // pairs of output tiles in one 32-deep instruction (mfma_k32); the 48-row output layer leaves one tile over
#pragma unroll
    for (int rt = 0; rt + 1 < RT; rt += 2)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            HalfFrag a0, a1;
            a0.u = wb[(kt * RT + rt) * 64 + lane];
            a1.u = wb[(kt * RT + rt + 1) * 64 + lane];
            acc[kt] = mfma_k32(a0.h, a1.h, d[rt], d[rt + 1], acc[kt]);
        }
    if (RT % 2) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            HalfFrag a;
            a.u = wb[(kt * RT + RT - 1) * 64 + lane];
            acc[kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a.h, d[RT - 1], acc[kt], 0, 0, 0);
        }
    }
    mfma_settle(acc);
    mfma_hold(d);
}

// delta of the hidden layer whose activations are a: relu'(a) * acc, saturated, f16
__device__ __forceinline__ void mask_h(const f32x4_t (&acc)[4], const h4_t (&a)[4], h4_t (&d)[4])
{
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
        d[kt] = h4_t{(float)a[kt][0] > 0.0f ? sat_h(acc[kt][0]) : (_Float16)0.0f, (float)a[kt][1] > 0.0f ? sat_h(acc[kt][1]) : (_Float16)0.0f,
                     (float)a[kt][2] > 0.0f ? sat_h(acc[kt][2]) : (_Float16)0.0f, (float)a[kt][3] > 0.0f ? sat_h(acc[kt][3]) : (_Float16)0.0f};
}

// block-level sum of the four waves' accumulators in LDS, in wave order (fixed order: reproducible)
template <int RT, int KT>
__device__ __forceinline__ void flush_h(const f32x4_t (&gacc)[RT][KT], int lane, int n_i, bool first, float *red)
{
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float *p = red + (16 * rt + 4 * g + c) * n_i + 16 * kt + i;
                *p = first ? gacc[rt][kt][c] : *p + gacc[rt][kt][c];
            }
}

// enc: the f16 encoding as net_forward_h_kernel stored it, [unit][2][64 lanes] (chain tiles 0 and 1);
// dl: rows of n_out floats (dL/dout times the loss scale); dscale: extra power-of-two scale of the
// deltas while they are f16 (loss scale 128 / batch 524 288 ~ 2e-4 sits at the f16 subnormal edge;
// chosen by the host from the batch size, divided out before anything leaves the kernel);
// denc: level-major, [8 levels][n points][4 features]; partial: one row of n_mlp block sums per block
__global__ __launch_bounds__(kHalfThreads, 1) void net_train_h_kernel(NetLayout L, const uint2 *fragh, const uint2 *fragb, const uint2 *enc, const float *dl,
                                                                        int n, float dscale, float *denc, float *partial)
{
    extern __shared__ uint2 lds_h[];
    const uint32_t nf = L.n_mlp / 4;
    for (uint32_t e = threadIdx.x; e < nf; e += kHalfThreads) {
        lds_h[e] = fragh[e];
        lds_h[nf + e] = fragb[e];
    }
    __syncthreads();
    const uint2 *wf0 = lds_h + L.w_off[0] / 4, *wf1 = lds_h + L.w_off[1] / 4, *wf2 = lds_h + L.w_off[2] / 4;
    const uint2 *wb0 = lds_h + nf + L.w_off[0] / 4, *wb1 = lds_h + nf + L.w_off[1] / 4, *wb2 = lds_h + nf + L.w_off[2] / 4, *wb3 = lds_h + nf + L.w_off[3] / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int n_units = (n + 15) / 16;
    const h4_t ident = identity_frag(i, g);
    const float inv_scale = 1.0f / dscale;
    f32x4_t g0[4][2], g1[4][4], g2[4][4], g3[3][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < 2) g0[rt][kt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            g1[rt][kt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            g2[rt][kt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            if (rt < 3) g3[rt][kt] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        }
    // the inputs of a unit (2 encoding tiles, 12 values of dL/dout per lane) are fetched one unit ahead:
    // with one wave per SIMD nothing else hides the memory latency
    const int stride = gridDim.x * (kHalfThreads / 64);
    uint2 enc_next[2];
    float dl_next[3][4];
    auto fetch = [&](int unit) {
        const int pt = unit * 16 + i;
        const int u = unit < n_units ? unit : n_units - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) enc_next[h] = enc[((size_t)u * 2 + h) * 64 + lane];
        // addresses clamped instead of branches: all twelve loads in flight at once
        const float *row = dl + (size_t)(pt < n ? pt : n - 1) * L.n_out;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int o = 16 * t + 4 * g + c;
                dl_next[t][c] = row[o < L.n_out ? o : 0];
            }
    };
    int unit = blockIdx.x * (kHalfThreads / 64) + wave;
    if (unit < n_units) fetch(unit);
    for (; unit < n_units; unit += stride) {
        // the weight fragments are re-read from LDS in every iteration (kept in registers they would take 208
        // of them and spill the accumulators)
        asm volatile("" ::: "memory");
        const int pt = unit * 16 + i;
        const bool valid = pt < n;
        h4_t a0[2], a1[4], a2[4], a3[4], d[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            HalfFrag e;
            e.u = enc_next[h];
            a0[h] = e.h;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (valid && 16 * t + 4 * g + c < L.n_out) ? dl_next[t][c] * dscale : 0.0f;
            d[t] = h4_t{sat_h(v[0]), sat_h(v[1]), sat_h(v[2]), sat_h(v[3])};
        }
        fetch(unit + stride);
        asm volatile("" ::: "memory");      // keeps the loads up here, ahead of the unit's arithmetic
        d[3] = h4_t{(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
        hidden_h<2>(wf0, lane, a0, a1);
        hidden_h<4>(wf1, lane, a1, a2);
        hidden_h<4>(wf2, lane, a2, a3);
        f32x4_t acc[4];
        wgrad_h<3, 4>(d, a3, ident, g3);
        back_h<3, 4>(wb3, lane, d, acc);
        mask_h(acc, a3, d);
        wgrad_h<4, 4>(d, a2, ident, g2);
        back_h<4, 4>(wb2, lane, d, acc);
        mask_h(acc, a2, d);
        wgrad_h<4, 4>(d, a1, ident, g1);
        back_h<4, 4>(wb1, lane, d, acc);
        mask_h(acc, a1, d);
        wgrad_h<4, 2>(d, a0, ident, g0);
        // dL/d(encoding) = W0^T delta1: 32 features = the levels g and g + 4 of this lane's point
        f32x4_t e2[2];
        back_h<4, 2>(wb0, lane, d, e2);
        if (valid) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)      // level 4 kt + g of point pt, level-major
                *reinterpret_cast<float4 *>(denc + ((size_t)(4 * kt + g) * n + pt) * 4) =
                    float4{e2[kt][0] * inv_scale, e2[kt][1] * inv_scale, e2[kt][2] * inv_scale, e2[kt][3] * inv_scale};
        }
    }
    // ---- the block's sums: wave after wave into LDS (over the weights, no longer needed), then one
    // row of the partial-sum table; net_train_h_reduce_kernel adds the rows in fixed point
    float *red = reinterpret_cast<float *>(lds_h);
    for (int w = 0; w < kHalfThreads / 64; ++w) {
        __syncthreads();
        if (wave == w) {
            flush_h<4, 2>(g0, lane, 32, w == 0, red + L.w_off[0]);
            flush_h<4, 4>(g1, lane, 64, w == 0, red + L.w_off[1]);
            flush_h<4, 4>(g2, lane, 64, w == 0, red + L.w_off[2]);
            flush_h<3, 4>(g3, lane, 64, w == 0, red + L.w_off[3]);
        }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < L.n_mlp; e += kHalfThreads) partial[(size_t)blockIdx.x * L.n_mlp + e] = red[e] * inv_scale;
}

// grad[j] += sum over the rows of the table, in fixed point; blockIdx.y = a group of 16 rows
__global__ void net_train_h_reduce_kernel(NetLayout L, const float *partial, int rows, fx_t *grad)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= L.n_mlp) return;
    const int r0 = blockIdx.y * 16, r1 = min(rows, r0 + 16);
    fx_t s = 0;
    for (int r = r0; r < r1; ++r) {
        const float v = partial[(size_t)r * L.n_mlp + j];
        if (v != 0.0f) s += to_fx(v);
    }
    if (s != 0) fx_add(grad + j, s);
}

}  // namespace wost
