/*
 * wost_net.c -- CPU oracle of the guiding network (the optimizer of
 * a27): DenseGrid encoding -> bias-free ReLU MLP -> raw mixture parameters, plus backward pass
 * and the Adam-in-EMA optimizer step.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference builds this from tiny-cuda-nn (ext/tcnn, https://github.com/NVlabs/tiny-cuda-nn,
 * .gitmodules:1-3, pinned commit unknown, directory EMPTY in /root/reference) through the adapter
 * util/network.h:21-196 with the configuration data/ladybug/n.json:49-81.  There is no test or
 * golden vector for it in the reference => PARITY UNPINNED; this file restates tiny-cuda-nn's
 * published algorithms (grid.h: grid_scale / grid_resolution / dense grid_index / linear
 * interpolation; fully_fused_mlp.h: y = W x without bias, ReLU hidden, no output activation;
 * adam.h adam_step; ema.h debiased exponential moving average) in plain fp32.  The reference
 * runs the MLP in half precision on tensor cores; fp32 here is the numerically stronger
 * stand-in the HIP path is compared against.
 *
 * Parameter vector order (util/network.h:99-117: network first, then encoding):
 *   [W1: n_neurons x enc] [W(hidden-1) x: n_neurons x n_neurons] [Wout: n_out_padded x n_neurons]
 *   [grid level 0 .. L-1: res^D (rounded up to 8) x n_features]
 *
 * D = 2 is the network of GuidedIntegrator<2> (guided/parameters.h:16-24: 2 inputs, 8 x 4 + 1 = 33 outputs); D = 3 the one of
 * GuidedIntegrator<3> (:26-33: 3 inputs, 8 x 5 + 1 = 41 outputs): trilinear interpolation over the 8 corners of a cell, dense index
 * x + y res + z res^2 (grid.h grid_index, stride *= resolution per dimension).  The wo_net_* entry points are the 2-D network, the
 * wo_net3_* ones the 3-D network; both run the same code below.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "wost_oracle.h"

#define WN_MAX_LEVELS 16

typedef struct {
    int res[WN_MAX_LEVELS];
    float scale[WN_MAX_LEVELS];
    size_t level_off[WN_MAX_LEVELS + 1]; /* in entries (x n_features floats) */
    int enc;        /* encoded width */
    size_t n_mlp, n_grid;
} wn_layout;

static void layout_d(const wo_net_config *c, int dims, wn_layout *l)
{
    const float log2s = log2f(c->per_level_scale);
    size_t off = 0;
    for (int i = 0; i < c->n_levels; ++i) {
        l->scale[i] = exp2f((float)i * log2s) * (float)c->base_resolution - 1.0f;   /* grid.h grid_scale */
        l->res[i] = (int)ceilf(l->scale[i]) + 1;                                     /* grid.h grid_resolution */
        size_t n = (size_t)l->res[i] * l->res[i];
        if (dims == 3) n *= (size_t)l->res[i];
        n = (n + 7) / 8 * 8;
        l->level_off[i] = off;
        off += n;
    }
    l->level_off[c->n_levels] = off;
    l->enc = c->n_levels * c->n_features;
    l->n_grid = off * c->n_features;
    l->n_mlp = (size_t)c->n_neurons * l->enc + (size_t)(c->n_hidden_layers - 1) * c->n_neurons * c->n_neurons +
               (size_t)c->n_output_padded * c->n_neurons;
}

static uint64_t n_params_d(const wo_net_config *c, int dims)
{
    wn_layout l;
    layout_d(c, dims, &l);
    return l.n_mlp + l.n_grid;
}
uint64_t wo_net_n_params(const wo_net_config *c) { return n_params_d(c, 2); }
uint64_t wo_net3_n_params(const wo_net_config *c) { return n_params_d(c, 3); }

/* encode one point (x[dims]); optionally return the 2^dims corner indices/weights per level for backward.
 * Corner k of a cell: bit 0 = +x, bit 1 = +y, bit 2 = +z; weight = product of the per-axis weights in the order x, y(, z). */
static void encode(const wo_net_config *c, int dims, const wn_layout *l, const float *grid, const float *x, float *enc,
                   size_t *cidx, float *cw)
{
    const int nc = 1 << dims;
    for (int lv = 0; lv < c->n_levels; ++lv) {
        const float s = l->scale[lv];
        const int res = l->res[lv];
        const size_t n_level = l->level_off[lv + 1] - l->level_off[lv];
        float pf[3] = { 0.0f, 0.0f, 0.0f };
        uint32_t pi[3] = { 0, 0, 0 };
        for (int d = 0; d < dims; ++d) {
            const float p = fmaf(s, x[d], 0.5f), fl = floorf(p);
            pf[d] = p - fl;
            pi[d] = (uint32_t)(int)fl;
        }
        float f[8] = {0};
        for (int k = 0; k < nc; ++k) {
            const uint32_t cx = pi[0] + (k & 1), cy = pi[1] + ((k >> 1) & 1), cz = pi[2] + ((k >> 2) & 1);
            float w = ((k & 1) ? pf[0] : 1.0f - pf[0]) * ((k & 2) ? pf[1] : 1.0f - pf[1]);
            size_t lin = (size_t)cx + (size_t)cy * (size_t)res;
            if (dims == 3) {
                w = w * ((k & 4) ? pf[2] : 1.0f - pf[2]);
                lin += (size_t)cz * (size_t)res * (size_t)res;
            }
            const size_t idx = lin % n_level;                                       /* dense grid_index */
            const float *g = grid + (l->level_off[lv] + idx) * c->n_features;
            for (int q = 0; q < c->n_features; ++q) f[q] += w * g[q];
            if (cidx) { cidx[nc * lv + k] = l->level_off[lv] + idx; cw[nc * lv + k] = w; }
        }
        for (int q = 0; q < c->n_features; ++q) enc[lv * c->n_features + q] = f[q];
    }
}

/* forward for n points; out: n x n_output_padded.  acts (optional): per point enc + hidden*n_neurons */
static int forward_d(const wo_net_config *c, int dims, const float *params, const float *xy, int n, float *out, float *acts)
{
    wn_layout l;
    layout_d(c, dims, &l);
    const float *grid = params + l.n_mlp;
    const int H = c->n_neurons, E = l.enc, NL = c->n_hidden_layers;
    const int act_stride = E + NL * H;
    float *buf = malloc(sizeof(float) * (size_t)act_stride);
    for (int p = 0; p < n; ++p) {
        float *a = acts ? acts + (size_t)p * act_stride : buf;
        encode(c, dims, &l, grid, xy + (size_t)dims * p, a, NULL, NULL);
        const float *W = params;
        const float *in = a;
        int n_in = E;
        for (int layer = 0; layer < NL; ++layer) {
            float *o = a + E + layer * H;
            for (int r = 0; r < H; ++r) {
                float s = 0.0f;
                for (int k = 0; k < n_in; ++k) s = fmaf(W[(size_t)r * n_in + k], in[k], s);
                o[r] = s > 0.0f ? s : 0.0f;
            }
            W += (size_t)H * n_in;
            in = o;
            n_in = H;
        }
        for (int r = 0; r < c->n_output_padded; ++r) {
            float s = 0.0f;
            for (int k = 0; k < H; ++k) s = fmaf(W[(size_t)r * H + k], in[k], s);
            out[(size_t)p * c->n_output_padded + r] = s;
        }
    }
    free(buf);
    return 0;
}
int wo_net_forward(const wo_net_config *c, const float *params, const float *xy, int n, float *out, float *acts)
{
    return forward_d(c, 2, params, xy, n, out, acts);
}
int wo_net3_forward(const wo_net_config *c, const float *params, const float *xyz, int n, float *out, float *acts)
{
    return forward_d(c, 3, params, xyz, n, out, acts);
}

/* Gradient of sum_p <dl_dout[p], out[p]> w.r.t. every parameter (grad must hold n_params floats).
 * A gradient is a sum over the points; to make it independent of the order in which a parallel
 * machine adds, every partial sum is converted to 64-bit fixed point (2^-36) and the integers are
 * added: grid terms one by one, weight terms as fmaf chains over chunks of 1024
 * consecutive points (four interleaved chains per chunk).  The HIP kernels do exactly the same, so both gradients are bit-identical. */
#define WN_FX_SCALE 68719476736.0   /* 2^36 */
#define WN_WGRAD_CHUNK 1024

static int backward_d(const wo_net_config *c, int dims, const float *params, const float *xy, const float *dl_dout, int n,
                      float *grad)
{
    wn_layout l;
    layout_d(c, dims, &l);
    const int ncorner = 1 << dims;
    const int H = c->n_neurons, E = l.enc, NL = c->n_hidden_layers, NO = c->n_output_padded;
    const int act_stride = E + NL * H;        /* encoding + hidden activations of one point */
    const int del_stride = NO + NL * H;       /* delta of the output layer, then of hidden layer 0.. */
    const size_t n_params = l.n_mlp + l.n_grid;
    long long *fx = calloc(n_params, sizeof(long long));
    float *acts = malloc(sizeof(float) * (size_t)WN_WGRAD_CHUNK * act_stride);
    float *dels = malloc(sizeof(float) * (size_t)WN_WGRAD_CHUNK * del_stride);
    float *denc = malloc(sizeof(float) * (size_t)E);
    size_t *cidx = malloc(sizeof(size_t) * 8 * c->n_levels);
    float *cw = malloc(sizeof(float) * 8 * c->n_levels);
    const float *grid = params + l.n_mlp;
    size_t woff[WN_MAX_LEVELS];
    woff[0] = 0;
    woff[1] = (size_t)H * E;
    for (int i = 2; i <= NL; ++i) woff[i] = woff[i - 1] + (size_t)H * H;
    for (int p0 = 0; p0 < n; p0 += WN_WGRAD_CHUNK) {
        const int cnt = n - p0 < WN_WGRAD_CHUNK ? n - p0 : WN_WGRAD_CHUNK;
        for (int q = 0; q < cnt; ++q) {
            const int p = p0 + q;
            float *a = acts + (size_t)q * act_stride;
            float *d = dels + (size_t)q * del_stride;
            encode(c, dims, &l, grid, xy + (size_t)dims * p, a, cidx, cw);
            /* forward, keeping activations */
            const float *in = a;
            int n_in = E;
            for (int layer = 0; layer < NL; ++layer) {
                const float *W = params + woff[layer];
                float *o = a + E + layer * H;
                for (int r = 0; r < H; ++r) {
                    float s = 0.0f;
                    for (int k = 0; k < n_in; ++k) s = fmaf(W[(size_t)r * n_in + k], in[k], s);
                    o[r] = s > 0.0f ? s : 0.0f;
                }
                in = o; n_in = H;
            }
            /* deltas: output layer, then back through the hidden layers (sums over r ascending) */
            for (int r = 0; r < NO; ++r) d[r] = dl_dout[(size_t)p * NO + r];
            const float *dcur = d;
            int n_o = NO;
            for (int layer = NL; layer >= 1; --layer) {
                const float *W = params + woff[layer];
                const float *h = a + E + (layer - 1) * H;
                float *dn = d + NO + (layer - 1) * H;
                for (int k = 0; k < H; ++k) {
                    float s = 0.0f;
                    for (int r = 0; r < n_o; ++r) s = fmaf(W[(size_t)r * H + k], dcur[r], s);
                    dn[k] = h[k] > 0.0f ? s : 0.0f;                      /* ReLU' */
                }
                dcur = dn; n_o = H;
            }
            for (int k = 0; k < E; ++k) {
                float s = 0.0f;
                const float *W = params + woff[0];
                for (int r = 0; r < H; ++r) s = fmaf(W[(size_t)r * E + k], dcur[r], s);
                denc[k] = s;
            }
            /* grid: every (corner, feature) term on its own */
            long long *gG = fx + l.n_mlp;
            for (int lv = 0; lv < c->n_levels; ++lv)
                for (int k = 0; k < ncorner; ++k)
                    for (int f = 0; f < c->n_features; ++f) {
                        const float t = cw[ncorner * lv + k] * denc[lv * c->n_features + f];
                        gG[cidx[ncorner * lv + k] * c->n_features + f] += llrint((double)t * WN_FX_SCALE);
                    }
        }
        /* weights: per (row, column) four fmaf chains over the 4-point groups of the chunk taken round
         * robin (group j goes to chain j mod 4: what the four waves of a block do), added in the
         * fixed order ((c0 + c1) + c2) + c3 */
        for (int layer = 0; layer <= NL; ++layer) {
            const int n_i = layer == 0 ? E : H, n_o = layer == NL ? NO : H;
            const int doff = layer == NL ? 0 : NO + layer * H;            /* delta of this layer's output */
            const int ioff = layer == 0 ? 0 : E + (layer - 1) * H;         /* this layer's input */
            long long *gW = fx + woff[layer];
            for (int r = 0; r < n_o; ++r)
                for (int k = 0; k < n_i; ++k) {
                    float c4[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                    for (int q = 0; q < cnt; ++q)
                        c4[(q >> 2) & 3] = fmaf(dels[(size_t)q * del_stride + doff + r], acts[(size_t)q * act_stride + ioff + k],
                                                c4[(q >> 2) & 3]);
                    const float s = ((c4[0] + c4[1]) + c4[2]) + c4[3];
                    if (s != 0.0f) gW[(size_t)r * n_i + k] += llrint((double)s * WN_FX_SCALE);
                }
        }
    }
    for (size_t i = 0; i < n_params; ++i) grad[i] = (float)((double)fx[i] / WN_FX_SCALE);
    free(fx); free(acts); free(dels); free(denc); free(cidx); free(cw);
    return 0;
}
int wo_net_backward(const wo_net_config *c, const float *params, const float *xy, const float *dl_dout, int n, float *grad)
{
    return backward_d(c, 2, params, xy, dl_dout, n, grad);
}
int wo_net3_backward(const wo_net_config *c, const float *params, const float *xyz, const float *dl_dout, int n, float *grad)
{
    return backward_d(c, 3, params, xyz, dl_dout, n, grad);
}

/* one optimizer step: tiny-cuda-nn's adam_step (include/tiny-cuda-nn/optimizers/adam.h, v1.6/v1.7:
 * pinned commit unknown) nested in its EMA optimizer (optimizers/ema.h), the pair the
 * reference configures in data/ladybug/n.json:69-81.  Restated from the published kernel:
 *   - gradient = grad / loss_scale;
 *   - a NON-matrix parameter (the grid encoding, i >= n_matrix_weights) whose gradient is exactly 0
 *     is left alone: no moment decay, no step, no step count;
 *   - l2_reg * w is added for matrix weights only;
 *   - every parameter debiases with its own step counter param_steps[i] (matrix weights move on
 *     every step, so theirs equals `step`);
 *   - the EMA (ema_step) runs over all parameters with the global step.
 * `step` counts from 1; param_steps is n_params zero-initialised counters owned by the caller. */
static int optimizer_step_d(const wo_net_config *c, int dims, float *params, float *m1, float *m2, float *ema_raw,
                            float *inference_params, const float *grad, int step, float loss_scale, uint32_t *param_steps)
{
    wn_layout l;
    layout_d(c, dims, &l);
    const uint64_t n = l.n_mlp + l.n_grid;
    const float debias = 1.0f / (1.0f - powf(c->ema_decay, (float)step));
    for (uint64_t i = 0; i < n; ++i) {
        const float w = params[i];
        float g = grad[i] / loss_scale;
        float nw = w;
        if (i < l.n_mlp || g != 0.0f) {
            if (i < l.n_mlp) g += c->l2_reg * w;
            const float a = m1[i] = c->beta1 * m1[i] + (1.0f - c->beta1) * g;
            const float b = m2[i] = c->beta2 * m2[i] + (1.0f - c->beta2) * (g * g);
            const uint32_t s = ++param_steps[i];
            const float lr = c->learning_rate * sqrtf(1.0f - powf(c->beta2, (float)s)) / (1.0f - powf(c->beta1, (float)s));
            nw = w - (lr / (sqrtf(b) + c->epsilon)) * a;
            params[i] = nw;
        }
        ema_raw[i] = c->ema_decay * ema_raw[i] + (1.0f - c->ema_decay) * nw;
        inference_params[i] = ema_raw[i] * debias;
    }
    return 0;
}
int wo_net_optimizer_step(const wo_net_config *c, float *params, float *m1, float *m2, float *ema_raw,
                          float *inference_params, const float *grad, int step, float loss_scale, uint32_t *param_steps)
{
    return optimizer_step_d(c, 2, params, m1, m2, ema_raw, inference_params, grad, step, loss_scale, param_steps);
}
int wo_net3_optimizer_step(const wo_net_config *c, float *params, float *m1, float *m2, float *ema_raw,
                           float *inference_params, const float *grad, int step, float loss_scale, uint32_t *param_steps)
{
    return optimizer_step_d(c, 3, params, m1, m2, ema_raw, inference_params, grad, step, loss_scale, param_steps);
}

int wo_net_levels(const wo_net_config *c, int *res, float *scale)
{
    wn_layout l;
    layout_d(c, 2, &l);
    for (int i = 0; i < c->n_levels; ++i) { res[i] = l.res[i]; scale[i] = l.scale[i]; }
    return l.enc;
}
