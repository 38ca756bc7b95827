// wost_vmm_device.h -- device functions of the guided path's directional distribution
// (SURVEY.md 8a row a24): polynomial log-Bessel, von Mises pdf and d/dkappa (reference
// util/vonmises.h:17-93,121-172), Best-Fisher rejection sampling in double precision (:95-118)
// and the 8-lobe mixture VMM<2,8> built from raw network outputs
// (integrator/guided/distribution.h:136-198, train.h:50-79).  Shared by the batch entry points
// (wost_vmm.hip) and the guided integrator kernels (wost_guided.hip).  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include "wost_math.h"

namespace wost {

static __constant__ float VM_COEF_SMALL[2][7] = {
    {1.0f, 3.5156229f, 3.0899424f, 1.2067492f, 0.2659732f, 0.360768e-1f, 0.45813e-2f},
    {0.5f, 0.87890594f, 0.51498869f, 0.15084934f, 0.2658733e-1f, 0.301532e-2f, 0.32411e-3f}};
static __constant__ float VM_COEF_LARGE[2][9] = {
    {0.39894228f, 0.1328592e-1f, 0.225319e-2f, -0.157565e-2f, 0.916281e-2f, -0.2057706e-1f, 0.2635537e-1f,
     -0.1647633e-1f, 0.392377e-2f},
    {0.39894228f, -0.3988024e-1f, -0.362018e-2f, 0.163801e-2f, -0.1031555e-1f, 0.2282967e-1f, -0.2895312e-1f,
     0.1787654e-1f, -0.420059e-2f}};

#define VM_2PI 6.28318530717958647693f
#define VM_LOG_2PI 1.83787706640934548356f   // logf(2 pi)
#define VM_PI_D 3.14159265358979323846

__device__ __forceinline__ float eval_poly(float y, const float *coeff, int n)
{
    float ret = coeff[n - 1];
    for (int i = n - 2; i >= 0; --i) ret = coeff[i] + y * ret;
    return ret;
}

__device__ __forceinline__ float log_bessel(float x, int order)
{
    float y = x / 3.75f;
    y *= y;
    float small = eval_poly(y, VM_COEF_SMALL[order], 7);
    if (order == 1) small = fabsf(x) * small;
    small = det_logf(small);
    if (x < 3.75f) return small;      // the asymptotic branch only where it is selected
    y = 3.75f / x;
    return x - 0.5f * det_logf(x) + det_logf(eval_poly(y, VM_COEF_LARGE[order], 9));
}

__device__ __forceinline__ float vm_log_eval(float kappa, float cos_theta)
{
    const float ret = kappa * cos_theta;
    return ret - VM_LOG_2PI - log_bessel(kappa, 0);
}

__device__ __forceinline__ float vm_eval(float kappa, float cos_theta)
{
    if (kappa < 1e-3f) return 1.0f / VM_2PI;
    return det_expf(vm_log_eval(kappa, cos_theta));
}

// vm_eval with log I0(kappa) supplied by the caller (it depends on the lobe only)
__device__ __forceinline__ float vm_eval_lb(float kappa, float lb, float cos_theta)
{
    if (kappa < 1e-3f) return 1.0f / VM_2PI;
    const float ret = kappa * cos_theta;
    return det_expf(ret - VM_LOG_2PI - lb);
}

__device__ __forceinline__ float vm_dlog_dkappa(float kappa, float cosTheta)
{
    if (kappa < 3.75f) {
        const float *coeff = VM_COEF_SMALL[0];
        const float coef = 0.0711111111111111f, c142 = 0.142222222222222f, c010 = 0.0101135802469136f;
        const float kappa2 = kappa * kappa;
        const float term7 = coeff[6] * kappa2;
        const float term6 = coeff[5] + coef * term7;
        const float term5 = coeff[4] + coef * kappa2 * term6;
        const float term4 = coeff[3] + coef * kappa2 * term5;
        const float term3 = coeff[2] + coef * kappa2 * term4;
        const float term2 = coeff[1] + coef * kappa2 * term3;
        const float numerator = coef * kappa2 * (coef * kappa2 * (coef * kappa2 * (coef * kappa2 * (c010 * coeff[6] * kappa * kappa2 + c142 * kappa * term6) + c142 * kappa * term5) + c142 * kappa * term4) + c142 * kappa * term3) + c142 * kappa * term2;
        const float denominator = coeff[0] + coef * kappa2 * term2;
        return cosTheta - (numerator / denominator);
    }
    // large-argument branch: d/dx [x - log(x)/2 + log P(3.75/x)], evaluated in double like the
    // reference's spelled-out expression (its 3.75 literals are doubles)
    const float *K = VM_COEF_LARGE[0];
    const double x = kappa, t = 3.75 / x;
    double P = 0.0, dP = 0.0;
    for (int i = 8; i >= 0; --i) P = K[i] + t * P;
    for (int i = 8; i >= 1; --i) dP = i * (double)K[i] + t * dP;
    dP *= -(t / x);
    return (float)(cosTheta - 1.0 - dP / P + 0.5 / x);
}

__device__ __forceinline__ double vm_proposal_r(float kappa)
{
    const double k = kappa;
    const double tau = 1.0 + sqrt(1.0 + 4.0 * k * k);
    const double rho = (tau - sqrt(2.0 * tau)) / (2.0 * k);
    const double proposalR = (1.0 + rho * rho) / (2.0 * rho);
    const double proposalRTaylor = 1.0 / k + k;
    return (kappa < 1e-5) ? proposalRTaylor : proposalR;
}

__device__ __forceinline__ double pcg_next_double(Pcg &r)
{
    // reference core/sampler.h:74-85
    const uint64_t u = ((uint64_t)pcg_next_uint(r) << 20) | 0x3ff0000000000000ULL;
    return __longlong_as_double((long long)u) - 1.0;
}

__device__ __forceinline__ float vm_rejection_sample(float kappa, double proposal_r, Pcg &rng)
{
    if (kappa < 1e-3f) return VM_2PI * pcg_next_float(rng);
    for (;;) {
        const double u1 = pcg_next_double(rng);
        const double u2 = pcg_next_double(rng);
        const double u3 = pcg_next_double(rng);
        const double z = det_cospi_d(u1);
        const double f = (1.0 + proposal_r * z) / (proposal_r + z);
        const double c = (double)kappa * (proposal_r - f);
        const bool accept = ((c * (2.0 - c) - u2) > 0.0) || (det_log_d(c / u2) + 1.0 - c >= 0.0);
        if (accept) {
            // fmod(a + pi, 2 pi) - pi for a in [-pi, pi], spelled out (fmod is exact)
            double a = copysign(1.0, u3 - 0.5) * det_acos_d(f) + VM_PI_D;
            if (a >= 2 * VM_PI_D) a -= 2 * VM_PI_D;
            return (float)(a - VM_PI_D);
        }
    }
}

// one lobe of the mixture from its four raw network outputs (train.h:60-79, distribution.h:150-165): everything that
// does not depend on the other lobes.  A function of its own so that the lanes of a wave can prepare different lobes of
// the same point (guided_sample_kernel) -- the arithmetic per lobe is the same wherever it runs.
struct VmmLobe {
    float lambda, kappa, mux, muy, lb;
};
__device__ __forceinline__ VmmLobe vmm_lobe(float r0, float r1, float x, float y)
{
    VmmLobe l;
    l.lambda = det_expf(fmaxf(fminf(r0, 15.0f), -10.0f));
    l.kappa = det_expf(fmaxf(fminf(r1, 15.0f), -10.0f));
    // Eigen's normalized() (Eigen/src/Core/Dot.h, the library behind mu_original.normalized(),
    // distribution.h:160): v / sqrt(z) when z = |v|^2 > 0, else v unchanged -- a zero vector
    // stays zero instead of becoming NaN (half-precision outputs do underflow to exact zeros)
    const float z = x * x + y * y, nn = sqrtf(z);
    l.mux = z > 0.0f ? x / nn : x;
    l.muy = z > 0.0f ? y / nn : y;
    l.lb = log_bessel(l.kappa, 0);
    return l;
}

// VMM<2,8>: lambda = exp(clamp(x,-10,15)), kappa likewise, mu = normalize(x,y), weights lambda/sum
struct Vmm {
    // register vectors (constant indices after unrolling), not arrays: the struct must not end up in scratch
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    f32x8 weight, kap, mux, muy;
    f32x8 lb;       // log I0(kappa) of every lobe: pdf() is evaluated up to twice per build

    // d(j) = raw network output j of this point
    template <class F>
    __device__ __forceinline__ void build(F d)
    {
        f32x8 lambda;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const VmmLobe l = vmm_lobe(d(4 * k), d(4 * k + 1), d(4 * k + 2), d(4 * k + 3));
            lambda[k] = l.lambda; kap[k] = l.kappa; mux[k] = l.mux; muy[k] = l.muy; lb[k] = l.lb;
        }
        finish(lambda);
    }

    // the mixture from lobes that vmm_lobe has already prepared (kap, mux, muy, lb set by the caller): the weights
    __device__ __forceinline__ void finish(const f32x8 &lambda)
    {
        float total = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) total += lambda[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) weight[k] = lambda[k] / total;
    }

    __device__ __forceinline__ void build(const float *d)
    {
        build([d](int j) { return d[j]; });
    }

    __device__ __forceinline__ float pdf(float wx, float wy) const
    {
        float val = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) val += weight[k] * vm_eval_lb(kap[k], lb[k], wx * mux[k] + wy * muy[k]);
        return val;
    }

    // pdf(a) and, where `two` is set, pdf(b): one pass over the lobes
    __device__ __forceinline__ void pdf_pair(float ax, float ay, float bx, float by, bool two, float &pa, float &pb) const
    {
        pa = 0.0f;
        pb = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            pa += weight[k] * vm_eval_lb(kap[k], lb[k], ax * mux[k] + ay * muy[k]);
            if (two) pb += weight[k] * vm_eval_lb(kap[k], lb[k], bx * mux[k] + by * muy[k]);
        }
    }

    // VMM::sample (distribution.h:186-198): lobe picked by one float draw, then the lobe's
    // rejection sampler, rotated into the frame whose tangent is mu
    __device__ __forceinline__ void sample(Pcg &rng, float &ox, float &oy) const
    {
        float u = pcg_next_float(rng);
        int pick = 0;
        bool found = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (!found) {
                if (u < weight[k]) { pick = k; found = true; }
                else u -= weight[k];
            }
        }
        float pk = kap[0], pmx = mux[0], pmy = muy[0];
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k == pick) { pk = kap[k]; pmx = mux[k]; pmy = muy[k]; }
        const float theta = vm_rejection_sample(pk, vm_proposal_r(pk), rng);
        float vx, vy;
        det_sincosf(theta, &vx, &vy);
        float px = -pmy, py = pmx;   // frameFromTangent(mu): N = normalize(-mu.y, mu.x), T = mu
        const float pz = px * px + py * py, pl = sqrtf(pz);
        if (pz > 0.0f) { px /= pl; py /= pl; }      // Eigen normalized(): a zero vector stays zero
        ox = pmx * vx + px * vy;
        oy = pmy * vx + py * vy;
    }
};

// ---- the same mixture kept in a lane's LDS column instead of forty registers ------------------------------------------
// (guided_sample_kernel: the lanes of a 16-point unit prepare the lobes, four lanes per point, and leave them in the column
// of the walker's lane.)  Entries: weight[k] at k, kappa at 8 + k, log I0(kappa) at 16 + k, mean at 24 + k / 32 + k, the
// selection logit at 40.  Rolled loops over the lobes, one lobe's five values live at a time; the arithmetic and its order
// are Vmm's (pdf sums ascend in k, the pick subtracts the weights in turn), so the results are the same bits.
constexpr int kVmmColWords = 41;

template <class COL>
__device__ __forceinline__ float vmm_col(const COL &c, int e)
{
    return __uint_as_float(c.get(e));
}

template <class COL>
__device__ __forceinline__ void vmm_col_pdf_pair(const COL &c, float ax, float ay, float bx, float by, bool two, float &pa, float &pb)
{
    pa = 0.0f;
    pb = 0.0f;
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
        const float w = vmm_col(c, k), kap = vmm_col(c, 8 + k), lb = vmm_col(c, 16 + k), mx = vmm_col(c, 24 + k), my = vmm_col(c, 32 + k);
        pa += w * vm_eval_lb(kap, lb, ax * mx + ay * my);
        if (two) pb += w * vm_eval_lb(kap, lb, bx * mx + by * my);
    }
}

template <class COL>
__device__ __forceinline__ void vmm_col_sample(const COL &c, Pcg &rng, float &ox, float &oy)
{
    float u = pcg_next_float(rng);
    int pick = 0;
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
        const float w = vmm_col(c, k);
        if (u < w) { pick = k; break; }
        u -= w;
    }
    const float pk = vmm_col(c, 8 + pick), pmx = vmm_col(c, 24 + pick), pmy = vmm_col(c, 32 + pick);
    const float theta = vm_rejection_sample(pk, vm_proposal_r(pk), rng);
    float vx, vy;
    det_sincosf(theta, &vx, &vy);
    float px = -pmy, py = pmx;   // frameFromTangent(mu): N = normalize(-mu.y, mu.x), T = mu
    const float pz = px * px + py * py, pl = sqrtf(pz);
    if (pz > 0.0f) { px /= pl; py /= pl; }      // Eigen normalized(): a zero vector stays zero
    ox = pmx * vx + px * vy;
    oy = pmy * vx + py * vy;
}

}  // namespace wost
