// wost_vmm3_device.h -- device functions of the 3-D guided path's directional distribution: the von Mises-Fisher lobe
// (reference util/vmf.h:21-70) and the eight-lobe mixture VMM<3,8> (integrator/guided/distribution.h:279-436), shared by the
// batch entry points (wost_vmm3.hip) and the guided 3-D kernels (wost_guided3.hip).  gfx950 only.
#pragma once
#include "wost_device3.h"

namespace wost {

// ---- von Mises-Fisher lobe on the sphere (reference util/vmf.h:21-70, the Jakob [2012] form; the lobes of
// GuidedIntegrator<3>'s mixture -- that integrator is not built, the distribution is its first piece) ----------------
constexpr float kVmfEpsilon = 1e-5f;   // M_EPSILON, core/math/include/krrmath/constants.h

// VMF::eval(cosTheta)
__device__ __forceinline__ float vmf_eval(float kappa, float cos_theta)
{
    if (kappa < kVmfEpsilon) return 1.0f / WOST_4PI;
    return det_expf(kappa * fminf(0.0f, cos_theta - 1.0f)) * kappa / (WOST_2PI * (1.0f - det_expf(-2.0f * kappa)));
}

// VMF::sample(sampler, mu): two draws, the lobe about +z turned into the frame of mu
__device__ __forceinline__ V3 vmf_sample(float kappa, V3 mu, Pcg &rng)
{
    const float u0 = pcg_next_float(rng), u1 = pcg_next_float(rng);
    float c, s;
    sincos_2pi(u1, c, s);
    V3 local;
    if (kappa < kVmfEpsilon) {
        const float z = 1 - 2 * u0, r = sqrtf(1 - z * z);               // uniformSampleSphere<3>
        local = v3(r * c, r * s, z);
    } else {
        const float cos_theta = 1.0f + det_logf(1.0f + (-u0 + det_expf(-2.0f * kappa) * u0)) / kappa;
        const float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
        local = v3(c * sin_theta, s * sin_theta, cos_theta);
    }
    return frame_to_world(mu, local.x, local.y, local.z);
}

// ---- VMM<3,8>: the mixture of eight vMF lobes (reference integrator/guided/distribution.h:279-436, train.h:60-105 and
// 492-553 with common3d: 5 numbers per lobe -- lambda, kappa, mean vector -- and the selection logit = 41 outputs) ------
constexpr int kVmm3Lobes = 8;
struct Vmm3 {
    float lambda[kVmm3Lobes], kappa[kVmm3Lobes], weight[kVmm3Lobes], total;
    V3 mu[kVmm3Lobes], mo[kVmm3Lobes];
};

__device__ __forceinline__ float clamp_act(float v) { return fmaxf(fminf(v, 15.0f), -10.0f); }

__device__ __forceinline__ void vmm3_build(Vmm3 &m, const float *data)
{
    m.total = 0.0f;
#pragma unroll
    for (int i = 0; i < kVmm3Lobes; ++i) {
        const float *d = data + 5 * i;
        m.lambda[i] = det_expf(clamp_act(d[0]));
        m.kappa[i] = det_expf(clamp_act(d[1]));
        // Eigen normalized(): v / sqrt(z) when z = squaredNorm > 0, else v unchanged
        const float z = (d[2] * d[2] + d[3] * d[3]) + d[4] * d[4], n = sqrtf(z);
        m.mo[i] = v3(d[2], d[3], d[4]);
        m.mu[i] = z > 0.0f ? v3(d[2] / n, d[3] / n, d[4] / n) : m.mo[i];
        m.total += m.lambda[i];
    }
#pragma unroll
    for (int i = 0; i < kVmm3Lobes; ++i) m.weight[i] = m.lambda[i] / m.total;
}

__device__ __forceinline__ float vmm3_lobe_pdf(const Vmm3 &m, int i, V3 w)
{
    return vmf_eval(m.kappa[i], (w.x * m.mu[i].x + w.y * m.mu[i].y) + w.z * m.mu[i].z);
}

__device__ __forceinline__ float vmm3_pdf(const Vmm3 &m, V3 w)
{
    float val = 0.0f;
#pragma unroll
    for (int i = 0; i < kVmm3Lobes; ++i) val += m.weight[i] * vmm3_lobe_pdf(m, i, w);
    return val;
}

// one draw picks the lobe, two more the direction
__device__ __forceinline__ V3 vmm3_sample(const Vmm3 &m, Pcg &rng)
{
    float u = pcg_next_float(rng);
    int pick = 0;
    bool done = false;
#pragma unroll
    for (int i = 0; i < kVmm3Lobes; ++i) {
        if (!done && u < m.weight[i]) { pick = i; done = true; }
        if (!done) u -= m.weight[i];
    }
    float kap = m.kappa[0];
    V3 mu = m.mu[0];
#pragma unroll
    for (int i = 1; i < kVmm3Lobes; ++i)
        if (pick == i) { kap = m.kappa[i]; mu = m.mu[i]; }
    return vmf_sample(kap, mu, rng);
}

}  // namespace wost
