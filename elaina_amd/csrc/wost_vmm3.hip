// wost_vmm3.hip -- the directional distribution of the 3-D guided path as batch entry points of the C-ABI (wost3_vmf_*,
// wost3_vmm_*: include/wost.h): von Mises-Fisher lobes and the mixture VMM<3,8> with its loss gradients, against which the CPU
// restatement is compared value by value (tests/test_vmf.py, tests/test_guided_3d.py).  gfx950 only.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/wost.h"
#include "wost_internal3.h"
#include "wost_vmm3_device.h"

namespace wost {

__global__ __launch_bounds__(256) void vmf_eval_kernel(const float *kappa, const float *cos_theta, int n, float *pdf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pdf[i] = vmf_eval(kappa[i], cos_theta[i]);
}

__global__ __launch_bounds__(256) void vmf_sample_kernel(const float *kappa, const float *mu, const uint64_t *seed, int n, int per_point, float *dirs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pcg rng{0, 1};
    pcg_set_seed(rng, seed[i], 1);
    const V3 m = v3(mu[3 * i], mu[3 * i + 1], mu[3 * i + 2]);
    for (int k = 0; k < per_point; ++k) {
        const V3 w = vmf_sample(kappa[i], m, rng);
        float *o = dirs + 3 * ((size_t)i * per_point + k);
        o[0] = w.x; o[1] = w.y; o[2] = w.z;
    }
}

__global__ __launch_bounds__(256) void vmm3_pdf_sample_kernel(const float *raw, const float *wi, const uint64_t *seed, int n, float *pdf, float *dir)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Vmm3 m;
    vmm3_build(m, raw + 40 * (size_t)i);
    if (pdf) pdf[i] = vmm3_pdf(m, v3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]));
    if (dir) {
        Pcg rng{0, 1};
        pcg_set_seed(rng, seed[i], 1);
        const V3 w = vmm3_sample(m, rng);
        dir[3 * i] = w.x; dir[3 * i + 1] = w.y; dir[3 * i + 2] = w.z;
    }
}

// compute_dL_doutput_divergence with GuidedOutput = common3d around VMM<3,N>::gradients_probability
__global__ __launch_bounds__(256) void vmm3_loss_gradients_kernel(const float *raw, const float *dir, const float *li, const float *dir_pdf,
                                                                  const unsigned char *on_neumann, const float *normal, int n, float loss_scale,
                                                                  float *dl_draw, float *likelihood)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float eps = 1e-5f;
    const float scale = loss_scale / (float)n;
    const float *data = raw + 41 * (size_t)t;
    float *grad = dl_draw + 41 * (size_t)t;
    Vmm3 m;
    vmm3_build(m, data);
    const V3 w = v3(dir[3 * t], dir[3 * t + 1], dir[3 * t + 2]);
    const bool on_n = on_neumann ? on_neumann[t] != 0 : false;
    V3 r = v3(0.0f, 0.0f, 0.0f);
    if (on_n) {
        const V3 nn = v3(normal[3 * t], normal[3 * t + 1], normal[3 * t + 2]);
        const float d = (w.x * nn.x + w.y * nn.y) + w.z * nn.z;
        r = v3(w.x - 2 * d * nn.x, w.y - 2 * d * nn.y, w.z - 2 * d * nn.z);
    }
    float pk[kVmm3Lobes], pkr[kVmm3Lobes];
#pragma unroll
    for (int k = 0; k < kVmm3Lobes; ++k) {
        pk[k] = vmm3_lobe_pdf(m, k, w);
        pkr[k] = on_n ? vmm3_lobe_pdf(m, k, r) : 0.0f;
    }
    float probability = 0.0f;
    float g5[kVmm3Lobes][5];
#pragma unroll
    for (int sg = 0; sg < kVmm3Lobes; ++sg) {
        const float lambda = m.lambda[sg], kappa = m.kappa[sg];
        const float ox = m.mo[sg].x, oy = m.mo[sg].y, oz = m.mo[sg].z;
        const V3 mu = m.mu[sg];
        const float vmf = pk[sg];
        probability += m.weight[sg] * vmf;
        float vmfr = 0.0f;
        if (on_n) { vmfr = pkr[sg]; probability += m.weight[sg] * vmfr; }
        float dF_dlambda = (vmf + vmfr) * (m.total - lambda) / (m.total * m.total);
#pragma unroll
        for (int k = 0; k < kVmm3Lobes; ++k) {
            if (k == sg) continue;
            dF_dlambda -= m.weight[k] / m.total * pk[k];
            if (on_n) dF_dlambda -= m.weight[k] / m.total * pkr[k];
        }
        float ik;
        if (kappa < 1) ik = 0.000962f + -0.344883f * kappa + 0.030147f * (kappa * kappa);
        else ik = 1 / kappa - (1 + det_expf(-2 * kappa)) / (1 - det_expf(-2 * kappa));
        float dF_dkappa = m.weight[sg] * vmf * ((w.x * mu.x + w.y * mu.y + w.z * mu.z) + ik);
        if (on_n) dF_dkappa += m.weight[sg] * vmfr * ((r.x * mu.x + r.y * mu.y + r.z * mu.z) + ik);
        const float n2 = (ox * ox + oy * oy) + oz * oz;
        float denom = n2 * sqrtf(n2);
        if (denom < eps) denom = eps;
        const float x = w.x, y = w.y, z = w.z, xr = r.x, yr = r.y, zr = r.z;
        float dF_dx = m.weight[sg] * vmf * kappa * (-ox * oy * y - ox * oz * z + (oy * oy) * x + (oz * oz) * x) / denom;
        if (on_n) dF_dx += m.weight[sg] * vmfr * kappa * (-ox * oy * yr - ox * oz * zr + (oy * oy) * xr + (oz * oz) * xr) / denom;
        float dF_dy = m.weight[sg] * vmf * kappa * (-ox * oy * x - oy * oz * z + (ox * ox) * y + (oz * oz) * y) / denom;
        if (on_n) dF_dy += m.weight[sg] * vmfr * kappa * (-ox * oy * xr - oy * oz * zr + (ox * ox) * yr + (oz * oz) * yr) / denom;
        float dF_dz = m.weight[sg] * vmf * kappa * (-ox * oz * x - oy * oz * y + (ox * ox) * z + (oy * oy) * z) / denom;
        if (on_n) dF_dz += m.weight[sg] * vmfr * kappa * (-ox * oz * xr - oy * oz * yr + (ox * ox) * zr + (oy * oy) * zr) / denom;
        g5[sg][0] = dF_dlambda; g5[sg][1] = dF_dkappa; g5[sg][2] = dF_dx; g5[sg][3] = dF_dy; g5[sg][4] = dF_dz;
    }
    const float Li = li[t];
    const float dirPdf = dir_pdf[t] + eps;
    const float guidePdf = probability + eps;
    const float prefix = -Li / dirPdf / guidePdf * scale;
    if (likelihood) likelihood[t] = -Li / dirPdf * det_logf(guidePdf);
#pragma unroll
    for (int sg = 0; sg < kVmm3Lobes; ++sg) {
        grad[5 * sg + 0] = prefix * g5[sg][0] * det_expf(clamp_act(data[5 * sg + 0]));
        grad[5 * sg + 1] = prefix * g5[sg][1] * det_expf(clamp_act(data[5 * sg + 1]));
        grad[5 * sg + 2] = prefix * g5[sg][2];
        grad[5 * sg + 3] = prefix * g5[sg][3];
        grad[5 * sg + 4] = prefix * g5[sg][4];
    }
    const float e = 0.2f;
    const float uni = on_n ? 1.0f / WOST_2PI : 1.0f / WOST_4PI;
    const float sgm = 1.0f / (1.0f + det_expf(-data[40]));
    grad[40] = scale * (-e) * Li * (guidePdf - uni) / (dirPdf * dirPdf) * (sgm * (1 - sgm));
}

void launch_vmm3_loss_gradients(hipStream_t stream, const float *raw, const float *dir, const float *li, const float *dir_pdf,
                                const unsigned char *on_neumann, const float *normal, int n, float loss_scale, float *dl_draw, float *likelihood)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(vmm3_loss_gradients_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, raw, dir, li, dir_pdf, on_neumann, normal, n,
                       loss_scale, dl_draw, likelihood);
}

}  // namespace wost

using namespace wost;

extern "C" {

int wost3_vmf_eval(int device, const float *kappa, const float *cos_theta, int32_t n, float *pdf)
{
    if (!kappa || !cos_theta || !pdf || n < 0) return set_error(WOST_ERR_INVALID, "null argument");
    if (n == 0) return WOST_OK;
    W3_TRY(hipSetDevice(device));
    Scratch3 s;
    float *d_k, *d_c, *d_p;
    W3_TRY(s.alloc(&d_k, n)); W3_TRY(s.alloc(&d_c, n)); W3_TRY(s.alloc(&d_p, n));
    W3_TRY(hipMemcpy(d_k, kappa, (size_t)n * 4, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_c, cos_theta, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(vmf_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_k, d_c, n, d_p);
    W3_TRY(hipGetLastError());
    W3_TRY(hipMemcpy(pdf, d_p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return WOST_OK;
}

int wost3_vmf_sample(int device, const float *kappa, const float *mu, const uint64_t *seed, int32_t n, int32_t per_point, float *dirs)
{
    if (!kappa || !mu || !seed || !dirs || n < 0 || per_point < 1) return set_error(WOST_ERR_INVALID, "null argument");
    if (n == 0) return WOST_OK;
    W3_TRY(hipSetDevice(device));
    Scratch3 s;
    float *d_k, *d_m, *d_o;
    uint64_t *d_s;
    W3_TRY(s.alloc(&d_k, n)); W3_TRY(s.alloc(&d_m, (size_t)n * 3)); W3_TRY(s.alloc(&d_s, n)); W3_TRY(s.alloc(&d_o, (size_t)n * per_point * 3));
    W3_TRY(hipMemcpy(d_k, kappa, (size_t)n * 4, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_m, mu, (size_t)n * 12, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_s, seed, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(vmf_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_k, d_m, d_s, n, per_point, d_o);
    W3_TRY(hipGetLastError());
    W3_TRY(hipMemcpy(dirs, d_o, (size_t)n * per_point * 12, hipMemcpyDeviceToHost));
    return WOST_OK;
}

int wost3_vmm_pdf_sample(int device, const float *raw, const float *wi, const uint64_t *seed, int32_t n, float *pdf, float *sample_dir)
{
    if (!raw || !wi || (sample_dir && !seed) || n < 0) return set_error(WOST_ERR_INVALID, "null argument");
    if (n == 0) return WOST_OK;
    W3_TRY(hipSetDevice(device));
    Scratch3 s;
    float *d_r, *d_w, *d_p = nullptr, *d_d = nullptr;
    uint64_t *d_s = nullptr;
    W3_TRY(s.alloc(&d_r, (size_t)n * 40)); W3_TRY(s.alloc(&d_w, (size_t)n * 3));
    W3_TRY(hipMemcpy(d_r, raw, (size_t)n * 160, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_w, wi, (size_t)n * 12, hipMemcpyHostToDevice));
    if (pdf) W3_TRY(s.alloc(&d_p, n));
    if (sample_dir) {
        W3_TRY(s.alloc(&d_d, (size_t)n * 3)); W3_TRY(s.alloc(&d_s, n));
        W3_TRY(hipMemcpy(d_s, seed, (size_t)n * 8, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(vmm3_pdf_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_r, d_w, d_s, n, d_p, d_d);
    W3_TRY(hipGetLastError());
    if (pdf) W3_TRY(hipMemcpy(pdf, d_p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (sample_dir) W3_TRY(hipMemcpy(sample_dir, d_d, (size_t)n * 12, hipMemcpyDeviceToHost));
    return WOST_OK;
}

int wost3_vmm_loss_gradients(int device, const float *raw, const float *dir, const float *li, const float *dir_pdf, const uint8_t *on_neumann,
                             const float *normal, int32_t n, float loss_scale, float *dl_draw, float *likelihood)
{
    if (!raw || !dir || !li || !dir_pdf || !dl_draw || (on_neumann && !normal) || n < 0) return set_error(WOST_ERR_INVALID, "null argument");
    if (n == 0) return WOST_OK;
    W3_TRY(hipSetDevice(device));
    Scratch3 s;
    float *d_r, *d_d, *d_l, *d_p, *d_n = nullptr, *d_g, *d_k = nullptr;
    unsigned char *d_o = nullptr;
    W3_TRY(s.alloc(&d_r, (size_t)n * 41)); W3_TRY(s.alloc(&d_d, (size_t)n * 3)); W3_TRY(s.alloc(&d_l, n)); W3_TRY(s.alloc(&d_p, n));
    W3_TRY(s.alloc(&d_g, (size_t)n * 41));
    W3_TRY(hipMemcpy(d_r, raw, (size_t)n * 164, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_d, dir, (size_t)n * 12, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_l, li, (size_t)n * 4, hipMemcpyHostToDevice));
    W3_TRY(hipMemcpy(d_p, dir_pdf, (size_t)n * 4, hipMemcpyHostToDevice));
    if (on_neumann) {
        W3_TRY(s.alloc(&d_o, n)); W3_TRY(s.alloc(&d_n, (size_t)n * 3));
        W3_TRY(hipMemcpy(d_o, on_neumann, (size_t)n, hipMemcpyHostToDevice));
        W3_TRY(hipMemcpy(d_n, normal, (size_t)n * 12, hipMemcpyHostToDevice));
    }
    if (likelihood) W3_TRY(s.alloc(&d_k, n));
    hipLaunchKernelGGL(vmm3_loss_gradients_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_r, d_d, d_l, d_p, d_o, d_n, n, loss_scale, d_g, d_k);
    W3_TRY(hipGetLastError());
    W3_TRY(hipMemcpy(dl_draw, d_g, (size_t)n * 164, hipMemcpyDeviceToHost));
    if (likelihood) W3_TRY(hipMemcpy(likelihood, d_k, (size_t)n * 4, hipMemcpyDeviceToHost));
    return WOST_OK;
}

}  // extern "C"
