// wost_vmm.hip -- deterministic pieces of the GUIDED path's directional distribution on the
// device, exposed as batch entry points of the C-ABI so that they can be
// pinned against the reference's known-answer constants before the guided integrator itself
// is built: polynomial log-Bessel, von Mises log-pdf and d/dkappa (reference util/vonmises.h:17-93,
// 121-172), Best-Fisher rejection sampling in double precision with three 53-bit draws per trial
// (:95-118), and the 8-lobe mixture VMM<2,8> assembled from raw network outputs
// (integrator/guided/distribution.h:136-198, train.h:50-79).  gfx950 only.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/wost.h"
#include "wost_internal.h"
#include "wost_math.h"
#include "wost_vmm_device.h"

namespace wost {

__global__ void vonmises_eval_kernel(const float *kappa, const float *cos_theta, int n, float *log_i0, float *log_i1,
                                     float *log_pdf, float *dlog)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (log_i0) log_i0[i] = log_bessel(kappa[i], 0);
    if (log_i1) log_i1[i] = log_bessel(kappa[i], 1);
    if (log_pdf) log_pdf[i] = vm_log_eval(kappa[i], cos_theta[i]);
    if (dlog) dlog[i] = vm_dlog_dkappa(kappa[i], cos_theta[i]);
}

__global__ void vonmises_sample_kernel(const float *kappa, const uint64_t *seed, int n, int per_point, float *theta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pcg rng;
    pcg_set_seed(rng, seed[i], 1);
    const double pr = vm_proposal_r(kappa[i]);
    for (int k = 0; k < per_point; ++k) theta[(size_t)i * per_point + k] = vm_rejection_sample(kappa[i], pr, rng);
}

// VMM<2,8> pdf and one sample per point (stream setSeed(seed, 1))
__global__ void vmm_pdf_sample_kernel(const float *raw, const float *wi, const uint64_t *seed, int n, float *pdf,
                                      float *dir)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Vmm m;
    m.build(raw + 32 * (size_t)i);
    if (pdf) pdf[i] = m.pdf(wi[2 * i], wi[2 * i + 1]);
    if (dir) {
        Pcg rng;
        pcg_set_seed(rng, seed[i], 1);
        m.sample(rng, dir[2 * i], dir[2 * i + 1]);
    }
}

// training-side gradient of the mixture and of the selection logit: reference
// integrator/guided/distribution.h:201-264 (gradients_probability) + train.h:492-553
// (compute_dL_doutput_divergence), one thread per training sample.
constexpr int kLossBlock = 128;
__global__ void vmm_loss_gradients_kernel(const float *raw, const float *dir, const float *li, const float *dir_pdf,
                                          const uint8_t *on_neumann, const float *normal, int n, float loss_scale,
                                          float *dl_draw, float *likelihood)
{
    // The rows of a block's samples are 33 floats each, back to back: read by their owners (lane t row t, 132 bytes apart) every
    // load instruction touched 64 cache lines; the block moves its 128 x 33 floats as ONE contiguous piece through LDS instead
    // (row stride 33 words: odd, so the owners' reads and writes there are conflict-free) -- 59 -> 3x us per 524 288-sample batch
    __shared__ float sh[kLossBlock * 33];
    const int base = blockIdx.x * kLossBlock;
    const int t = base + (int)threadIdx.x;
    const int n_here = min(kLossBlock, n - base);
    for (int k = threadIdx.x; k < n_here * 33; k += kLossBlock) sh[k] = raw[33 * (size_t)base + k];
    __syncthreads();
    const bool valid = t < n;
    float grad[33];
#pragma unroll
    for (int j = 0; j < 33; ++j) grad[j] = 0.0f;
    if (valid) {
    const float eps = 1e-5f;  // M_EPSILON
    const float scale = loss_scale / (float)n;
    float lambda[8], kap[8], mux[8], muy[8], ox[8], oy[8], pk[8], pkr[8];
    float rawv[33];
#pragma unroll
    for (int j = 0; j < 33; ++j) rawv[j] = sh[33 * threadIdx.x + j];
    float total = 0.0f;
    const float wx = dir[2 * t], wy = dir[2 * t + 1];
    const bool on_n = on_neumann && on_neumann[t] != 0;
    float rx = 0.0f, ry = 0.0f;
    if (on_n) {
        const float nx = normal[2 * t], ny = normal[2 * t + 1];
        const float dd = wx * nx + wy * ny;
        rx = wx - 2 * dd * nx;
        ry = wy - 2 * dd * ny;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        lambda[k] = det_expf(fmaxf(fminf(rawv[4 * k], 15.0f), -10.0f));
        kap[k] = det_expf(fmaxf(fminf(rawv[4 * k + 1], 15.0f), -10.0f));
        ox[k] = rawv[4 * k + 2];
        oy[k] = rawv[4 * k + 3];
        const float z = ox[k] * ox[k] + oy[k] * oy[k], nn = sqrtf(z);
        mux[k] = z > 0.0f ? ox[k] / nn : ox[k];      // Eigen normalized(): a zero vector stays zero
        muy[k] = z > 0.0f ? oy[k] / nn : oy[k];
        total += lambda[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float lb = log_bessel(kap[k], 0);       // shared by the direction and its mirror image
        pk[k] = vm_eval_lb(kap[k], lb, wx * mux[k] + wy * muy[k]);
        pkr[k] = on_n ? vm_eval_lb(kap[k], lb, rx * mux[k] + ry * muy[k]) : 0.0f;
    }
    // the unscaled partial derivatives stay in registers; one pass over dL/draw at the end
    float g_lambda[8], g_kappa[8], g_x[8], g_y[8];
    float probability = 0.0f;
#pragma unroll
    for (int sg = 0; sg < 8; ++sg) {
        const float w = lambda[sg] / total;
        const float vm = pk[sg], vmr = pkr[sg];
        probability += w * vm;
        if (on_n) probability += w * vmr;
        float dF_dlambda = (vm + vmr) * (total - lambda[sg]) / (total * total);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k == sg) continue;
            const float wk = lambda[k] / total;
            dF_dlambda -= wk / total * pk[k];
            if (on_n) dF_dlambda -= wk / total * pkr[k];
        }
        float dF_dkappa = w * (vm * vm_dlog_dkappa(kap[sg], wx * mux[sg] + wy * muy[sg]));
        if (on_n) dF_dkappa += w * (vmr * vm_dlog_dkappa(kap[sg], rx * mux[sg] + ry * muy[sg]));
        const float n2 = ox[sg] * ox[sg] + oy[sg] * oy[sg];
        float denom = n2 * sqrtf(n2);                 // |mu|^3 (the reference's powf(., 1.5f))
        if (denom < eps) denom = eps;
        float dF_dx = w * vm * kap[sg] * oy[sg] * (-ox[sg] * wy + oy[sg] * wx) / denom;
        if (on_n) dF_dx += w * vmr * kap[sg] * oy[sg] * (-ox[sg] * ry + oy[sg] * rx) / denom;
        float dF_dy = w * vm * kap[sg] * ox[sg] * (ox[sg] * wy - oy[sg] * wx) / denom;
        if (on_n) dF_dy += w * vmr * kap[sg] * ox[sg] * (ox[sg] * ry - oy[sg] * rx) / denom;
        g_lambda[sg] = dF_dlambda;
        g_kappa[sg] = dF_dkappa;
        g_x[sg] = dF_dx;
        g_y[sg] = dF_dy;
    }
    const float Li = li[t];
    const float dirPdf = dir_pdf[t] + eps;
    const float guidePdf = probability + eps;
    const float prefix = -Li / dirPdf / guidePdf * scale;
    if (likelihood) likelihood[t] = -Li / dirPdf * det_logf(guidePdf);
#pragma unroll
    for (int sg = 0; sg < 8; ++sg) {
        // d exp(clamp(x)) / dx as the reference takes it = exp(clamp(x)) = the lobe's own lambda / kappa
        grad[4 * sg + 0] = prefix * g_lambda[sg] * lambda[sg];
        grad[4 * sg + 1] = prefix * g_kappa[sg] * kap[sg];
        grad[4 * sg + 2] = prefix * g_x[sg];
        grad[4 * sg + 3] = prefix * g_y[sg];
    }
    const float uni = on_n ? (float)(1.0 / VM_PI_D) : 1.0f / VM_2PI;
    const float sgm = 1.0f / (1.0f + det_expf(-rawv[32]));
    grad[32] = scale * (-0.2f) * Li * (guidePdf - uni) / (dirPdf * dirPdf) * (sgm * (1 - sgm));
    }
    __syncthreads();      // every owner has read its row
    if (valid) {
#pragma unroll
        for (int j = 0; j < 33; ++j) sh[33 * threadIdx.x + j] = grad[j];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n_here * 33; k += kLossBlock) dl_draw[33 * (size_t)base + k] = sh[k];
}

void launch_vmm_loss_gradients(hipStream_t stream, const float *raw, const float *dir, const float *li,
                               const float *dir_pdf, const uint8_t *on_neumann, const float *normal, int n,
                               float loss_scale, float *dl_draw, float *likelihood)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(vmm_loss_gradients_kernel, dim3((n + kLossBlock - 1) / kLossBlock), dim3(kLossBlock), 0, stream, raw, dir, li, dir_pdf,
                       on_neumann, normal, n, loss_scale, dl_draw, likelihood);
}

struct DevBufs {
    std::vector<void *> ptrs;
    ~DevBufs()
    {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <class T> hipError_t in(T **d, const T *h, size_t count)
    {
        *d = nullptr;
        if (!h) return hipSuccess;
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, count * sizeof(T) + 16);
        if (e != hipSuccess) return e;
        ptrs.push_back(q);
        *d = reinterpret_cast<T *>(q);
        return hipMemcpy(q, h, count * sizeof(T), hipMemcpyHostToDevice);
    }
    template <class T> hipError_t out(T **d, const T *h, size_t count)
    {
        *d = nullptr;
        if (!h) return hipSuccess;
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, count * sizeof(T) + 16);
        if (e != hipSuccess) return e;
        ptrs.push_back(q);
        *d = reinterpret_cast<T *>(q);
        return hipSuccess;
    }
};

#define VMM_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return set_error(WOST_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

static int pick_device(int device)
{
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return set_error(WOST_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n_dev) return set_error(WOST_ERR_INVALID, "device index out of range");
    if (hipSetDevice(device) != hipSuccess) return set_error(WOST_ERR_DEVICE, "hipSetDevice failed");
    return WOST_OK;
}

}  // namespace wost

using namespace wost;

extern "C" {

int wost_vonmises_eval(int device, const float *kappa, const float *cos_theta, int32_t n, float *log_i0, float *log_i1,
                       float *log_pdf, float *dlogpdf_dkappa)
{
    if (!kappa || !cos_theta || n < 0) return set_error(WOST_ERR_INVALID, "null argument");
    if (n == 0) return WOST_OK;
    int rc = pick_device(device);
    if (rc != WOST_OK) return rc;
    DevBufs b;
    float *dk, *dc, *o0, *o1, *o2, *o3;
    VMM_TRY(b.in(&dk, kappa, n));
    VMM_TRY(b.in(&dc, cos_theta, n));
    VMM_TRY(b.out(&o0, log_i0, n));
    VMM_TRY(b.out(&o1, log_i1, n));
    VMM_TRY(b.out(&o2, log_pdf, n));
    VMM_TRY(b.out(&o3, dlogpdf_dkappa, n));
    hipLaunchKernelGGL(vonmises_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dk, dc, n, o0, o1, o2, o3);
    VMM_TRY(hipGetLastError());
    if (log_i0) VMM_TRY(hipMemcpy(log_i0, o0, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (log_i1) VMM_TRY(hipMemcpy(log_i1, o1, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (log_pdf) VMM_TRY(hipMemcpy(log_pdf, o2, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (dlogpdf_dkappa) VMM_TRY(hipMemcpy(dlogpdf_dkappa, o3, (size_t)n * 4, hipMemcpyDeviceToHost));
    return WOST_OK;
}

int wost_vonmises_sample(int device, const float *kappa, const uint64_t *seed, int32_t n, int32_t per_point, float *theta)
{
    if (!kappa || !seed || !theta || n < 0 || per_point < 1) return set_error(WOST_ERR_INVALID, "bad argument");
    if (n == 0) return WOST_OK;
    int rc = pick_device(device);
    if (rc != WOST_OK) return rc;
    DevBufs b;
    float *dk, *dt;
    uint64_t *ds;
    VMM_TRY(b.in(&dk, kappa, n));
    VMM_TRY(b.in(&ds, seed, n));
    VMM_TRY(b.out(&dt, theta, (size_t)n * per_point));
    hipLaunchKernelGGL(vonmises_sample_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, dk, ds, n, per_point, dt);
    VMM_TRY(hipGetLastError());
    VMM_TRY(hipMemcpy(theta, dt, (size_t)n * per_point * 4, hipMemcpyDeviceToHost));
    return WOST_OK;
}

int wost_vmm_pdf_sample(int device, const float *raw, const float *wi, const uint64_t *seed, int32_t n, float *pdf,
                        float *sample_dir)
{
    if (!raw || n < 0 || (pdf && !wi) || (sample_dir && !seed)) return set_error(WOST_ERR_INVALID, "bad argument");
    if (n == 0) return WOST_OK;
    int rc = pick_device(device);
    if (rc != WOST_OK) return rc;
    DevBufs b;
    float *dr, *dw, *dp, *dd;
    uint64_t *ds;
    VMM_TRY(b.in(&dr, raw, (size_t)n * 32));
    VMM_TRY(b.in(&dw, wi, (size_t)n * 2));
    VMM_TRY(b.in(&ds, seed, n));
    VMM_TRY(b.out(&dp, pdf, n));
    VMM_TRY(b.out(&dd, sample_dir, (size_t)n * 2));
    hipLaunchKernelGGL(vmm_pdf_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dr, dw, ds, n, dp, dd);
    VMM_TRY(hipGetLastError());
    if (pdf) VMM_TRY(hipMemcpy(pdf, dp, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (sample_dir) VMM_TRY(hipMemcpy(sample_dir, dd, (size_t)n * 8, hipMemcpyDeviceToHost));
    return WOST_OK;
}

int wost_vmm_loss_gradients(int device, const float *raw, const float *dir, const float *li, const float *dir_pdf,
                            const uint8_t *on_neumann, const float *normal, int32_t n, float loss_scale, float *dl_draw,
                            float *likelihood)
{
    if (!raw || !dir || !li || !dir_pdf || !dl_draw || n < 0 || (on_neumann && !normal))
        return set_error(WOST_ERR_INVALID, "bad argument");
    if (n == 0) return WOST_OK;
    int rc = pick_device(device);
    if (rc != WOST_OK) return rc;
    DevBufs b;
    float *dr, *dd, *dli, *dp, *dn, *dg, *dl;
    uint8_t *don;
    VMM_TRY(b.in(&dr, raw, (size_t)n * 33));
    VMM_TRY(b.in(&dd, dir, (size_t)n * 2));
    VMM_TRY(b.in(&dli, li, n));
    VMM_TRY(b.in(&dp, dir_pdf, n));
    VMM_TRY(b.in(&don, on_neumann, n));
    VMM_TRY(b.in(&dn, normal, (size_t)n * 2));
    VMM_TRY(b.out(&dg, dl_draw, (size_t)n * 33));
    VMM_TRY(b.out(&dl, likelihood, n));
    hipLaunchKernelGGL(vmm_loss_gradients_kernel, dim3((n + kLossBlock - 1) / kLossBlock), dim3(kLossBlock), 0, 0, dr, dd, dli, dp, don, dn, n,
                       loss_scale, dg, dl);
    VMM_TRY(hipGetLastError());
    VMM_TRY(hipMemcpy(dl_draw, dg, (size_t)n * 33 * 4, hipMemcpyDeviceToHost));
    if (likelihood) VMM_TRY(hipMemcpy(likelihood, dl, (size_t)n * 4, hipMemcpyDeviceToHost));
    return WOST_OK;
}

}  // extern "C"
