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
#include "wost_math.h"

namespace wost {

int set_error(int code, const std::string &msg);

__constant__ float VM_COEF_SMALL[2][7] = {
    {1.0f, 3.5156229f, 3.0899424f, 1.2067492f, 0.2659732f, 0.360768e-1f, 0.45813e-2f},
    {0.5f, 0.87890594f, 0.51498869f, 0.15084934f, 0.2658733e-1f, 0.301532e-2f, 0.32411e-3f}};
__constant__ float VM_COEF_LARGE[2][9] = {
    {0.39894228f, 0.1328592e-1f, 0.225319e-2f, -0.157565e-2f, 0.916281e-2f, -0.2057706e-1f, 0.2635537e-1f,
     -0.1647633e-1f, 0.392377e-2f},
    {0.39894228f, -0.3988024e-1f, -0.362018e-2f, 0.163801e-2f, -0.1031555e-1f, 0.2282967e-1f, -0.2895312e-1f,
     0.1787654e-1f, -0.420059e-2f}};

#define VM_2PI 6.28318530717958647693f
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
    small = logf(small);
    y = 3.75f / x;
    const float large = x - 0.5f * logf(x) + logf(eval_poly(y, VM_COEF_LARGE[order], 9));
    return (x < 3.75f) ? small : large;
}

__device__ __forceinline__ float vm_log_eval(float kappa, float cos_theta)
{
    const float ret = kappa * cos_theta;
    return ret - logf(VM_2PI) - log_bessel(kappa, 0);
}

__device__ __forceinline__ float vm_eval(float kappa, float cos_theta)
{
    if (kappa < 1e-3f) return 1.0f / VM_2PI;
    return expf(vm_log_eval(kappa, cos_theta));
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
        const double z = cos(VM_PI_D * u1);
        const double f = (1.0 + proposal_r * z) / (proposal_r + z);
        const double c = (double)kappa * (proposal_r - f);
        const bool accept = ((c * (2.0 - c) - u2) > 0.0) || (log(c / u2) + 1.0 - c >= 0.0);
        if (accept) return (float)(fmod((copysign(1.0, u3 - 0.5) * acos(f)) + VM_PI_D, 2 * VM_PI_D) - VM_PI_D);
    }
}

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

// VMM<2,8>: lambda = exp(clamp(x,-10,15)), kappa likewise, mu = normalize(x,y), weights lambda/sum
__global__ void vmm_pdf_sample_kernel(const float *raw, const float *wi, const uint64_t *seed, int n, float *pdf,
                                      float *dir)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *d = raw + 32 * (size_t)i;
    float lambda[8], kap[8], mux[8], muy[8];
    float total = 0.0f;
    for (int k = 0; k < 8; ++k) {
        lambda[k] = expf(fmaxf(fminf(d[4 * k], 15.0f), -10.0f));
        kap[k] = expf(fmaxf(fminf(d[4 * k + 1], 15.0f), -10.0f));
        const float x = d[4 * k + 2], y = d[4 * k + 3];
        const float nn = sqrtf(x * x + y * y);
        mux[k] = x / nn;
        muy[k] = y / nn;
        total += lambda[k];
    }
    if (pdf) {
        const float wx = wi[2 * i], wy = wi[2 * i + 1];
        float val = 0.0f;
        for (int k = 0; k < 8; ++k) val += (lambda[k] / total) * vm_eval(kap[k], wx * mux[k] + wy * muy[k]);
        pdf[i] = val;
    }
    if (dir) {
        Pcg rng;
        pcg_set_seed(rng, seed[i], 1);
        float u = pcg_next_float(rng);
        int pick = 0;
        bool found = false;
        for (int k = 0; k < 8; ++k) {
            const float w = lambda[k] / total;
            if (!found) {
                if (u < w) { pick = k; found = true; }
                else u -= w;
            }
        }
        float pk = kap[0], pmx = mux[0], pmy = muy[0];
        for (int k = 1; k < 8; ++k)
            if (k == pick) { pk = kap[k]; pmx = mux[k]; pmy = muy[k]; }
        const float theta = vm_rejection_sample(pk, vm_proposal_r(pk), rng);
        const float vx = cosf(theta), vy = sinf(theta);
        float px = -pmy, py = pmx;   // frameFromTangent(mu): N = normalize(-mu.y, mu.x), T = mu
        const float pl = sqrtf(px * px + py * py);
        px /= pl; py /= pl;
        dir[2 * i] = pmx * vx + px * vy;
        dir[2 * i + 1] = pmy * vx + py * vy;
    }
}

// training-side gradient of the mixture and of the selection logit: reference
// integrator/guided/distribution.h:201-264 (gradients_probability) + train.h:492-553
// (compute_dL_doutput_divergence), one thread per training sample.
__global__ void vmm_loss_gradients_kernel(const float *raw, const float *dir, const float *li, const float *dir_pdf,
                                          const uint8_t *on_neumann, const float *normal, int n, float loss_scale,
                                          float *dl_draw, float *likelihood)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float eps = 1e-5f;  // M_EPSILON
    const float scale = loss_scale / (float)n;
    const float *d = raw + 33 * (size_t)t;
    float *grad = dl_draw + 33 * (size_t)t;
    float lambda[8], kap[8], mux[8], muy[8], ox[8], oy[8], pk[8], pkr[8];
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
    for (int k = 0; k < 8; ++k) {
        lambda[k] = expf(fmaxf(fminf(d[4 * k], 15.0f), -10.0f));
        kap[k] = expf(fmaxf(fminf(d[4 * k + 1], 15.0f), -10.0f));
        ox[k] = d[4 * k + 2];
        oy[k] = d[4 * k + 3];
        const float nn = sqrtf(ox[k] * ox[k] + oy[k] * oy[k]);
        mux[k] = ox[k] / nn;
        muy[k] = oy[k] / nn;
        total += lambda[k];
    }
    for (int k = 0; k < 8; ++k) {
        pk[k] = vm_eval(kap[k], wx * mux[k] + wy * muy[k]);
        pkr[k] = on_n ? vm_eval(kap[k], rx * mux[k] + ry * muy[k]) : 0.0f;
    }
    float probability = 0.0f;
    for (int sg = 0; sg < 8; ++sg) {
        const float w = lambda[sg] / total;
        const float vm = pk[sg], vmr = pkr[sg];
        probability += w * vm;
        if (on_n) probability += w * vmr;
        float dF_dlambda = (vm + vmr) * (total - lambda[sg]) / (total * total);
        for (int k = 0; k < 8; ++k) {
            if (k == sg) continue;
            const float wk = lambda[k] / total;
            dF_dlambda -= wk / total * pk[k];
            if (on_n) dF_dlambda -= wk / total * pkr[k];
        }
        float dF_dkappa = w * (vm * vm_dlog_dkappa(kap[sg], wx * mux[sg] + wy * muy[sg]));
        if (on_n) dF_dkappa += w * (vmr * vm_dlog_dkappa(kap[sg], rx * mux[sg] + ry * muy[sg]));
        float denom = powf(ox[sg] * ox[sg] + oy[sg] * oy[sg], 1.5f);
        if (denom < eps) denom = eps;
        float dF_dx = w * vm * kap[sg] * oy[sg] * (-ox[sg] * wy + oy[sg] * wx) / denom;
        if (on_n) dF_dx += w * vmr * kap[sg] * oy[sg] * (-ox[sg] * ry + oy[sg] * rx) / denom;
        float dF_dy = w * vm * kap[sg] * ox[sg] * (ox[sg] * wy - oy[sg] * wx) / denom;
        if (on_n) dF_dy += w * vmr * kap[sg] * ox[sg] * (ox[sg] * ry - oy[sg] * rx) / denom;
        grad[4 * sg + 0] = dF_dlambda;
        grad[4 * sg + 1] = dF_dkappa;
        grad[4 * sg + 2] = dF_dx;
        grad[4 * sg + 3] = dF_dy;
    }
    const float Li = li[t];
    const float dirPdf = dir_pdf[t] + eps;
    const float guidePdf = probability + eps;
    const float prefix = -Li / dirPdf / guidePdf * scale;
    if (likelihood) likelihood[t] = -Li / dirPdf * logf(guidePdf);
    for (int sg = 0; sg < 8; ++sg) {
        grad[4 * sg + 0] = prefix * grad[4 * sg + 0] * expf(fmaxf(fminf(d[4 * sg], 15.0f), -10.0f));
        grad[4 * sg + 1] = prefix * grad[4 * sg + 1] * expf(fmaxf(fminf(d[4 * sg + 1], 15.0f), -10.0f));
        grad[4 * sg + 2] = prefix * grad[4 * sg + 2];
        grad[4 * sg + 3] = prefix * grad[4 * sg + 3];
    }
    const float uni = on_n ? (float)(1.0 / VM_PI_D) : 1.0f / VM_2PI;
    const float sgm = 1.0f / (1.0f + expf(-d[32]));
    grad[32] = scale * (-0.2f) * Li * (guidePdf - uni) / (dirPdf * dirPdf) * (sgm * (1 - sgm));
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
    hipLaunchKernelGGL(vmm_loss_gradients_kernel, dim3((n + 127) / 128), dim3(128), 0, 0, dr, dd, dli, dp, don, dn, n,
                       loss_scale, dg, dl);
    VMM_TRY(hipGetLastError());
    VMM_TRY(hipMemcpy(dl_draw, dg, (size_t)n * 33 * 4, hipMemcpyDeviceToHost));
    if (likelihood) VMM_TRY(hipMemcpy(likelihood, dl, (size_t)n * 4, hipMemcpyDeviceToHost));
    return WOST_OK;
}

}  // extern "C"
