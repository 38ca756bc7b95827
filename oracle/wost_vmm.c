/*
 * wost_vmm.c -- CPU oracle for the deterministic pieces of the GUIDED path's directional
 * distribution: polynomial log-Bessel, von Mises pdf and d/dkappa,
 * Best-Fisher rejection sampling in double precision, and the 8-lobe mixture VMM<2,8> built
 * from raw network outputs.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates (paths relative to /root/reference):
 *   util/vonmises.h:17-93 (coefficients, evalPoly, logModifiedBesselFn), :95-118
 *   (rejectionSample), :121-209 (VonMises), integrator/guided/distribution.h:19-45,136-198
 *   (VMFKernel<2>, VMM<2,N>), integrator/guided/train.h:50-79 (output activations),
 *   util/transformation.h:47-50 (frameFromTangent), core/sampler.h:74-85 (nextDouble).
 * Pinned by the reference's own known-answer constants (test/vonmises_test.cu:5-22,57-59,
 * 124-148, commented out there but numerically valid -- SURVEY.md section 4).
 * Transcendentals are libm here and ocml on the GPU: parity is within the 1e-5 relative
 * tolerance the reference's tests use, not bit-exact.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "wost_oracle.h"
#include "wost_internal.h"
#include "wost_detmath.h"

#define WV_2PI 6.28318530717958647693f
#define WV_LOG_2PI 1.83787706640934548356f   /* logf(2 pi) */
#define WV_PI_D 3.14159265358979323846

static const float COEF_SMALL[2][7] = {
    {1.0f, 3.5156229f, 3.0899424f, 1.2067492f, 0.2659732f, 0.360768e-1f, 0.45813e-2f},
    {0.5f, 0.87890594f, 0.51498869f, 0.15084934f, 0.2658733e-1f, 0.301532e-2f, 0.32411e-3f}};
static const float COEF_LARGE[2][9] = {
    {0.39894228f, 0.1328592e-1f, 0.225319e-2f, -0.157565e-2f, 0.916281e-2f, -0.2057706e-1f, 0.2635537e-1f,
     -0.1647633e-1f, 0.392377e-2f},
    {0.39894228f, -0.3988024e-1f, -0.362018e-2f, 0.163801e-2f, -0.1031555e-1f, 0.2282967e-1f, -0.2895312e-1f,
     0.1787654e-1f, -0.420059e-2f}};

float wo_eval_poly(float y, const float *coeff, int n)
{
    float ret = coeff[n - 1];
    for (int i = n - 2; i >= 0; --i) ret = coeff[i] + y * ret;
    return ret;
}

float wo_eval_poly_large0(float y) { return wo_eval_poly(y, COEF_LARGE[0], 9); }

float wo_log_bessel(float x, int order)
{
    float y = x / 3.75f;
    y *= y;
    float small = wo_eval_poly(y, COEF_SMALL[order], 7);
    if (order == 1) small = fabsf(x) * small;
    small = wo_logf(small);
    y = 3.75f / x;
    float large = x - 0.5f * wo_logf(x) + wo_logf(wo_eval_poly(y, COEF_LARGE[order], 9));
    return (x < 3.75) ? small : large;
}

float wo_vm_log_eval(float kappa, float cos_theta)
{
    float ret = kappa * cos_theta;
    return ret - WV_LOG_2PI - wo_log_bessel(kappa, 0);
}

float wo_vm_eval(float kappa, float cos_theta)
{
    if (kappa < 1e-3f) return 1.0f / WV_2PI;
    return wo_expf(wo_vm_log_eval(kappa, cos_theta));
}

float wo_vm_dlog_dkappa(float kappa, float cosTheta)
{
    if (kappa < 3.75f) {
        const float *coeff = COEF_SMALL[0];
        const float coef = 0.0711111111111111f, c142 = 0.142222222222222f, c010 = 0.0101135802469136f;
        float kappa2 = kappa * kappa;
        float term7 = coeff[6] * kappa2;
        float term6 = coeff[5] + coef * term7;
        float term5 = coeff[4] + coef * kappa2 * term6;
        float term4 = coeff[3] + coef * kappa2 * term5;
        float term3 = coeff[2] + coef * kappa2 * term4;
        float term2 = coeff[1] + coef * kappa2 * term3;
        float numerator = coef * kappa2 * (coef * kappa2 * (coef * kappa2 * (coef * kappa2 * (c010 * coeff[6] * kappa * kappa2 + c142 * kappa * term6) + c142 * kappa * term5) + c142 * kappa * term4) + c142 * kappa * term3) + c142 * kappa * term2;
        float denominator = coeff[0] + coef * kappa2 * term2;
        return cosTheta - (numerator / denominator);
    } else {
        /* vonmises.h:153-161: the analytic derivative of the large-argument branch; the
         * reference spells the nested Horner form out in double constants -- same value: */
        const float *K = COEF_LARGE[0];
        double x = kappa, t = 3.75 / x;
        /* P(t) = sum K_i t^i, dP/dx = -(t/x) * sum i K_i t^(i-1) */
        double P = 0.0, dP = 0.0;
        for (int i = 8; i >= 0; --i) P = K[i] + t * P;
        for (int i = 8; i >= 1; --i) dP = i * (double)K[i] + t * dP;
        dP *= -(t / x);
        /* d/dx [x - 0.5 log x + log P] = 1 - 0.5/x + dP/P ; d log p = cos - that */
        return (float)(cosTheta - 1.0 - dP / P + 0.5 / x);
    }
}

double wo_vm_proposal_r(float kappa)
{
    double k = kappa;
    double tau = 1.0 + sqrt(1.0 + 4.0 * k * k);
    double rho = (tau - sqrt(2.0 * tau)) / (2.0 * k);
    double proposalR = (1.0 + rho * rho) / (2.0 * rho);
    double proposalRTaylor = 1.0 / k + k;
    return (kappa < 1e-5) ? proposalRTaylor : proposalR;
}

float wo_vm_rejection_sample(float kappa, double proposal_r, wo_pcg *rng)
{
    if (kappa < 1e-3f) return WV_2PI * wo_pcg_next_float(rng);
    for (;;) {
        double u1 = wo_pcg_next_double(rng);
        double u2 = wo_pcg_next_double(rng);
        double u3 = wo_pcg_next_double(rng);
        double z = wo_cospi_d(u1);
        double f = (1.0 + proposal_r * z) / (proposal_r + z);
        double c = (double)kappa * (proposal_r - f);
        int accept = ((c * (2.0 - c) - u2) > 0.0) || (wo_log_d(c / u2) + 1.0 - c >= 0.0);
        if (accept)
        {
            /* fmod(a + pi, 2 pi) - pi for a in [-pi, pi], spelled out (fmod is exact) */
            double a = copysign(1.0, u3 - 0.5) * wo_acos_d(f) + WV_PI_D;
            if (a >= 2 * WV_PI_D) a -= 2 * WV_PI_D;
            return (float)(a - WV_PI_D);
        }
    }
}

/* ---- VMM<2,8> ------------------------------------------------------------------------- */

static float clampf(float v, float lo, float hi) { return fmaxf(fminf(v, hi), lo); }

void wo_vmm_build(wv_vmm *m, const float *data)
{
    m->total = 0.0f;
    for (int i = 0; i < WV_NCOMP; ++i) {
        const float *d = data + 4 * i;
        m->sg[i].lambda = wo_expf(clampf(d[0], -10.0f, 15.0f));   /* train.h:60-72, Exponential */
        m->sg[i].kappa = wo_expf(clampf(d[1], -10.0f, 15.0f));
        m->sg[i].ox = d[2]; m->sg[i].oy = d[3];                 /* None */
        /* Eigen normalized() (Eigen/src/Core/Dot.h; ext/eigen is an empty submodule, .gitmodules): v / sqrt(z)
         * when z = squaredNorm > 0, else v unchanged */
        const float z = d[2] * d[2] + d[3] * d[3], n = sqrtf(z);
        m->sg[i].mux = z > 0.0f ? d[2] / n : d[2]; m->sg[i].muy = z > 0.0f ? d[3] / n : d[3];
        m->total += m->sg[i].lambda;
    }
    for (int i = 0; i < WV_NCOMP; ++i) m->weight[i] = m->sg[i].lambda / m->total;
}

float wo_vmm_pdf(const wv_vmm *m, float wx, float wy)
{
    float val = 0.0f;
    for (int i = 0; i < WV_NCOMP; ++i)
        val += m->weight[i] * wo_vm_eval(m->sg[i].kappa, wx * m->sg[i].mux + wy * m->sg[i].muy);
    return val;
}

static void lobe_sample(const wv_lobe *l, wo_pcg *rng, float *ox, float *oy)
{
    float theta = wo_vm_rejection_sample(l->kappa, wo_vm_proposal_r(l->kappa), rng);
    float vx, vy;
    wo_sincosf(theta, &vx, &vy);
    /* frameFromTangent(mu): N = normalize(-mu.y, mu.x), T = mu; world = T*v.x + N*v.y */
    float px = -l->muy, py = l->mux;
    const float pz = px * px + py * py, pl = sqrtf(pz);
    if (pz > 0.0f) { px /= pl; py /= pl; }             /* Eigen normalized(): a zero vector stays zero */
    *ox = l->mux * vx + px * vy;
    *oy = l->muy * vx + py * vy;
}

/* VMM::sample (distribution.h:186-198): pick a lobe by one float draw, then sample it */
void wo_vmm_sample(const wv_vmm *m, wo_pcg *rng, float *ox, float *oy)
{
    float u = wo_pcg_next_float(rng);
    int pick = 0, found = 0;
    for (int k = 0; k < WV_NCOMP && !found; ++k) {
        if (u < m->weight[k]) { pick = k; found = 1; }
        else u -= m->weight[k];
    }
    lobe_sample(&m->sg[pick], rng, ox, oy);
}

int wo_vonmises_eval(const float *kappa, const float *cos_theta, int n, float *log_i0, float *log_i1,
                     float *log_pdf, float *dlog_dkappa)
{
    for (int i = 0; i < n; ++i) {
        if (log_i0) log_i0[i] = wo_log_bessel(kappa[i], 0);
        if (log_i1) log_i1[i] = wo_log_bessel(kappa[i], 1);
        if (log_pdf) log_pdf[i] = wo_vm_log_eval(kappa[i], cos_theta[i]);
        if (dlog_dkappa) dlog_dkappa[i] = wo_vm_dlog_dkappa(kappa[i], cos_theta[i]);
    }
    return 0;
}

int wo_vonmises_sample(const float *kappa, const uint64_t *seed, int n, int per_point, float *theta)
{
    for (int i = 0; i < n; ++i) {
        wo_pcg rng;
        wo_pcg_set_seed(&rng, seed[i], 1);
        double pr = wo_vm_proposal_r(kappa[i]);
        for (int k = 0; k < per_point; ++k) theta[(size_t)i * per_point + k] = wo_vm_rejection_sample(kappa[i], pr, &rng);
    }
    return 0;
}

int wo_vmm_pdf_sample(const float *raw, const float *wi, const uint64_t *seed, int n, float *pdf, float *dir)
{
    for (int i = 0; i < n; ++i) {
        wv_vmm m;
        wo_vmm_build(&m, raw + 32 * (size_t)i);
        if (pdf) pdf[i] = wo_vmm_pdf(&m, wi[2 * i], wi[2 * i + 1]);
        if (dir) {
            wo_pcg rng;
            wo_pcg_set_seed(&rng, seed[i], 1);
            wo_vmm_sample(&m, &rng, &dir[2 * i], &dir[2 * i + 1]);
        }
    }
    return 0;
}

/* ---- training-side gradient of the mixture -----------------------
 * restates integrator/guided/distribution.h:201-264 (VMM<2,N>::gradients_probability),
 * integrator/guided/train.h:81-105 (d_network_to_d_params) and :492-553
 * (compute_dL_doutput_divergence).  `raw`: 33 floats per sample (8 lobes x (lambda, kappa,
 * mu.x, mu.y) + selection logit); reference record per sample: dir[2], Li, dirPdf, onNeumann,
 * normal[2].  Outputs: dL/draw (33 per sample) and the per-sample likelihood term. */
#define WV_EPS 1e-5f     /* M_EPSILON, core/math/include/krrmath/constants.h */

static float wv_d_exp_act(float v) { return wo_expf(clampf(v, -10.0f, 15.0f)); }

int wo_vmm_loss_gradients(const float *raw, const float *dir, const float *li, const float *dir_pdf,
                          const unsigned char *on_neumann, const float *normal, int n, float loss_scale,
                          float *dl_draw, float *likelihood)
{
    const float scale = loss_scale / (float)n;                /* train.h:512 */
    for (int t = 0; t < n; ++t) {
        const float *data = raw + 33 * (size_t)t;
        float *grad = dl_draw + 33 * (size_t)t;
        wv_vmm m;
        wo_vmm_build(&m, data);
        const float wx = dir[2 * t], wy = dir[2 * t + 1];
        const int on_n = on_neumann ? on_neumann[t] : 0;
        float rx = 0.0f, ry = 0.0f;
        if (on_n) {   /* reflect(wi, n) = wi - 2 (wi.n) n, util/transformation.h:69-72 */
            const float nx = normal[2 * t], ny = normal[2 * t + 1];
            const float d = wx * nx + wy * ny;
            rx = wx - 2 * d * nx; ry = wy - 2 * d * ny;
        }
        float pdf_k[WV_NCOMP], pdf_kr[WV_NCOMP];
        for (int k = 0; k < WV_NCOMP; ++k) {
            pdf_k[k] = wo_vm_eval(m.sg[k].kappa, wx * m.sg[k].mux + wy * m.sg[k].muy);
            pdf_kr[k] = on_n ? wo_vm_eval(m.sg[k].kappa, rx * m.sg[k].mux + ry * m.sg[k].muy) : 0.0f;
        }
        float probability = 0.0f;
        for (int sg = 0; sg < WV_NCOMP; ++sg) {
            const float lambda = m.sg[sg].lambda, kappa = m.sg[sg].kappa;
            const float mox = m.sg[sg].ox, moy = m.sg[sg].oy;
            const float vm = pdf_k[sg];
            probability += m.weight[sg] * vm;
            float vmr = 0.0f;
            if (on_n) { vmr = pdf_kr[sg]; probability += m.weight[sg] * vmr; }
            float dF_dlambda = (vm + vmr) * (m.total - lambda) / (m.total * m.total);
            for (int k = 0; k < WV_NCOMP; ++k) {
                if (k == sg) continue;
                dF_dlambda -= m.weight[k] / m.total * pdf_k[k];
                if (on_n) dF_dlambda -= m.weight[k] / m.total * pdf_kr[k];
            }
            /* d pdf / d kappa = pdf * d log pdf / d kappa (vonmises.h:165-168) */
            float dF_dkappa = m.weight[sg] * (vm * wo_vm_dlog_dkappa(kappa, wx * m.sg[sg].mux + wy * m.sg[sg].muy));
            if (on_n) dF_dkappa += m.weight[sg] * (vmr * wo_vm_dlog_dkappa(kappa, rx * m.sg[sg].mux + ry * m.sg[sg].muy));
            const float n2 = mox * mox + moy * moy;
            float denom = n2 * sqrtf(n2);           /* |mu|^3 (the reference's powf(., 1.5f)) */
            if (denom < WV_EPS) denom = WV_EPS;
            float dF_dx = m.weight[sg] * vm * kappa * moy * (-mox * wy + moy * wx) / denom;
            if (on_n) dF_dx += m.weight[sg] * vmr * kappa * moy * (-mox * ry + moy * rx) / denom;
            float dF_dy = m.weight[sg] * vm * kappa * mox * (mox * wy - moy * wx) / denom;
            if (on_n) dF_dy += m.weight[sg] * vmr * kappa * mox * (mox * ry - moy * rx) / denom;
            grad[4 * sg + 0] = dF_dlambda; grad[4 * sg + 1] = dF_dkappa;
            grad[4 * sg + 2] = dF_dx; grad[4 * sg + 3] = dF_dy;
        }
        const float Li = li[t];
        const float dirPdf = dir_pdf[t] + WV_EPS;
        const float guidePdf = probability + WV_EPS;
        const float prefix = -Li / dirPdf / guidePdf * scale;
        if (likelihood) likelihood[t] = -Li / dirPdf * wo_logf(guidePdf);
        for (int sg = 0; sg < WV_NCOMP; ++sg) {
            grad[4 * sg + 0] = prefix * grad[4 * sg + 0] * wv_d_exp_act(data[4 * sg + 0]);
            grad[4 * sg + 1] = prefix * grad[4 * sg + 1] * wv_d_exp_act(data[4 * sg + 1]);
            grad[4 * sg + 2] = prefix * grad[4 * sg + 2];
            grad[4 * sg + 3] = prefix * grad[4 * sg + 3];
        }
        /* selection probability (train.h:541-552): Logistic activation */
        const float e = 0.2f;
        const float uni = on_n ? (float)(1.0 / WV_PI_D) : 1.0f / WV_2PI;
        const float sgm = 1.0f / (1.0f + wo_expf(-data[32]));
        grad[32] = scale * (-e) * Li * (guidePdf - uni) / (dirPdf * dirPdf) * (sgm * (1 - sgm));
    }
    return 0;
}
