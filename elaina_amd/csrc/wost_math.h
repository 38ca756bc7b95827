// wost_math.h -- device-side deterministic arithmetic of the Walk-on-Stars path.
//
// Written for gfx950 only (no host/CUDA dual path).  Every function here is specified
// operation by operation in DESIGN.md ("deterministic math"); the CPU oracle implements
// the same specification independently in C, and the parity tests compare the two bit
// for bit.  Build with -ffp-contract=off: fused multiply-adds appear only where written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wost {

#define WOST_2PI 6.28318530717958647693f
#define WOST_PCG32_MULT 0x5851f42d4c957f2dULL

__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by)
{
    return __builtin_fmaf(ax, bx, ay * by);
}
__device__ __forceinline__ float cross2(float ax, float ay, float bx, float by)
{
    return __builtin_fmaf(ax, by, -(ay * bx));
}

// ---- PCG32 (reference core/sampler.h:20-27,46-72,87-98) ---------------------------------
struct Pcg {
    uint64_t state;
    uint64_t inc;
};

__device__ __forceinline__ uint32_t pcg_next_uint(Pcg &r)
{
    uint64_t old = r.state;
    r.state = old * WOST_PCG32_MULT + r.inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

// two draws whose values are not needed: one LCG jump (x -> M^2 x + (M + 1) inc), which
// leaves the generator in exactly the state two pcg_next_uint() calls would
__device__ __forceinline__ void pcg_skip2(Pcg &r)
{
    const uint64_t m2 = WOST_PCG32_MULT * WOST_PCG32_MULT;
    r.state = r.state * m2 + (WOST_PCG32_MULT + 1ull) * r.inc;
}

__device__ __forceinline__ float pcg_next_float(Pcg &r)
{
    return __uint_as_float((pcg_next_uint(r) >> 9) | 0x3f800000u) - 1.0f;
}

__device__ __forceinline__ void pcg_set_seed(Pcg &r, uint64_t initstate, uint64_t initseq)
{
    r.state = 0U;
    r.inc = (initseq << 1u) | 1u;
    pcg_next_uint(r);
    r.state += initstate;
    pcg_next_uint(r);
}

__device__ __forceinline__ void pcg_advance(Pcg &r, int64_t delta)
{
    uint64_t cur_mult = WOST_PCG32_MULT, cur_plus = r.inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta /= 2;
    }
    r.state = acc_mult * r.state + acc_plus;
}

__device__ __forceinline__ uint32_t interleave_32bit(uint32_t vx, uint32_t vy)
{
    // reference util/hash.h:13-28
    uint32_t x = vx & 0x0000ffff;
    uint32_t y = vy & 0x0000ffff;
    x = (x | (x << 8)) & 0x00FF00FF;
    x = (x | (x << 4)) & 0x0F0F0F0F;
    x = (x | (x << 2)) & 0x33333333;
    x = (x | (x << 1)) & 0x55555555;
    y = (y | (y << 8)) & 0x00FF00FF;
    y = (y | (y << 4)) & 0x0F0F0F0F;
    y = (y | (y << 2)) & 0x33333333;
    y = (y | (y << 1)) & 0x55555555;
    return x | (y << 1);
}

// per-pixel seeding of prepareSolve (reference integrator/uniform/integrator.cu:71-77)
__device__ __forceinline__ void pcg_seed_pixel(Pcg &r, int pixel_id, int width)
{
    uint32_t px = (uint32_t)(pixel_id % width);
    uint32_t py = (uint32_t)(pixel_id / width);
    pcg_set_seed(r, (uint64_t)interleave_32bit(px, py), 0);
    int delta = 256 * pixel_id;
    pcg_advance(r, (int64_t)delta);
}

// ---- cos/sin(2*pi*u), u in [0,1) ---------------------------------------------------------
// Octant reduction is exact for u = k * 2^-23; sin/cos(pi/4 * g) by Taylor polynomials in
// g with fp32-rounded coefficients, Horner with fma; branch-free octant fix-up.
__device__ __forceinline__ void sincos_2pi(float u, float &c, float &s)
{
    float r = u * 8.0f;
    int j = (int)r;
    float f = r - (float)j;
    float g = (j & 1) ? (1.0f - f) : f;
    float z = g * g;
    float ps = 0x1.507834p-22f;
    ps = __builtin_fmaf(ps, z, -0x1.32d2ccp-15f);
    ps = __builtin_fmaf(ps, z, 0x1.466bc6p-9f);
    ps = __builtin_fmaf(ps, z, -0x1.4abbcep-4f);
    ps = __builtin_fmaf(ps, z, 0x1.921fb6p-1f);
    float sg = ps * g;
    float pc = -0x1.a6d1f2p-26f;
    pc = __builtin_fmaf(pc, z, 0x1.e1f506p-19f);
    pc = __builtin_fmaf(pc, z, -0x1.55d3c8p-12f);
    pc = __builtin_fmaf(pc, z, 0x1.03c1f0p-6f);
    pc = __builtin_fmaf(pc, z, -0x1.3bd3ccp-2f);
    pc = __builtin_fmaf(pc, z, 1.0f);
    float cg = pc;
    // octants 1,2,5,6 swap the roles of sin and cos
    bool swap = ((j + 1) & 2) != 0;
    float cc = swap ? sg : cg;
    float ss = swap ? cg : sg;
    // cos is negative in octants 2..5, sin in octants 4..7
    bool negc = ((j + 2) & 4) != 0;
    bool negs = (j & 4) != 0;
    c = negc ? -cc : cc;
    s = negs ? -ss : ss;
}

// ---- natural log, x > 0 finite -------------------------------------------------------------
__device__ __forceinline__ float det_logf(float x)
{
    uint32_t u = __float_as_uint(x);
    int e = 0;
    if (u < 0x00800000u) {
        u = __float_as_uint(x * 8388608.0f);
        e = -23;
    }
    e += (int)((u >> 23) & 0xff) - 127;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    if (m > 0x1.6a09e6p+0f) {
        m = m * 0.5f;
        e += 1;
    }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = 0x1.c71c72p-3f;
    p = __builtin_fmaf(p, z, 0x1.24924ap-2f);
    p = __builtin_fmaf(p, z, 0x1.99999ap-2f);
    p = __builtin_fmaf(p, z, 0x1.555556p-1f);
    p = p * z;
    float r = __builtin_fmaf(s, p, s + s);
    return __builtin_fmaf((float)e, 0x1.62e430p-1f, r);
}

// ---- deterministic replacements for the libm calls of the guided path -----------------------
// expf, cosf/sinf of an angle, and the fp64 cos/acos/log of the Best-Fisher rejection sampler
// (reference util/vonmises.h:95-118): fixed sequences of IEEE operations and explicit fma, so the
// device rounds exactly like the CPU restatement and a walk cannot diverge on a last-ulp
// difference between two math libraries.  A few ulp in fp32, < 1e-15
// relative in fp64.
__device__ __forceinline__ float det_two_pow(int k)   // 2^k, k in [-126, 127]
{
    return __uint_as_float((uint32_t)(k + 127) << 23);
}

// e^x: k = floor(x log2(e) + 1/2), r = x - k ln2 in two pieces, degree-7 Taylor in r, 2^k in
// two factors so that subnormal results round once
__device__ __forceinline__ float det_expf(float x)
{
    if (x != x) return x;
    if (x > 88.8f) return __builtin_inff();
    if (x < -104.0f) return 0.0f;
    const float k = floorf(__builtin_fmaf(x, 1.44269504088896341f, 0.5f));
    float r = __builtin_fmaf(k, -0.693359375f, x);
    r = __builtin_fmaf(k, 2.12194440e-4f, r);
    float p = 1.0f / 5040.0f;
    p = __builtin_fmaf(p, r, 1.0f / 720.0f);
    p = __builtin_fmaf(p, r, 1.0f / 120.0f);
    p = __builtin_fmaf(p, r, 1.0f / 24.0f);
    p = __builtin_fmaf(p, r, 1.0f / 6.0f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const int ki = (int)k;
    const int k1 = ki / 2, k2 = ki - k1;
    return p * det_two_pow(k1) * det_two_pow(k2);
}

// cos/sin of an angle in [-pi, pi]: turn fraction, then the octant kernels of the uniform path
__device__ __forceinline__ void det_sincosf(float theta, float *c, float *s)
{
    float u = theta * 0.15915494309189535f;
    if (u < 0.0f) u += 1.0f;
    if (!(u < 1.0f)) u = 0.0f;
    sincos_2pi(u, *c, *s);
}

// ---- fp64 -------------------------------------------------------------------------------
__device__ __forceinline__ double det_cos_taylor_d(double x)   // |x| <= pi/4
{
    const double z = x * x;
    double p = (1.0 / 2432902008176640000.0);
    p = __builtin_fma(p, z, -(1.0 / 6402373705728000.0));
    p = __builtin_fma(p, z, (1.0 / 20922789888000.0));
    p = __builtin_fma(p, z, -(1.0 / 87178291200.0));
    p = __builtin_fma(p, z, (1.0 / 479001600.0));
    p = __builtin_fma(p, z, -(1.0 / 3628800.0));
    p = __builtin_fma(p, z, (1.0 / 40320.0));
    p = __builtin_fma(p, z, -(1.0 / 720.0));
    p = __builtin_fma(p, z, (1.0 / 24.0));
    p = __builtin_fma(p, z, -(1.0 / 2.0));
    p = __builtin_fma(p, z, (1.0));
    return p;
}

__device__ __forceinline__ double det_sin_taylor_d(double x)   // |x| <= pi/4
{
    const double z = x * x;
    double p = (1.0 / 51090942171709440000.0);
    p = __builtin_fma(p, z, -(1.0 / 121645100408832000.0));
    p = __builtin_fma(p, z, (1.0 / 355687428096000.0));
    p = __builtin_fma(p, z, -(1.0 / 1307674368000.0));
    p = __builtin_fma(p, z, (1.0 / 6227020800.0));
    p = __builtin_fma(p, z, -(1.0 / 39916800.0));
    p = __builtin_fma(p, z, (1.0 / 362880.0));
    p = __builtin_fma(p, z, -(1.0 / 5040.0));
    p = __builtin_fma(p, z, (1.0 / 120.0));
    p = __builtin_fma(p, z, -(1.0 / 6.0));
    p = __builtin_fma(p, z, (1.0));
    return x * p;
}

// cos(pi u), u in [0, 1)
__device__ __forceinline__ double det_cospi_d(double u)
{
    const double PI = 3.14159265358979323846;
    int neg = 0;
    if (u > 0.5) { u = 1.0 - u; neg = 1; }
    const double r = (u <= 0.25) ? det_cos_taylor_d(PI * u) : det_sin_taylor_d(PI * (0.5 - u));
    return neg ? -r : r;
}

// asin(x), 0 <= x <= 0.5: x (1 + z r1 (1 + z r2 (...))), r_n = (2n-1)^2 / (2n (2n+1))
__device__ __forceinline__ double det_asin_core_d(double x)
{
    constexpr double R[26] = { (1.0 * 1.0) / (2.0 * 3.0), (3.0 * 3.0) / (4.0 * 5.0), (5.0 * 5.0) / (6.0 * 7.0), (7.0 * 7.0) / (8.0 * 9.0), (9.0 * 9.0) / (10.0 * 11.0), (11.0 * 11.0) / (12.0 * 13.0), (13.0 * 13.0) / (14.0 * 15.0), (15.0 * 15.0) / (16.0 * 17.0), (17.0 * 17.0) / (18.0 * 19.0), (19.0 * 19.0) / (20.0 * 21.0), (21.0 * 21.0) / (22.0 * 23.0), (23.0 * 23.0) / (24.0 * 25.0), (25.0 * 25.0) / (26.0 * 27.0), (27.0 * 27.0) / (28.0 * 29.0), (29.0 * 29.0) / (30.0 * 31.0), (31.0 * 31.0) / (32.0 * 33.0), (33.0 * 33.0) / (34.0 * 35.0), (35.0 * 35.0) / (36.0 * 37.0), (37.0 * 37.0) / (38.0 * 39.0), (39.0 * 39.0) / (40.0 * 41.0), (41.0 * 41.0) / (42.0 * 43.0), (43.0 * 43.0) / (44.0 * 45.0), (45.0 * 45.0) / (46.0 * 47.0), (47.0 * 47.0) / (48.0 * 49.0), (49.0 * 49.0) / (50.0 * 51.0), (51.0 * 51.0) / (52.0 * 53.0) };
    const double z = x * x;
    double t = 1.0;
    for (int n = 26 - 1; n >= 0; --n) t = __builtin_fma(z * R[n], t, 1.0);
    return x * t;
}

// acos(x), -1 <= x <= 1 (NaN outside)
__device__ __forceinline__ double det_acos_d(double x)
{
    const double PI = 3.14159265358979323846, HALF_PI = 1.57079632679489661923;
    if (!(x >= -1.0 && x <= 1.0)) return __builtin_nan("");
    if (x > 0.5) return 2.0 * det_asin_core_d(sqrt((1.0 - x) * 0.5));
    if (x < -0.5) return PI - 2.0 * det_asin_core_d(sqrt((1.0 + x) * 0.5));
    return HALF_PI - (x < 0.0 ? -det_asin_core_d(-x) : det_asin_core_d(x));
}

// ln(x): x = m 2^e, m in (sqrt(1/2), sqrt(2)], s = (m-1)/(m+1), ln m = 2 s sum z^k/(2k+1)
__device__ __forceinline__ double det_log_d(double x)
{
    if (x != x || x < 0.0) return __builtin_nan("");
    if (x == 0.0) return -__builtin_inff();
    if (x > 1.7976931348623157e308) return x;
    int e = 0;
    uint64_t u = (uint64_t)__double_as_longlong(x);
    if ((u >> 52) == 0) {                       // subnormal
        x *= 18014398509481984.0;               // 2^54
        e = -54;
        u = (uint64_t)__double_as_longlong(x);
    }
    e += (int)((u >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((long long)(u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = (1.0 / 21.0);
    p = __builtin_fma(p, z, (1.0 / 19.0));
    p = __builtin_fma(p, z, (1.0 / 17.0));
    p = __builtin_fma(p, z, (1.0 / 15.0));
    p = __builtin_fma(p, z, (1.0 / 13.0));
    p = __builtin_fma(p, z, (1.0 / 11.0));
    p = __builtin_fma(p, z, (1.0 / 9.0));
    p = __builtin_fma(p, z, (1.0 / 7.0));
    p = __builtin_fma(p, z, (1.0 / 5.0));
    p = __builtin_fma(p, z, (1.0 / 3.0));
    p = __builtin_fma(p, z, (1.0));
    return __builtin_fma((double)e, 0.6931471805599453, 2.0 * s * p);
}

}  // namespace wost
