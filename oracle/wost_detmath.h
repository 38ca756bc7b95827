/*
 * wost_detmath.h -- deterministic replacements for the libm calls of the guided path
 * (expf, cosf/sinf, and the fp64 cos/acos/log of the Best-Fisher rejection sampler,
 * reference util/vonmises.h:95-118), so that the CPU oracle and the HIP path round
 * identically and a walk cannot diverge on a last-ulp difference between glibc and the device
 * library.  Every function is a fixed sequence of IEEE operations and explicit fma calls (the
 * build uses -ffp-contract=off); DESIGN.md section 2.1 states the algorithms.  Accuracy: a few
 * ulp in fp32, < 1e-15 relative in fp64 -- far inside the fp16 noise of the reference network.
 * TEST INFRASTRUCTURE ONLY (oracle side; the product has its own statement in
 * elaina_amd/csrc/wost_math.h).
 */
#ifndef WOST_DETMATH_H
#define WOST_DETMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float wo_bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t wo_double_to_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double wo_bits_to_double(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
void wo_sincos_2pi(float u, float *c, float *s);

static inline float wo_two_pow(int k)   /* 2^k, k in [-126, 127] */
{
    return wo_bits_to_float((uint32_t)(k + 127) << 23);
}

/* e^x: k = floor(x log2(e) + 1/2), r = x - k ln2 in two pieces, degree-7 Taylor in r, 2^k in
 * two factors so that subnormal results round once */
static inline float wo_expf(float x)
{
    if (x != x) return x;
    if (x > 88.8f) return INFINITY;
    if (x < -104.0f) return 0.0f;
    const float k = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(k, -0.693359375f, x);
    r = fmaf(k, 2.12194440e-4f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    const int ki = (int)k;
    const int k1 = ki / 2, k2 = ki - k1;
    return p * wo_two_pow(k1) * wo_two_pow(k2);
}

/* cos/sin of an angle in [-pi, pi]: turn fraction, then the octant kernels of the uniform path */
static inline void wo_sincosf(float theta, float *c, float *s)
{
    float u = theta * 0.15915494309189535f;
    if (u < 0.0f) u += 1.0f;
    if (!(u < 1.0f)) u = 0.0f;
    wo_sincos_2pi(u, c, s);
}

/* ---- fp64 ------------------------------------------------------------------------------- */
static inline double wo_cos_taylor_d(double x)   /* |x| <= pi/4 */
{
    const double z = x * x;
    double p = (1.0 / 2432902008176640000.0);
    p = fma(p, z, -(1.0 / 6402373705728000.0));
    p = fma(p, z, (1.0 / 20922789888000.0));
    p = fma(p, z, -(1.0 / 87178291200.0));
    p = fma(p, z, (1.0 / 479001600.0));
    p = fma(p, z, -(1.0 / 3628800.0));
    p = fma(p, z, (1.0 / 40320.0));
    p = fma(p, z, -(1.0 / 720.0));
    p = fma(p, z, (1.0 / 24.0));
    p = fma(p, z, -(1.0 / 2.0));
    p = fma(p, z, (1.0));
    return p;
}

static inline double wo_sin_taylor_d(double x)   /* |x| <= pi/4 */
{
    const double z = x * x;
    double p = (1.0 / 51090942171709440000.0);
    p = fma(p, z, -(1.0 / 121645100408832000.0));
    p = fma(p, z, (1.0 / 355687428096000.0));
    p = fma(p, z, -(1.0 / 1307674368000.0));
    p = fma(p, z, (1.0 / 6227020800.0));
    p = fma(p, z, -(1.0 / 39916800.0));
    p = fma(p, z, (1.0 / 362880.0));
    p = fma(p, z, -(1.0 / 5040.0));
    p = fma(p, z, (1.0 / 120.0));
    p = fma(p, z, -(1.0 / 6.0));
    p = fma(p, z, (1.0));
    return x * p;
}

/* cos(pi u), u in [0, 1) */
static inline double wo_cospi_d(double u)
{
    const double PI = 3.14159265358979323846;
    int neg = 0;
    if (u > 0.5) { u = 1.0 - u; neg = 1; }
    const double r = (u <= 0.25) ? wo_cos_taylor_d(PI * u) : wo_sin_taylor_d(PI * (0.5 - u));
    return neg ? -r : r;
}

/* asin(x), 0 <= x <= 0.5: x (1 + z r1 (1 + z r2 (...))), r_n = (2n-1)^2 / (2n (2n+1)) */
static inline double wo_asin_core_d(double x)
{
    static const double R[26] = { (1.0 * 1.0) / (2.0 * 3.0), (3.0 * 3.0) / (4.0 * 5.0), (5.0 * 5.0) / (6.0 * 7.0), (7.0 * 7.0) / (8.0 * 9.0), (9.0 * 9.0) / (10.0 * 11.0), (11.0 * 11.0) / (12.0 * 13.0), (13.0 * 13.0) / (14.0 * 15.0), (15.0 * 15.0) / (16.0 * 17.0), (17.0 * 17.0) / (18.0 * 19.0), (19.0 * 19.0) / (20.0 * 21.0), (21.0 * 21.0) / (22.0 * 23.0), (23.0 * 23.0) / (24.0 * 25.0), (25.0 * 25.0) / (26.0 * 27.0), (27.0 * 27.0) / (28.0 * 29.0), (29.0 * 29.0) / (30.0 * 31.0), (31.0 * 31.0) / (32.0 * 33.0), (33.0 * 33.0) / (34.0 * 35.0), (35.0 * 35.0) / (36.0 * 37.0), (37.0 * 37.0) / (38.0 * 39.0), (39.0 * 39.0) / (40.0 * 41.0), (41.0 * 41.0) / (42.0 * 43.0), (43.0 * 43.0) / (44.0 * 45.0), (45.0 * 45.0) / (46.0 * 47.0), (47.0 * 47.0) / (48.0 * 49.0), (49.0 * 49.0) / (50.0 * 51.0), (51.0 * 51.0) / (52.0 * 53.0) };
    const double z = x * x;
    double t = 1.0;
    for (int n = 26 - 1; n >= 0; --n) t = fma(z * R[n], t, 1.0);
    return x * t;
}

/* acos(x), -1 <= x <= 1 (NaN outside) */
static inline double wo_acos_d(double x)
{
    const double PI = 3.14159265358979323846, HALF_PI = 1.57079632679489661923;
    if (!(x >= -1.0 && x <= 1.0)) return NAN;
    if (x > 0.5) return 2.0 * wo_asin_core_d(sqrt((1.0 - x) * 0.5));
    if (x < -0.5) return PI - 2.0 * wo_asin_core_d(sqrt((1.0 + x) * 0.5));
    return HALF_PI - (x < 0.0 ? -wo_asin_core_d(-x) : wo_asin_core_d(x));
}

/* ln(x): x = m 2^e, m in (sqrt(1/2), sqrt(2)], s = (m-1)/(m+1), ln m = 2 s sum z^k/(2k+1) */
static inline double wo_log_d(double x)
{
    if (x != x || x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x > 1.7976931348623157e308) return x;
    int e = 0;
    uint64_t u = wo_double_to_bits(x);
    if ((u >> 52) == 0) {                       /* subnormal */
        x *= 18014398509481984.0;               /* 2^54 */
        e = -54;
        u = wo_double_to_bits(x);
    }
    e += (int)((u >> 52) & 0x7ff) - 1023;
    double m = wo_bits_to_double((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = (1.0 / 21.0);
    p = fma(p, z, (1.0 / 19.0));
    p = fma(p, z, (1.0 / 17.0));
    p = fma(p, z, (1.0 / 15.0));
    p = fma(p, z, (1.0 / 13.0));
    p = fma(p, z, (1.0 / 11.0));
    p = fma(p, z, (1.0 / 9.0));
    p = fma(p, z, (1.0 / 7.0));
    p = fma(p, z, (1.0 / 5.0));
    p = fma(p, z, (1.0 / 3.0));
    p = fma(p, z, (1.0));
    return fma((double)e, 0.6931471805599453, 2.0 * s * p);
}

#endif
