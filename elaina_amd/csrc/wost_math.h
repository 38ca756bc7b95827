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

}  // namespace wost
