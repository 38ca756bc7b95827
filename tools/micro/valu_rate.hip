// valu_rate.hip -- issue rate of the vector instructions the walk kernels are made of (gfx950):
// v_fma_f32 against v_pk_fma_f32, v_max_f32, v_min_u32, v_cndmask, ds_read/ds_write, at 1..8 waves
// per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-7f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const f32x2 pm = {m, m}, pc = {c, c};
    uint32_t u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) {        // 8 independent v_fma_f32
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            } else if (KIND == 1) { // 4 independent v_pk_fma_f32 (the same 8 fmas)
                p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
                p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
            } else if (KIND == 2) { // 8 v_max_f32
                a0 = fmaxf(a0, a1 * 0.5f); a1 = fmaxf(a1, c); a2 = fmaxf(a2, a3); a3 = fmaxf(a3, c);
                a4 = fmaxf(a4, a5); a5 = fmaxf(a5, c); a6 = fmaxf(a6, a7); a7 = fmaxf(a7, c);
            } else if (KIND == 3) { // 8 integer min/max (the key sort)
                uint32_t lo = min(u0, u1), hi = max(u0, u1); u0 = lo + 1; u1 = hi;
                lo = min(u2, u3); hi = max(u2, u3); u2 = lo + 3; u3 = hi;
            }
        }
    }
    if (KIND == 1) { a0 = p0.x + p0.y; a1 = p1.x + p1.y; a2 = p2.x + p2.y; a3 = p3.x + p3.y; a4 = a5 = a6 = a7 = 0; }
    if (KIND == 3) { a0 = (float)(u0 + u1 + u2 + u3); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
static void run(const char *name, int ops_per_inner, float *out)
{
    const int iters = 2000;
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        const int blocks = 256 * waves_per_simd;   // 256 CUs, 4 waves per block = one per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double wave_instr = (double)iters * 16 * ops_per_inner;                     // per wave
        const double cyc_per_instr = ms * 1e-3 * 2.4e9 / (wave_instr * waves_per_simd);   // SIMD cycles per wave-instruction
        printf("%-14s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction at 2.4 GHz\n", name, waves_per_simd, ms, cyc_per_instr);
    }
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", 8, out);
    run<1>("v_pk_fma_f32", 4, out);
    run<2>("v_max_f32", 9, out);
    run<3>("v_min/max_u32", 6, out);
    hipFree(out);
    return 0;
}
