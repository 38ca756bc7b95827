// half_wave.hip -- does a wave64 vector instruction cost less when one 32-lane half of EXEC is empty?
// gfx950 issues a wave64 VALU instruction as two passes of 32 lanes (SIMD-32).  Three masks with the same number of
// active lanes (or all): all 64, lanes 0..31 only, even lanes only.  Build: hipcc --offload-arch=gfx950 -O3 half_wave.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    const int lane = threadIdx.x & 63;
    const bool on = MODE == 0 ? true : MODE == 1 ? lane < 32 : MODE == 2 ? (lane & 1) == 0 : MODE == 3 ? lane < 16 : lane == 0;
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    if (on) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f); a2 = __builtin_fmaf(a2, 1.0001f, 0.5f); a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
                a4 = __builtin_fmaf(a4, 1.0001f, 0.5f); a5 = __builtin_fmaf(a5, 1.0001f, 0.5f); a6 = __builtin_fmaf(a6, 1.0001f, 0.5f); a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
static void run(const char *name, float *out)
{
    const int iters = 2000, blocks = 256 * 8;       // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * 8;      // wave-instructions per SIMD (8 waves x 64 fma per trip)
    printf("%-28s %8.3f ms  %5.2f cycles of a SIMD per wave-instruction (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("all 64 lanes", out);
    run<1>("lanes 0..31", out);
    run<2>("even lanes", out);
    run<3>("lanes 0..15", out);
    run<4>("lane 0", out);
    return 0;
}
