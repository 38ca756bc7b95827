// op_rate.hip -- issue cost of the vector instructions the LBVH traversal is made of (gfx950), measured
// with inline assembly on independent registers, 8 waves per SIMD.  Cycles of a SIMD per wave-instruction.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/op_rate.hip -o tools/micro/op_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
#define OP2(name, asmtext)                                                                                  \
    __global__ __launch_bounds__(256) void k_##name(float *out, int iters, float seed)                      \
    {                                                                                                       \
        float a = seed + threadIdx.x, b = a * 0.5f + 1.0f, c = a * 0.25f + 2.0f, d = a + 3.0f;               \
        float r0 = a, r1 = b, r2 = c, r3 = d;                                                               \
        for (int i = 0; i < iters; ++i) {                                                                   \
            REP8(asm volatile(asmtext "\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(d));) \
        }                                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3;                                            \
    }

// four independent instructions per asm statement: r_k = op(r_k, src_k)
OP2(fma, "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %5, %6\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %7, %4")
OP2(add_f32, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7")
OP2(max_f32, "v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %5\n v_max_f32 %2, %2, %6\n v_max_f32 %3, %3, %7")
OP2(min_f32, "v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %5\n v_min_f32 %2, %2, %6\n v_min_f32 %3, %3, %7")
OP2(med3_f32, "v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %5, %6\n v_med3_f32 %2, %2, %6, %7\n v_med3_f32 %3, %3, %7, %4")
OP2(min3_f32, "v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %5, %6\n v_min3_f32 %2, %2, %6, %7\n v_min3_f32 %3, %3, %7, %4")
OP2(min_u32, "v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %5\n v_min_u32 %2, %2, %6\n v_min_u32 %3, %3, %7")
OP2(max_u32, "v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7")
OP2(min_i32, "v_min_i32 %0, %0, %4\n v_min_i32 %1, %1, %5\n v_min_i32 %2, %2, %6\n v_min_i32 %3, %3, %7")
OP2(add_u32, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7")
OP2(and_b32, "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %6\n v_and_b32 %3, %3, %7")
OP2(and_or, "v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %5, %6\n v_and_or_b32 %2, %2, %6, %7\n v_and_or_b32 %3, %3, %7, %4")
OP2(lshl_add, "v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %5\n v_lshl_add_u32 %2, %2, 2, %6\n v_lshl_add_u32 %3, %3, 2, %7")
OP2(mul_u24, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %5\n v_mul_u32_u24 %2, %2, %6\n v_mul_u32_u24 %3, %3, %7")
OP2(mul_lo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %5\n v_mul_lo_u32 %2, %2, %6\n v_mul_lo_u32 %3, %3, %7")
OP2(cndmask, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %7, vcc")
OP2(cmp_cnd, "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_f32 vcc, %2, %6\n v_cndmask_b32 %3, %3, %7, vcc")
OP2(cmp_sgpr, "v_cmp_lt_f32 s[20:21], %0, %4\n v_cmp_lt_f32 s[22:23], %1, %5\n v_cmp_lt_f32 s[24:25], %2, %6\n v_cmp_lt_f32 s[26:27], %3, %7")
OP2(sqrt, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3")
OP2(rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3")
OP2(mov, "v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7")
OP2(readlane, "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3")
OP2(salu, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_and_b64 s[22:23], s[22:23], exec\n s_or_b64 s[24:25], s[24:25], exec")

template <class K>
static void run(const char *name, K kernel, float *out)
{
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)iters * 32;
    printf("%-12s %.3f ms  %.2f cycles of a SIMD per wave-instruction (8 waves/SIMD, 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (wave_instr * 8));
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
#define R(n) run(#n, k_##n, out);
    R(fma) R(add_f32) R(max_f32) R(min_f32) R(med3_f32) R(min3_f32) R(min_u32) R(max_u32) R(min_i32) R(add_u32) R(and_b32) R(and_or)
    R(lshl_add) R(mul_u24) R(mul_lo) R(cndmask) R(cmp_cnd) R(cmp_sgpr) R(sqrt) R(rcp) R(mov) R(readlane) R(salu)
    return 0;
}
