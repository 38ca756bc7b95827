// gather_rate.hip -- what a per-lane gather costs on gfx950: 64 lanes reading 4 / 8 / 16 bytes each from
// random 64-byte-aligned records of a table (L1-, L2- or MALL-resident), from global memory and from
// LDS, against lanes that share records.  Reports wave-instructions per microsecond per CU and the
// cycles one CU spends per wave-instruction.  Build: hipcc --offload-arch=gfx950 -O3 gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }

// WIDTH: bytes per lane (4, 8, 16); SHARE: lanes per distinct record (1 = fully divergent)
template <int WIDTH, int SHARE>
__global__ __launch_bounds__(256) void gather_kernel(const float4 *table, uint32_t n_records, int iters, float *out)
{
    uint32_t s = (blockIdx.x * 256u + threadIdx.x / SHARE) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        // 8 independent gathers per trip, addresses do not depend on loaded data
        uint32_t idx[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) idx[k] = (lcg(s) >> 8) % n_records;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 *p = table + 4u * idx[k];      // 64-byte records
            if (WIDTH == 16) { const float4 v = *p; acc += v.x + v.w; }
            else if (WIDTH == 8) { const float2 v = *reinterpret_cast<const float2 *>(p); acc += v.x + v.y; }
            else { acc += *reinterpret_cast<const float *>(p); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int SHARE>
__global__ __launch_bounds__(1024) void lds_gather_kernel(const float4 *table, uint32_t n_records, int iters, float *out)
{
    extern __shared__ float4 lds[];
    for (uint32_t i = threadIdx.x; i < n_records; i += blockDim.x) lds[i] = table[i];
    __syncthreads();
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x / SHARE) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 v = lds[(lcg(s) >> 8) % n_records];
            acc += v.x + v.w;
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <class F>
static float time_ms(F f)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const int iters = 400;
    float *out;
    (void)hipMalloc(&out, 256 * 32 * 1024 * sizeof(float));
    for (size_t table_kb : {16, 1024, 65536, 1048576}) {
        const uint32_t n_records = (uint32_t)(table_kb * 1024 / 64);
        float4 *table;
        (void)hipMalloc(&table, table_kb * 1024);
        (void)hipMemset(table, 0, table_kb * 1024);
        const int blocks = 256 * 6;      // 6 blocks of 4 waves per CU
        const double wave_instr_per_cu = (double)iters * 8 * 6 * 4;
#define RUN(W, S)                                                                                                          \
    {                                                                                                                      \
        const float ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<W, S>), dim3(blocks), dim3(256), 0, 0, table, n_records, iters, out); }); \
        printf("table %7zu KB  %2d B/lane, %2d lanes/record: %8.3f ms  %7.1f cycles of a CU per wave-instruction (2.4 GHz)\n", table_kb, W, S, ms, \
               ms * 1e-3 * 2.4e9 / wave_instr_per_cu);                                                                    \
    }
        RUN(16, 1) RUN(8, 1) RUN(4, 1) RUN(16, 4) RUN(16, 16) RUN(16, 64)
#undef RUN
        (void)hipFree(table);
    }
    {
        const uint32_t n_records = 96 * 1024 / 16;      // a 96 KB table of 16-byte records in LDS
        float4 *table;
        (void)hipMalloc(&table, 96 * 1024);
        (void)hipMemset(table, 0, 96 * 1024);
        const double wave_instr_per_cu = (double)iters * 8 * 16;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lds_gather_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lds_gather_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        float ms = time_ms([&] { hipLaunchKernelGGL((lds_gather_kernel<1>), dim3(256), dim3(1024), 96 * 1024, 0, table, n_records, iters, out); });
        printf("LDS 96 KB, 16 B/lane, divergent:      %8.3f ms  %7.1f cycles of a CU per wave-instruction\n", ms, ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
        ms = time_ms([&] { hipLaunchKernelGGL((lds_gather_kernel<16>), dim3(256), dim3(1024), 96 * 1024, 0, table, n_records, iters, out); });
        printf("LDS 96 KB, 16 B/lane, 16 lanes/record: %8.3f ms  %7.1f cycles of a CU per wave-instruction\n", ms, ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
    }
    return 0;
}
