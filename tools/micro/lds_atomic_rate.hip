// lds_atomic_rate.hip -- what an LDS atomic costs on gfx950: 64 lanes adding to random entries of an LDS table
// (no two lanes of a wave on the same entry unless SHARE > 1), 32- and 64-bit integers, floats, with and
// without the returned value.  Reports wave-instructions per microsecond and CU and cycles per wave-instruction.
// Build: hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip -o lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }

// KIND 0: u32, 1: u64, 2: f32, 3: u32 returning, 4: u64 returning
template <int KIND, int SHARE>
__global__ __launch_bounds__(1024) void atomic_kernel(int n_entries, int iters, unsigned long long *out)
{
    extern __shared__ unsigned long long tab[];
    for (int i = threadIdx.x; i < n_entries; i += blockDim.x) tab[i] = 0;
    __syncthreads();
    uint32_t s = (blockIdx.x * 1024u + threadIdx.x / SHARE) * 2654435761u + 12345u;
    unsigned long long acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t e = (lcg(s) >> 8) % (uint32_t)n_entries;
            if (KIND == 0) atomicAdd(reinterpret_cast<unsigned int *>(tab) + 2 * e, 3u);
            else if (KIND == 1) atomicAdd(tab + e, 3ull);
            else if (KIND == 2) atomicAdd(reinterpret_cast<float *>(tab) + 2 * e, 1.5f);
            else if (KIND == 3) acc += atomicAdd(reinterpret_cast<unsigned int *>(tab) + 2 * e, 3u);
            else acc += atomicAdd(tab + e, 3ull);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tab[0] + acc;
    else if (acc == 0x123456789ull) out[blockIdx.x] = acc;
}

template <int KIND, int SHARE>
static void run(const char *name, int n_entries, unsigned long long *out)
{
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t lds = (size_t)n_entries * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(atomic_kernel<KIND, SHARE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((atomic_kernel<KIND, SHARE>), dim3(blocks), dim3(1024), lds, 0, n_entries, 10, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((atomic_kernel<KIND, SHARE>), dim3(blocks), dim3(1024), lds, 0, n_entries, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_cu = 16.0 * iters * 8;      // 16 waves per block, one block per CU
    printf("%-28s %5d entries, %2d lanes per entry: %7.1f wave-instr/us/CU, %6.1f cycles per wave-instruction (at 2.4 GHz)\n", name, n_entries, SHARE,
           wave_instr_per_cu / (ms * 1e3), ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
}

int main()
{
    unsigned long long *out;
    (void)hipMalloc((void **)&out, 256 * 8);
    for (int n : {256, 2048, 15000}) {
        run<0, 1>("ds_add_u32", n, out);
        run<1, 1>("ds_add_u64", n, out);
        run<2, 1>("ds_add_f32", n, out);
        run<3, 1>("ds_add_rtn_u32", n, out);
        run<4, 1>("ds_add_rtn_u64", n, out);
    }
    run<1, 4>("ds_add_u64", 2048, out);
    run<1, 16>("ds_add_u64", 2048, out);
    run<0, 16>("ds_add_u32", 2048, out);
    return 0;
}
