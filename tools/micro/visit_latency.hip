// visit_latency.hip -- what ONE dependent tree visit costs a lone walker on gfx950: a chain of visits, each a
// 96-byte node record (six 16-byte loads) whose contents decide the next record, plus N_VALU dependent vector
// instructions, from a table that is L1-, L2- or MALL/HBM-resident; with and without touching the four records
// the next visit can go to one visit ahead (they are contiguous: 384 bytes, four 128-byte lines).
// One wave per CU, `lanes` active lanes.  Reports ns per visit.
// Build: hipcc --offload-arch=gfx950 -O3 visit_latency.hip -o visit_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int N_VALU, bool PREFETCH>
__global__ __launch_bounds__(64) void chain_kernel(const float4 *nodes, uint32_t n_nodes, int visits, int lanes, float *out)
{
    if ((int)threadIdx.x >= lanes) return;
        uint32_t r = (blockIdx.x * 64u + threadIdx.x) * 2654435761u % n_nodes;
    float acc = 0.0f;
    uint32_t pf0 = 0, pf1 = 0, pf2 = 0, pf3 = 0;
    for (int i = 0; i < visits; ++i) {
        const float4 *nd = nodes + 6u * r;
        const float4 a = nd[0], b = nd[1], c = nd[2], d = nd[3], e = nd[4], f = nd[5];
        // the four records this visit can go to: contiguous, at a pseudo-random place (no short cycles, no locality)
        const uint32_t child0 = ((r * 2654435761u + (uint32_t)i * 40503u) >> 7) % (n_nodes / 4u) * 4u;
        if (PREFETCH) {
            asm volatile("" ::"v"(pf0), "v"(pf1), "v"(pf2), "v"(pf3));      // last visit's touches have landed by now
            const char *cp = reinterpret_cast<const char *>(nodes + 6u * child0);
            pf0 = *reinterpret_cast<const uint32_t *>(cp);
            pf1 = *reinterpret_cast<const uint32_t *>(cp + 128);
            pf2 = *reinterpret_cast<const uint32_t *>(cp + 256);
            pf3 = *reinterpret_cast<const uint32_t *>(cp + 380);
        }
        float v = a.x + b.y + c.z + d.w + e.x + f.y;
#pragma unroll
        for (int k = 0; k < N_VALU; ++k) v = __builtin_fmaf(v, 1.0000001f, 0.25f);
        acc += v;
        // which child: decided by the data
        const uint32_t pick = (__float_as_uint(v) >> 3) & 3u;
        r = child0 + pick;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc + (float)(pf0 ^ pf1 ^ pf2 ^ pf3);
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
    const int visits = 20000;
    float *out;
    (void)hipMalloc((void **)&out, 256 * 64 * sizeof(float));
    for (uint32_t n_nodes : {256u, 20000u, 4000000u}) {     // 24 KB (L1), 1.9 MB (L2), 384 MB (HBM)
        std::vector<float4> h((size_t)n_nodes * 6);
        for (size_t i = 0; i < h.size(); ++i) h[i] = float4{(float)(rand() % 1000) * 1e-3f, (float)(rand() % 1000) * 1e-3f, 0.5f, 0.25f};
        float4 *d;
        (void)hipMalloc((void **)&d, h.size() * sizeof(float4));
        (void)hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
        for (int lanes : {1, 4, 64}) {
            const float t0 = time_ms([&] { hipLaunchKernelGGL((chain_kernel<0, false>), dim3(256), dim3(64), 0, 0, d, n_nodes, visits, lanes, out); });
            const float t1 = time_ms([&] { hipLaunchKernelGGL((chain_kernel<150, false>), dim3(256), dim3(64), 0, 0, d, n_nodes, visits, lanes, out); });
            const float t2 = time_ms([&] { hipLaunchKernelGGL((chain_kernel<150, true>), dim3(256), dim3(64), 0, 0, d, n_nodes, visits, lanes, out); });
            const float t3 = time_ms([&] { hipLaunchKernelGGL((chain_kernel<0, true>), dim3(256), dim3(64), 0, 0, d, n_nodes, visits, lanes, out); });
            printf("table %7.1f KB, %2d lanes: load only %6.0f ns/visit, load + 150 VALU %6.0f, same with the children touched a visit ahead %6.0f (load only: %6.0f)\n",
                   n_nodes * 96.0 / 1024.0, lanes, t0 * 1e6 / visits, t1 * 1e6 / visits, t2 * 1e6 / visits, t3 * 1e6 / visits);
        }
        (void)hipFree(d);
    }
    return 0;
}
