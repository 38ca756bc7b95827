// wost_order.hip -- longest-chain-first order of a persistent walk launch (see wost_order.h).  gfx950 / MI355X only.
#include "wost_order.h"

#include <cstring>

#include <rocprim/rocprim.hpp>

namespace wost {

namespace {
constexpr unsigned kKeyBits = 10;      // sign (0) + 8 exponent bits + 2 mantissa bits of a non-negative float, inverted

__global__ __launch_bounds__(256) void order_keys_kernel(const float *d0_d2, uint32_t n, uint32_t *keys, uint32_t *vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t bits = __float_as_uint(d0_d2[i]) & 0x7fffffffu;
    keys[i] = 0x3ffu - (bits >> 21);
    vals[i] = i;
}
constexpr unsigned kEstBits = 12;

__global__ __launch_bounds__(256) void estimate_keys_kernel(const float *est, uint32_t n, uint32_t *keys, uint32_t *vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float e = est[i];
    const uint32_t q = e >= 32760.0f ? 0xfffu : (e > 0.0f ? (uint32_t)(e * 0.125f) : 0u);      // (a NaN sorts last)
    keys[i] = 0xfffu - q;
    vals[i] = i;
}
}  // namespace

int order_alloc(WalkOrder &o, size_t cap)
{
    order_free(o);
    if (cap == 0) return (int)hipSuccess;
    size_t bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, cap, 0u, kKeyBits);
    if (e != hipSuccess) return (int)e;
    size_t bytes2 = 0;
    e = rocprim::radix_sort_pairs(nullptr, bytes2, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, cap, 0u, kEstBits);
    if (e != hipSuccess) return (int)e;
    bytes = bytes > bytes2 ? bytes : bytes2;
    o.tmp_bytes = bytes;
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipMalloc((void **)&o.keys[k], cap * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc((void **)&o.vals[k], cap * sizeof(uint32_t));
    }
    if (e == hipSuccess) e = hipMalloc(&o.tmp, bytes > 0 ? bytes : 16);
    if (e != hipSuccess) {
        order_free(o);
        return (int)e;
    }
    o.cap = cap;
    return (int)hipSuccess;
}

void order_free(WalkOrder &o)
{
    for (int k = 0; k < 2; ++k) {
        if (o.keys[k]) (void)hipFree(o.keys[k]);
        if (o.vals[k]) (void)hipFree(o.vals[k]);
        o.keys[k] = o.vals[k] = nullptr;
    }
    if (o.tmp) (void)hipFree(o.tmp);
    o.tmp = nullptr;
    o.tmp_bytes = 0;
    o.cap = 0;
}

int order_by_distance(WalkOrder &o, const float *d0_d2, uint32_t n, hipStream_t stream, const uint32_t **order_out)
{
    if (n > o.cap) return (int)hipErrorInvalidValue;
    *order_out = o.vals[1];
    if (n == 0) return (int)hipSuccess;
    hipLaunchKernelGGL(order_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d0_d2, n, o.keys[0], o.vals[0]);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    size_t bytes = o.tmp_bytes;
    e = rocprim::radix_sort_pairs(o.tmp, bytes, o.keys[0], o.keys[1], o.vals[0], o.vals[1], (size_t)n, 0u, kKeyBits, stream);
    return (int)e;
}

int order_by_estimate(WalkOrder &o, const float *est, uint32_t n, hipStream_t stream, const uint32_t **order_out)
{
    if (n > o.cap) return (int)hipErrorInvalidValue;
    *order_out = o.vals[1];
    if (n == 0) return (int)hipSuccess;
    hipLaunchKernelGGL(estimate_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, est, n, o.keys[0], o.vals[0]);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    size_t bytes = o.tmp_bytes;
    e = rocprim::radix_sort_pairs(o.tmp, bytes, o.keys[0], o.keys[1], o.vals[0], o.vals[1], (size_t)n, 0u, kEstBits, stream);
    return (int)e;
}

}  // namespace wost
