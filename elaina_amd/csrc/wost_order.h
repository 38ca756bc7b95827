// wost_order.h -- the order in which a persistent walk launch hands pixels to its lanes (wost_order.hip).  Not part of the C-ABI.
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace wost {

// scratch of one handle: two key / value buffers for the sort and rocPRIM's temporary storage, sized for `cap` walkers
struct WalkOrder {
    uint32_t *keys[2] = {nullptr, nullptr};
    uint32_t *vals[2] = {nullptr, nullptr};
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    size_t cap = 0;
};

int order_alloc(WalkOrder &o, size_t cap);
void order_free(WalkOrder &o);

// order[k] = queue slot of the k-th walker to start: the pixel with the longest expected chain of walk steps first.
// A pixel's samples run one after the other (one PCG32 stream per pixel, reference integrator/uniform/integrator.cu:71-77), so a
// pixel is one indivisible job of spp walks; its length grows with the distance of the evaluation point from the Dirichlet
// boundary (config 2: 290 steps within one shell width, 2 900 at 100 shell widths; EXPERIMENTS 25), which init_kernel has
// cached per slot (`d0_d2`, the squared distance).  Keys are the exponent and two mantissa bits of that float (steps of 9 % in
// distance), sorted descending and stable: inside a bucket the tile order of the queue survives, and the order is the same in
// every run.  Returns a hipError_t as int; *order_out points into `o`.
int order_by_distance(WalkOrder &o, const float *d0_d2, uint32_t n, hipStream_t stream, const uint32_t **order_out);

// order[k] = queue slot of the walker with the k-th largest `est` (expected walk steps left, written by a persistent launch when
// it hands its pixels over): keys are est / 8 clamped to 12 bits, descending and stable.  An estimate of at least T (a multiple
// of 8) sorts before every estimate below T: the first `count(est >= T)` entries are exactly those walkers.
int order_by_estimate(WalkOrder &o, const float *est, uint32_t n, hipStream_t stream, const uint32_t **order_out);

}  // namespace wost
