// wost_cells.h -- closest-point query through per-cell candidate lists (device side of cell_grid.h).
//
// lbvh `nearest` (reference call site integrator/uniform/integrator.cu:138) answered without a tree
// descent: the cell of the query point names the chunks (16 segments + oriented box each) that can
// hold the closest segment of ANY point of the cell.  The query measures their boxes (independent
// gathers, four per step, no dependent chain), scans the nearest chunk exactly, and then scans the
// few other chunks whose box is not farther than the best distance found.  Exact minimum over a
// superset of the relevant segments, ties to the lowest ORIGINAL index: bit-identical to the tree
// traversal and to the oracle (DESIGN.md "segment distance").
//
// gfx950 notes: packed fp32 (v_pk_fma_f32) issues at half the rate of v_fma_f32 on this chip
// (tools/micro/valu_rate.hip), so everything here is plain scalar fp32; a wave's lanes run the same
// straight-line body per group of four candidates, and the divergent part (which chunks to scan) is
// turned into a per-lane bit mask that is drained one chunk per trip by all lanes together.
#pragma once

#include "wost_device.h"

#ifndef WOST_CELLS_SCAN_UNROLL
#define WOST_CELLS_SCAN_UNROLL 1
#endif
#ifndef WOST_CELLS_WAVES
#define WOST_CELLS_WAVES 5      // resident waves per SIMD the cell-list kernels are compiled for (96 VGPRs)
#endif

namespace wost {

// Slow, exact pass over the 16 entries of a chunk, taken only after the fast pass below has met an
// exact tie (two segments at the same squared distance: ties go to the lowest ORIGINAL index).  The
// result is the lexicographic minimum of (d2, original index) over everything seen so far, so running
// it over entries the fast pass has already accepted changes nothing.
__device__ __forceinline__ void cells_scan_ties(const DevMesh &m, uint32_t id, float qx, float qy, Closest &best, int32_t &best_orig)
{
    const DevCells &g = m.cells;
    const float *r = reinterpret_cast<const float *>(g.cseg + 20u * id);
#pragma unroll 1
    for (int j = 0; j < 16; ++j) {
        const float dj = obb_d2(r[j], r[16 + j], r[32 + j], r[48 + j], r[64 + j], 0.0f, qx, qy);
        const int e = (int)(16u * id) + j;
        if (dj < best.d2) {
            best.d2 = dj;
            best.slot = e;
            best_orig = -1;
        } else if (dj == best.d2 && e != best.slot) {
            if (best_orig < 0) best_orig = (best.slot >= 0) ? m.segOrig[g.cslot[best.slot]] : WOST_FAR_INDEX;
            const int sl = g.cslot[e];
            const int o = sl >= 0 ? m.segOrig[sl] : WOST_FAR_INDEX;
            if (o < best_orig) {
                best.slot = e;
                best_orig = o;
            }
        }
    }
}

// exact distances to the 16 segments of chunk `id`; best.slot is a chunk ENTRY index (16 id + j)
__device__ __forceinline__ void cells_scan(const DevMesh &m, uint32_t id, float qx, float qy, Closest &best, int32_t &best_orig)
{
    const float4 *r = m.cells.cseg + 20u * id;
    bool tie = false;
    // one group of four segments per trip: unrolled, the 20 gathers of a chunk would all be hoisted
    // and cost 80 registers (a resident wave per SIMD) for latency the other waves already hide
#pragma unroll WOST_CELLS_SCAN_UNROLL
    for (int q = 0; q < 4; ++q) {
        const float4 CX = r[q], CY = r[4 + q], UX = r[8 + q], UY = r[12 + q], HL = r[16 + q];
        const float d0 = obb_d2(CX.x, CY.x, UX.x, UY.x, HL.x, 0.0f, qx, qy);
        const float d1 = obb_d2(CX.y, CY.y, UX.y, UY.y, HL.y, 0.0f, qx, qy);
        const float d2 = obb_d2(CX.z, CY.z, UX.z, UY.z, HL.z, 0.0f, qx, qy);
        const float d3 = obb_d2(CX.w, CY.w, UX.w, UY.w, HL.w, 0.0f, qx, qy);
        const float mn = fminf(fminf(d0, d1), fminf(d2, d3));
        const int n_eq = (d0 == mn) + (d1 == mn) + (d2 == mn) + (d3 == mn);
        const int e = (int)(16u * id) + 4 * q + ((d0 == mn) ? 0 : (d1 == mn) ? 1 : (d2 == mn) ? 2 : 3);
        const bool win = (n_eq == 1) && mn < best.d2;
        tie = tie || (mn <= best.d2 && !win);
        best.d2 = win ? mn : best.d2;
        best.slot = win ? e : best.slot;
        best_orig = win ? -1 : best_orig;
    }
    if (tie) cells_scan_ties(m, id, qx, qy, best, best_orig);
}

__device__ __forceinline__ float cells_box_d2(const DevCells &g, uint32_t id, float qx, float qy)
{
    const float4 a = g.cbox[2u * id], b = g.cbox[2u * id + 1u];
    return obb_d2(a.x, a.y, a.z, a.w, b.x, b.y, qx, qy);
}

// Closest point of (qx, qy) on the mesh; returns the squared distance and the SLOT of the segment
// in the tree's leaf order (the index of segA / segInv / segCol), like closest_point().
__device__ __forceinline__ Closest closest_point_cells(const DevMesh &m, float qx, float qy)
{
    const DevCells &g = m.cells;
    const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h;
    const int ix = (int)floorf(fx), iy = (int)floorf(fy);
    const bool inside = fx >= 0.0f && fy >= 0.0f && ix < g.nx && iy < g.ny;      // false for NaN
    const uint32_t cell = inside ? (uint32_t)iy * (uint32_t)g.nx + (uint32_t)ix : (uint32_t)g.nx * (uint32_t)g.ny;
    const uint32_t beg = g.cell_off[cell], end = g.cell_off[cell + 1];
    // ---- pass 1: the nearest box ----------------------------------------------------------
    float m1 = WOST_INF;
    uint32_t arg = (uint32_t)g.n_chunks;
    for (uint32_t i = beg; i < end; ++i) {
        const uint2 pk = g.ids4[i];
        const uint32_t i0 = pk.x & 0xffffu, i1 = pk.x >> 16, i2 = pk.y & 0xffffu, i3 = pk.y >> 16;
        const float d0 = cells_box_d2(g, i0, qx, qy), d1 = cells_box_d2(g, i1, qx, qy);
        const float d2 = cells_box_d2(g, i2, qx, qy), d3 = cells_box_d2(g, i3, qx, qy);
        arg = d0 < m1 ? i0 : arg; m1 = fminf(d0, m1);
        arg = d1 < m1 ? i1 : arg; m1 = fminf(d1, m1);
        arg = d2 < m1 ? i2 : arg; m1 = fminf(d2, m1);
        arg = d3 < m1 ? i3 : arg; m1 = fminf(d3, m1);
    }
    Closest best{WOST_INF, -1};
    int32_t best_orig = -1;
    // ---- the nearest chunk, then pass 2: every other chunk whose box can still hold a closer or
    // tied segment.  Positions 0..63 of the list go into a per-lane bit mask that is drained one
    // chunk per trip by all lanes together (ONE inlined scan serves the nearest chunk and the mask);
    // longer lists (the all-chunks list of a query outside the grid) scan their tail on the spot.
    const uint16_t *ids = reinterpret_cast<const uint16_t *>(g.ids4 + beg);
    unsigned long long pending = 0ull;
    uint32_t cur = arg;
    bool take = true, first = true;
    for (;;) {
        if (take) cells_scan(m, cur, qx, qy, best, best_orig);      // the sentinel chunk scans as "nothing found"
        if (first) {
            first = false;
            for (uint32_t i = beg; i < end; ++i) {
                const uint2 pk = g.ids4[i];
                const uint32_t i0 = pk.x & 0xffffu, i1 = pk.x >> 16, i2 = pk.y & 0xffffu, i3 = pk.y >> 16;
                const float d0 = cells_box_d2(g, i0, qx, qy), d1 = cells_box_d2(g, i1, qx, qy);
                const float d2 = cells_box_d2(g, i2, qx, qy), d3 = cells_box_d2(g, i3, qx, qy);
                const uint32_t bits = ((d0 <= best.d2 && i0 != arg) ? 1u : 0u) | ((d1 <= best.d2 && i1 != arg) ? 2u : 0u) |
                                      ((d2 <= best.d2 && i2 != arg) ? 4u : 0u) | ((d3 <= best.d2 && i3 != arg) ? 8u : 0u);
                const uint32_t pos = (i - beg) * 4u;
                if (pos < 64u) pending |= (unsigned long long)bits << pos;
                uint32_t rest = pos < 64u ? 0u : bits;
                while (rest) {
                    const int k = __ffs((int)rest) - 1;
                    rest &= rest - 1u;
                    cells_scan(m, k == 0 ? i0 : k == 1 ? i1 : k == 2 ? i2 : i3, qx, qy, best, best_orig);
                }
            }
        }
        if (!pending) break;
        const int j = __ffsll((long long)pending) - 1;
        pending &= pending - 1ull;
        cur = ids[j];
        // the best distance may have shrunk since the bit was set: measure the box again
        take = cells_box_d2(g, cur, qx, qy) <= best.d2;
    }
    best.slot = best.slot >= 0 ? g.cslot[best.slot] : -1;
    return best;
}

}  // namespace wost
