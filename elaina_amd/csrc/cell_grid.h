// cell_grid.h -- candidate lists for the closest-point query, built on the host next to the LBVH.
//
// The shipped diffusion-curve scenes are made of ~60 000 segments of length ~0.1 while the
// epsilon shell is 1 and a walk step is as long as the distance to the boundary: a closest-point
// query answered by descending an 8-level tree spends most of its node visits finding the right
// neighbourhood (one dependent 96-byte gather and ~130 vector instructions per level).  This
// structure answers "which pieces of the boundary can be closest to ANY point of this cell" ahead
// of time:
//   * a CHUNK is 16 consecutive segments in the order of the LBVH's leaves (Morton order refined by
//     the builder's SAH sweep: a short, nearly straight piece of one curve) with the oriented
//     bounding box of the piece; the exact records of its segments are stored together, 320 bytes;
//   * a uniform grid of square cells covers the scene; every cell stores the ids of the chunks
//     that can hold the closest segment (or a segment tied with it) of any point of the cell:
//     all chunks whose box lies within  d(centre) + cell diagonal  of the centre (+ slack for
//     fp32 rounding), where d(centre) is the exact distance of the cell centre to the boundary;
//   * a query outside the grid uses one shared list of all chunks (correct, slow, rare: scenes
//     whose walkers leave the bounding region of the geometry are served by the tree kernels).
// A query then is: measure the boxes of the cell's chunks, scan the segments of the nearest one
// exactly, scan every other chunk whose box is not farther than the best distance so far.  The
// answer is the exact minimum over a superset of the relevant segments with ties broken by the
// lowest ORIGINAL index -- the same definition the tree traversal and the oracle implement, so
// results are bit-identical; only the candidate set is found differently.
#pragma once
#include <cstdint>
#include <vector>

#include "lbvh.h"

namespace wost {

struct HostCellGrid {
    bool valid = false;
    float ox = 0, oy = 0;       // world position of the corner of cell (0, 0)
    float h = 0, inv_h = 0;     // cell size
    int32_t nx = 0, ny = 0;
    int32_t n_chunks = 0;       // ceil(n_segs / 16)
    // Lists are stored in groups of four 16-bit chunk ids (8 bytes: one load), padded with the id
    // n_chunks -- a sentinel chunk whose box is infinitely far away.  cell_off[c] .. cell_off[c+1] are
    // the GROUPS of cell c = iy * nx + ix; cell nx*ny is the list of ALL chunks (queries outside the grid).
    std::vector<uint32_t> cell_off;
    std::vector<uint16_t> ids;
    // 8 floats per chunk, n_chunks + 1 entries (the last one is the sentinel): cx cy ux uy | hl hw 0 0:
    // oriented box of the chunk's segments, padded like the boxes of the tree (fit_obb)
    std::vector<float> chunk_box;
    // 80 floats per chunk (n_chunks + 1 entries): cx[16] cy[16] ux[16] uy[16] hl[16], the exact distance records of its
    // segments (the values the leaves of the tree hold); unused entries have cx = kFarCoord
    std::vector<float> chunk_seg;
    // 16 ints per chunk (n_chunks + 1 entries): slot of each segment in the tree's leaf order (-1 = unused entry): what a
    // closest-point query returns, and the index of segA / segInv / segCol / segOrig
    std::vector<int32_t> chunk_slot;
    // statistics of the build
    double mean_list = 0, max_list = 0;
};

// Builds the chunks and the lists for the tree `t` (a built LBVH).  The grid covers
// [lo, hi] (world bounding box of everything a walker can reach: geometry of both meshes and the
// evaluation frame) padded by two cells; max_cells bounds nx * ny.  Returns false (grid.valid =
// false) when the structure does not apply: no segments, more than 65 535 chunks, or an empty box.
bool build_cell_grid(const HostTree &t, const float lo[2], const float hi[2], int max_cells, HostCellGrid *grid);

}  // namespace wost
