// lbvh.h -- host-side builder of the wide LBVH the HIP kernels traverse.
//
// Replaces (does not port) lbvh::scene<2>::build_bvh() / compute_silhouettes() of
// the absent third-party snch-lbvh (reference call sites core/problem.cu:31-37,48-54).
//
// Layout, designed for per-lane gathers on gfx950 rather than for a pointer-chasing
// binary tree:
//   * segments are sorted by the 32-bit Morton code of their centroid;
//   * a leaf is 4 consecutive sorted segments;
//   * the hierarchy is an IMPLICIT complete 4-ary tree in heap order: node g has the
//     children 4g+1..4g+4, node 0 is the root, leaves are the nodes of level `levels`.
//     No child pointers are stored.
//   * every node, leaves included, stores its four children in ONE format: an oriented
//     box {centre, unit axis, half length, half width}, 6 x float4 = 96 bytes per node.  For
//     the children of a leaf -- the segments -- the half width is 0 and the box distance IS
//     the segment distance of the arithmetic contract, so one instruction stream serves the
//     whole traversal (no inner/leaf divergence inside a wave).  Oriented boxes hug the
//     thin diagonal polyline pieces of diffusion-curve scenes far better than axis-aligned
//     ones: ~40 % fewer node visits per query on the shipped scenes.
// Query results do not depend on this layout: the closest-point query returns the
// minimum distance with ties broken by the lowest ORIGINAL segment index, and node
// boxes are padded so that pruning can never cut a segment that ties or wins.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#ifndef WOST_NODE_FLOATS
#define WOST_NODE_FLOATS 24      // see wost_device.h
#endif

namespace wost {

constexpr int kLeafSize = 4;
constexpr int kArity = 4;
constexpr int kFarIndex = 0x7fffffff;
constexpr float kFarCoord = 1.0e18f;

// one boundary segment in ORIGINAL order, with everything the flat Neumann loops need
struct FlatSeg {
    float ax, ay, ex, ey;
    float inv_len2, len, nx, ny;  // unit normal (e.y, -e.x)/|e|
    float cx, cy, ux, uy;         // distance form: centre, unit axis ...
    float hl, pad0, pad1, pad2;   // ... and half length (64-byte record)
};

// silhouette candidate, one per mesh vertex (indexed by vertex id)
struct SilVertex {
    float x, y;
    int32_t prev, next;  // segment ending / starting at this vertex, -1 if none
};

struct HostTree {
    int32_t n_segs = 0;
    int32_t n_verts = 0;
    int32_t levels = 0;        // >= 1 when n_segs > 0
    int32_t first_leaf = 0;    // heap index of leaf 0 = (4^levels - 1) / 3
    int32_t n_leaves_cap = 0;  // 4^levels
    int32_t n_leaves = 0;      // leaves holding at least one real segment
    float pad = 0.0f;
    double obb_pad = 0.0;      // absolute padding of the oriented boxes
    std::vector<float> boxes;     // 4 floats per node g >= 1, stored at [g-1]: lox, loy, hix, hiy
    // 24 floats per node g in [0, first_leaf + n_leaves_cap): the four children as oriented
    // boxes, SoA: cx[4] cy[4] ux[4] uy[4] hl[4] hw[4].  Children of the nodes of the last
    // level are the segments themselves (hw = 0, no padding: that distance is exact).
    std::vector<float> nodes;
    // 20 floats per node: the SNCH normal cones of the four children, SoA:
    // axis.x[4] axis.y[4] cos(half)[4] sin(half)[4] radius[4].  The cone of a child covers the
    // normals of every segment incident to a vertex of its subtree; cos(half) <= 0 marks
    // "cannot prune" (open polyline end inside, or normals spread over >= 90 degrees).  The
    // radius bounds the subtree's vertices around the child's box centre.
    std::vector<float> cones;
    std::vector<int32_t> segVerts; // 2 ints per slot: the segment's vertex ids (-1 padding)
    std::vector<float> segA;      // 4 floats per slot
    std::vector<float> segInv;    // 1 float per slot
    std::vector<int32_t> segOrig; // original index per slot (kFarIndex for padding)
    std::vector<float> segCol;    // 12 floats per slot: left(i0) left(i1) right(i0) right(i1)
    std::vector<int32_t> origToSlot;
    std::vector<FlatSeg> flat;    // original order
    std::vector<float> flatCol;   // 12 floats per original segment
    std::vector<SilVertex> sil;   // one per mesh vertex (prev = next = -1: no incident segment)
    float aabb[4] = {0, 0, 0, 0}; // lox, loy, hix, hiy of the mesh
};

// Oriented box {cx cy ux uy hl hw} around n points (x y pairs) with the moments `sums` (lbvh_fit.h), padded by obb_pad (lbvh_build.cpp).
struct FitSums;
void fit_obb(const float *P, size_t n, const FitSums &sums, double obb_pad, float out[6]);

// Returns 0 on success, negative on invalid input (index out of range).
// refine: true = perimeter-weighted (SAH) top-down assignment of the Morton-ordered segments
// to the leaves of the implicit tree, false = plain LBVH (consecutive Morton groups).
// extra_levels: grow the implicit tree beyond the minimum depth (more split freedom).
int build_tree(int32_t n_verts, const float *verts, int32_t n_segs, const int32_t *segs,
               const float *colors, HostTree *out, bool refine = true, int extra_levels = 0);

}  // namespace wost
