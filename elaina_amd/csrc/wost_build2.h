// wost_build2.h -- the 2-D segment tree built on the device (wost_build2.hip).  Not part of the C-ABI.
#pragma once

#include <cstddef>

#include "../../include/wost.h"
#include "wost_device.h"

namespace wost {

// the uploaded tree of one boundary mesh: `view` points into ONE device allocation (`alloc`, `bytes`) that the caller owns
struct DeviceTree2 {
    DevMesh view;
    void *alloc = nullptr;
    size_t bytes = 0;
};

// Problem<2>::build_bvh on the current device: segment records, Morton order, the refined leaf assignment, oriented child boxes,
// normal cones and the scan copies of `d`, the same bits as lbvh_build.cpp's build_tree + the uploads of upload_mesh
// (view.obox / lens / sampBox -- the run boxes of an emissive boundary on the tree -- are left to the caller).
// WOST_OK, or an error recorded with set_error.
int build_tree_device(const wost_mesh_desc &d, DeviceTree2 &out);

}  // namespace wost
