// wost_internal3.h -- host-side declarations shared by the translation units of the 3-D path (not part of the C-ABI): the
// uploaded mesh, the context behind wost3_handle, the error macro.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/wost.h"
#include "wost_device3.h"
#include "wost_internal.h"

namespace wost {

struct HostMesh3 {
    int32_t n_tris = 0, n_edges = 0, levels = 1, first_leaf = 1;
    bool emissive = false;
    std::vector<float> nodes, tri, colors, cones, slotEdges;
    float ext = 0.0f;                 // largest coordinate
    std::vector<int32_t> triOrig, triVerts, flatVerts;
    std::vector<float> obox;          // index-ordered run boxes (emissive meshes above the flat limit)
    int32_t obox_off[12] = {0}, obox_levels = 0;
    std::vector<DevTri> flat;
    std::vector<DevEdge3> edges;
};

struct DeviceMesh3 {
    DevMesh3 view{};
    HostMesh3 host;
    std::vector<void *> allocs;
};

template <class T>
static hipError_t upload3(std::vector<void *> &allocs, const T *src, size_t count, const T **dst)
{
    *dst = nullptr;
    if (count == 0) return hipSuccess;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) return e;
    allocs.push_back(p);
    *dst = reinterpret_cast<const T *>(p);
    return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace wost

// (the handle type of the C-ABI lives outside the namespace; its members are the namespace's)
struct wost3_context {
    int device = 0;
    wost_settings settings{};
    wost::DevSettings dst{};
    wost::DevProbe3 probe{};
    wost::DeviceMesh3 dm, nm;
    uint8_t *mask = nullptr;
    wost::DevSource3 src{};          // rgb owned by the context
    size_t n_pixels = 0;
    float *field = nullptr;
    wost::Stats3Dev *stats = nullptr;
    uint32_t *cursor = nullptr;
    int wait_weight = 32, trav_burst = 3;   // a sweep over both constants: a leaf visit (four exact triangle distances) is dear, steps are served early
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

#define W3_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return set_error(WOST_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace wost {
// wost_build3.hip: the mesh of `d` built on the current device (records, Morton order, tree, cones) into s.view
int upload_mesh3(const wost3_mesh_desc &d, DeviceMesh3 &s);
// wost_vmm3.hip: dL/draw of the 3-D mixture loss for n training samples, all pointers on the device (rows of 41 floats)
void launch_vmm3_loss_gradients(hipStream_t stream, const float *raw, const float *dir, const float *li, const float *dir_pdf,
                                const unsigned char *on_neumann, const float *normal, int n, float loss_scale, float *dl_draw, float *likelihood);
}  // namespace wost

// device scratch of the batch entry points (freed on every return path)
struct Scratch3 {
    std::vector<void *> ptrs;
    ~Scratch3()
    {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <class T>
    hipError_t alloc(T **p, size_t count)
    {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, count * sizeof(T) + 16);
        if (e == hipSuccess) ptrs.push_back(q);
        *p = reinterpret_cast<T *>(q);
        return e;
    }
};


