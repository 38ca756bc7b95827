// wost_internal.h -- declarations shared by the translation units of libwost_hip.so
// (wost_hip.hip, wost_vmm.hip, wost_net.hip, wost_guided.hip).  Not part of the C-ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/wost.h"
#include "wost_device.h"

namespace wost {

// records the message returned by wost_last_error() on this thread; returns `code`
int set_error(int code, const std::string &msg);

// What the guided integrator needs from a wost_context (wost_hip.hip): the uploaded scene.
struct SceneView {
    int device;
    DevMesh dm, nm;
    DevSettings st;
    DevProbe probe;
    const uint8_t *mask;   // device, width*height bytes or nullptr
    hipStream_t stream;
    DevSource src;         // rgb == nullptr: no source term
};
SceneView scene_view(wost_handle h);

// wost_vmm.hip: dL/draw of the mixture loss for n training samples, all pointers on the device
// (raw and dl_draw have a stride of 33 floats)
void launch_vmm_loss_gradients(hipStream_t stream, const float *raw, const float *dir, const float *li,
                               const float *dir_pdf, const uint8_t *on_neumann, const float *normal, int n,
                               float loss_scale, float *dl_draw, float *likelihood);

// wost_net.hip: the network on device pointers.  count_dev (optional) overrides max_n with a
// queue size that lives on the device; max_n only sizes the launch.  feature_stride = 0: outputs
// as rows of n_output floats per point; > 0: output o of point p at out[o * feature_stride + p].
int net_inference_dev(wost_net_handle h, const float *xy_dev, const uint32_t *count_dev, int max_n, float *out_dev,
                      bool use_inference_params, hipStream_t stream, size_t feature_stride = 0);
int net_forward_train_dev(wost_net_handle h, const float *xy_dev, int n, hipStream_t stream, float **out_dev,
                          float **dl_dev);
int net_backward_update_dev(wost_net_handle h, const float *xy_dev, int n, float loss_scale, int apply_update,
                            hipStream_t stream);
int net_apply_update_dev(wost_net_handle h, float loss_scale, hipStream_t stream);
void *net_gradient_buffer(wost_net_handle h, uint64_t *count);   // int64 fixed point, device
int net_optimizer_steps(const wost_net *h);
// kernels and fills the *_dev entry points above have issued on this network so far
uint64_t net_launch_count(const wost_net *h);
// shared-network mode: the summed gradient of `ranks` ranks is divided by their number before the step
void net_set_gradient_divisor(wost_net_handle h, float ranks);
int net_n_output(const wost_net *h);

}  // namespace wost
