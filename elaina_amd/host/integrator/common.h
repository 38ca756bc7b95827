// common.h -- what the integrators share on the host: the per-channel RGB buffers that stand in
// for the reference's Film objects (util/film.h), their export (exportImage / exportEnergy,
// reference integrator/common.h:126-239) and the SDF debug channels (:52-123).
#pragma once
#include <vector>

#include "../../../include/wost.h"
#include "core/common.h"
#include "util/image_io.h"

namespace elaina {

class IntegratorOutputs {
public:
    IntegratorOutputs(Vector2i frame, const fs::path &base) : frameSize_(frame), basePath(base) {}

    void exportImage(ExportImageChannel imageType, const string &file_name);
    void exportEnergy(ExportImageChannel imageType, ToneMapping tone, const string &file_name);
    const fs::path &get_basePath() const { return basePath; }
    // RGB per pixel of a channel (empty until that channel has been produced)
    const std::vector<float> &get_channel(ExportImageChannel c) const { return channels[(size_t)c]; }

protected:
    void render_sdf(wost_handle scene, int which_mesh, ExportImageChannel c);
    void set_gray_channel(ExportImageChannel c, const std::vector<float> &gray);   // a distance per pixel as an RGB channel

    Vector2i frameSize_;
    fs::path basePath;
    std::vector<float> channels[(size_t)ExportImageChannel::CHANNEL_COUNT];
};

void check_wost(int rc, const char *what);

// normalised energy x in [0, 1] -> colour (reference util/tonemapping.cuh)
void tone_map(ToneMapping tone, float x, float rgb[3]);

}  // namespace elaina
