#include "common.h"

#include <algorithm>
#include <cmath>
#include <fstream>

namespace elaina {

void check_wost(int rc, const char *what)
{
    if (rc != WOST_OK) throw std::runtime_error(string(what) + " failed (" + std::to_string(rc) + "): " + wost_last_error());
}

static void gray_to_rgb(const std::vector<float> &g, std::vector<float> &rgb)
{
    rgb.resize(g.size() * 3);
    for (size_t i = 0; i < g.size(); ++i) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = g[i];
}

void IntegratorOutputs::render_sdf(wost_handle scene, int which_mesh, ExportImageChannel c)
{
    std::vector<float> d((size_t)frameSize_.x * frameSize_.y);
    check_wost(wost_render_sdf(scene, which_mesh, d.data()), "wost_render_sdf");
    gray_to_rgb(d, channels[(size_t)c]);
}

// ---- colormaps of saveEnergy (reference util/film.h:107-145, util/tonemapping.cuh) ------------
// MATLAB_JET is the reference's piecewise-linear formula.  The reference's MATLAB_PARULA and
// IDL_RDBU are fitted polynomial tables (several hundred coefficients); here they are linear
// interpolations through published anchor colours of the same maps (ColorBrewer RdBu-11; nine
// samples of parula), within a few 1/255 of the originals -- previews, not parity data.
static void lerp_anchors(const float (*a)[3], int n, float x, float rgb[3])
{
    x = std::min(std::max(x, 0.0f), 1.0f) * (float)(n - 1);
    const int i = std::min((int)x, n - 2);
    const float t = x - (float)i;
    for (int c = 0; c < 3; ++c) rgb[c] = a[i][c] * (1.0f - t) + a[i + 1][c] * t;
}

void tone_map(ToneMapping tone, float x, float rgb[3])
{
    static const float rdbu[11][3] = {{103, 0, 31}, {178, 24, 43}, {214, 96, 77}, {244, 165, 130}, {253, 219, 199},
                                      {247, 247, 247}, {209, 229, 240}, {146, 197, 222}, {67, 147, 195}, {33, 102, 172},
                                      {5, 48, 97}};
    static const float parula[9][3] = {{0.2422f, 0.1504f, 0.6603f}, {0.2810f, 0.3228f, 0.9579f}, {0.1786f, 0.5289f, 0.9682f},
                                       {0.0689f, 0.6948f, 0.8394f}, {0.2161f, 0.7843f, 0.5923f}, {0.6720f, 0.7793f, 0.2227f},
                                       {0.9970f, 0.7659f, 0.2199f}, {0.9632f, 0.9000f, 0.1300f}, {0.9769f, 0.9839f, 0.0805f}};
    auto clamp01 = [](float v) { return std::min(std::max(v, 0.0f), 1.0f); };
    switch (tone) {
    case ToneMapping::MATLAB_JET:
        rgb[0] = clamp01(x < 0.7f ? 4.0f * x - 1.5f : -4.0f * x + 4.5f);
        rgb[1] = clamp01(x < 0.5f ? 4.0f * x - 0.5f : -4.0f * x + 3.5f);
        rgb[2] = clamp01(x < 0.3f ? 4.0f * x + 0.5f : -4.0f * x + 2.5f);
        break;
    case ToneMapping::MATLAB_PARULA:
        lerp_anchors(parula, 9, x, rgb);
        break;
    case ToneMapping::IDL_RDBU:
        lerp_anchors(rdbu, 11, x, rgb);
        for (int c = 0; c < 3; ++c) rgb[c] /= 255.0f;
        break;
    default:
        rgb[0] = rgb[1] = rgb[2] = x;
        break;
    }
}

static void save_all(const fs::path &base, const string &file_name, int w, int h, const std::vector<float> &rgb)
{
    // .exr and .png like the reference (integrator/common.h:197-201), .pfm = the raw fp32 field
    ELAINA_LOG(Info, "Exporting image to %s.exr / .png / .pfm", (base / file_name).string().c_str());
    write_exr(base / (file_name + ".exr"), w, h, rgb);
    write_png(base / (file_name + ".png"), w, h, rgb);
    write_pfm(base / (file_name + ".pfm"), w, h, rgb);
}

void IntegratorOutputs::exportImage(ExportImageChannel imageType, const string &file_name)
{
    const std::vector<float> &c = channels[(size_t)imageType];
    if (c.empty()) throw std::runtime_error(string("channel ") + channel_name(imageType) + " has not been produced");
    save_all(basePath, file_name, frameSize_.x, frameSize_.y, c);
}

void IntegratorOutputs::exportEnergy(ExportImageChannel imageType, ToneMapping tone, const string &file_name)
{
    // reference util/film.h:107-145: luminance = dot(rgb, (0.299, 0.587, 0.114)), min/max normalisation
    const std::vector<float> &c = channels[(size_t)imageType];
    if (c.empty()) throw std::runtime_error(string("channel ") + channel_name(imageType) + " has not been produced");
    const int w = frameSize_.x, h = frameSize_.y;
    std::vector<float> e((size_t)w * h);
    float mn = INFINITY, mx = -INFINITY;
    for (size_t i = 0; i < e.size(); ++i) {
        e[i] = c[3 * i] * 0.299f + c[3 * i + 1] * 0.587f + c[3 * i + 2] * 0.114f;
        mn = std::min(mn, e[i]);
        mx = std::max(mx, e[i]);
    }
    const float span = mx - mn;
    if (std::isnan(mn) || std::isnan(mx) || span == 0.0f)
        ELAINA_LOG(Warning, "Invalid min/max values for tone mapping: min = %f, max = %f", mn, mx);
    std::vector<float> rgb(e.size() * 3);
    for (size_t i = 0; i < e.size(); ++i) {
        if (tone == ToneMapping::NONE) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = e[i];
        else tone_map(tone, (e[i] - mn) / span, &rgb[3 * i]);
    }
    save_all(basePath, file_name, w, h, rgb);
}

}  // namespace elaina
