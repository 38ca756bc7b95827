#include "common.h"

#include "util/colormap_tables.h"

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

void IntegratorOutputs::set_gray_channel(ExportImageChannel c, const std::vector<float> &gray) { gray_to_rgb(gray, channels[(size_t)c]); }

void IntegratorOutputs::render_sdf(wost_handle scene, int which_mesh, ExportImageChannel c)
{
    std::vector<float> d((size_t)frameSize_.x * frameSize_.y);
    check_wost(wost_render_sdf(scene, which_mesh, d.data()), "wost_render_sdf");
    gray_to_rgb(d, channels[(size_t)c]);
}

// ---- colormaps of saveEnergy (reference util/film.h:107-145, util/tonemapping.cuh) ------------
// MATLAB_JET is the reference's piecewise-linear formula; MATLAB_PARULA and IDL_RDBU evaluate the
// reference's fitted tables (util/tonemapping.cuh:53-383, :385-480), restated as numbers in
// util/colormap_tables.h, with the reference's arithmetic: parula in float Horner form on
// dx = float(x - x0), RdBu in double Horner form, / 255, clamped.
static double rdbu_channel(const RdBuPiece *rows, double x)
{
    const RdBuPiece *r = rows;
    while (!(x < r->hi)) ++r;          // the last row has hi = 2
    double v = r->c[0];
    for (int k = 1; k < 6; ++k) v = v * x + r->c[k];
    return v;
}

void tone_map(ToneMapping tone, float x, float rgb[3])
{
    auto clamp01 = [](float v) { return std::min(std::max(v, 0.0f), 1.0f); };
    switch (tone) {
    case ToneMapping::MATLAB_JET:
        rgb[0] = clamp01((float)(x < 0.7 ? 4.0 * x - 1.5 : -4.0 * x + 4.5));
        rgb[1] = clamp01((float)(x < 0.5 ? 4.0 * x - 0.5 : -4.0 * x + 3.5));
        rgb[2] = clamp01((float)(x < 0.3 ? 4.0 * x + 0.5 : -4.0 * x + 2.5));
        break;
    case ToneMapping::MATLAB_PARULA: {
        if (x < 0.0 || 1.0 < x || std::isnan(x)) {
            rgb[0] = rgb[1] = rgb[2] = 0.0f;
            break;
        }
        const ParulaPiece *p = kParula;
        while (!(x < p->hi)) ++p;      // the last piece has hi = 2
        const float dx = (float)((double)x - p->x0);
        for (int c = 0; c < 3; ++c) rgb[c] = ((p->c3[c] * dx + p->c2[c]) * dx + p->c1[c]) * dx + p->c0[c];
        break;
    }
    case ToneMapping::IDL_RDBU: {
        const double xd = std::isnan(x) ? 0.0 : (double)x;
        rgb[0] = clamp01((float)(rdbu_channel(kRdBuRed, xd) / 255.0));
        rgb[1] = clamp01((float)(rdbu_channel(kRdBuGreen, xd) / 255.0));
        rgb[2] = clamp01((float)(rdbu_channel(kRdBuBlue, xd) / 255.0));
        break;
    }
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
