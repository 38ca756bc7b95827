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

void write_pfm(const fs::path &path, int width, int height, const std::vector<float> &rgb)
{
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot write " + path.string());
    f << "PF\n" << width << " " << height << "\n-1.0\n";  // little endian, rows top to bottom as stored
    f.write(reinterpret_cast<const char *>(rgb.data()), (std::streamsize)(rgb.size() * sizeof(float)));
}

void write_ppm(const fs::path &path, int width, int height, const std::vector<float> &rgb)
{
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot write " + path.string());
    f << "P6\n" << width << " " << height << "\n255\n";
    std::vector<unsigned char> px(rgb.size());
    for (size_t i = 0; i < rgb.size(); ++i) {
        const float v = std::isfinite(rgb[i]) ? std::min(std::max(rgb[i], 0.0f), 1.0f) : 0.0f;
        px[i] = (unsigned char)(v * 255.0f + 0.5f);
    }
    f.write(reinterpret_cast<const char *>(px.data()), (std::streamsize)px.size());
}

void IntegratorOutputs::exportImage(ExportImageChannel imageType, const string &file_name)
{
    const std::vector<float> &c = channels[(size_t)imageType];
    if (c.empty()) throw std::runtime_error(string("channel ") + channel_name(imageType) + " has not been produced");
    const int w = frameSize_.x, h = frameSize_.y;
    ELAINA_LOG(Info, "Exporting image to %s.pfm / .ppm", (basePath / file_name).string().c_str());
    write_pfm(basePath / (file_name + ".pfm"), w, h, c);
    write_ppm(basePath / (file_name + ".ppm"), w, h, c);
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
    if (tone != ToneMapping::NONE)
        for (float &v : e) v = (v - mn) / span;
    if (tone != ToneMapping::NONE && tone != ToneMapping::NONE_NORMALIZED)
        ELAINA_LOG(Warning, "colormaps are not built (SURVEY.md 8f.1): writing the normalised energy as grey");
    std::vector<float> rgb;
    gray_to_rgb(e, rgb);
    write_pfm(basePath / (file_name + ".pfm"), w, h, rgb);
    write_ppm(basePath / (file_name + ".ppm"), w, h, rgb);
}

}  // namespace elaina
