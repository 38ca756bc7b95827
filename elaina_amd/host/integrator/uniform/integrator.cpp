#include "integrator.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <fstream>

namespace elaina {

UniformIntegratorSettings UniformIntegratorSettings::from_json(const json &j)
{
    UniformIntegratorSettings s;
    const auto fsz = json_get_or_throw<std::vector<int>>(j, "frameSize");
    if (fsz.size() != 2) throw std::runtime_error("frameSize must have 2 entries");
    s.frameSize = {fsz[0], fsz[1]};
    s.debugPixel = json_get_or_throw<unsigned>(j, "debugPixel");
    s.samplesPerPixel = json_get_or_throw<int>(j, "samplesPerPixel");
    s.maxWalkingDepth = json_get_or_throw<unsigned>(j, "maxWalkingDepth");
    s.saveSppMetricsDuration = json_get_or_throw<int>(j, "saveSppMetricsDuration");
    s.saveSppMetricsUntil = json_get_or_throw<int>(j, "saveSppMetricsUntil");
    s.saveTimeMetricsDuration = json_get_or_throw<int>(j, "saveTimeMetricsDuration");
    s.epsilonShell = json_get_or_throw<float>(j, "epsilonShell");
    return s;
}

static void check(int rc, const char *what)
{
    if (rc != WOST_OK) throw std::runtime_error(string(what) + " failed (" + std::to_string(rc) + "): " + wost_last_error());
}

UniformIntegrator<2>::UniformIntegrator(Problem<2> &problem_, const IntegratorSettings &settings, const fs::path &basePath_,
                                        int device)
    : problem(problem_), integratorSettings(settings), basePath(basePath_)
{
    if (settings.saveSppMetricsDuration > 0 || settings.saveTimeMetricsDuration > 0)
        ELAINA_LOG(Warning, "periodic metric dumps are not built (SURVEY.md 8f.4); ignoring save*Metrics* settings");
    const wost_scene_desc sd = problem.scene_desc(settings.frameSize.x, settings.frameSize.y);
    wost_settings st{settings.frameSize.x, settings.frameSize.y, settings.samplesPerPixel, (int32_t)settings.maxWalkingDepth,
                     settings.epsilonShell};
    check(wost_create(&sd, &st, device, &handle), "wost_create");
}

UniformIntegrator<2>::~UniformIntegrator()
{
    if (handle) wost_destroy(handle);
}

uint64_t UniformIntegrator<2>::solve()
{
    const auto start = std::chrono::high_resolution_clock::now();
    const int n = integratorSettings.frameSize.x * integratorSettings.frameSize.y;
    std::vector<float> &f = channels[(size_t)ExportImageChannel::SOLUTION];
    f.assign((size_t)n * 3, 0.0f);
    check(wost_solve(handle, 0, n, f.data(), &last_stats), "wost_solve");
    const auto end = std::chrono::high_resolution_clock::now();
    return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count();
}

static void gray_to_rgb(const std::vector<float> &g, std::vector<float> &rgb)
{
    rgb.resize(g.size() * 3);
    for (size_t i = 0; i < g.size(); ++i) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = g[i];
}

void UniformIntegrator<2>::renderDirichletSDF()
{
    const int n = integratorSettings.frameSize.x * integratorSettings.frameSize.y;
    std::vector<float> d((size_t)n);
    check(wost_render_sdf(handle, WOST_MESH_DIRICHLET, d.data()), "wost_render_sdf");
    gray_to_rgb(d, channels[(size_t)ExportImageChannel::DIRICHLET_SDF]);
}

void UniformIntegrator<2>::renderSilhouetteSDF()
{
    const int n = integratorSettings.frameSize.x * integratorSettings.frameSize.y;
    std::vector<float> d((size_t)n);
    check(wost_render_sdf(handle, WOST_MESH_NEUMANN, d.data()), "wost_render_sdf");
    gray_to_rgb(d, channels[(size_t)ExportImageChannel::NEUMANN_SDF]);
}

void UniformIntegrator<2>::renderSource()
{
    throw std::runtime_error("renderSource: the source term is outside this build's scope (SURVEY.md 8f.2)");
}

void UniformIntegrator<2>::queryNetwork(const VectorType &)
{
    throw std::runtime_error("queryNetwork: not implemented for the uniform integrator (reference integrator.cu:661-664)");
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

void UniformIntegrator<2>::exportImage(ExportImageChannel imageType, const string &file_name)
{
    const std::vector<float> &c = channels[(size_t)imageType];
    if (c.empty()) throw std::runtime_error(string("channel ") + channel_name(imageType) + " has not been produced");
    const int w = integratorSettings.frameSize.x, h = integratorSettings.frameSize.y;
    ELAINA_LOG(Info, "Exporting image to %s.pfm / .ppm", (basePath / file_name).string().c_str());
    write_pfm(basePath / (file_name + ".pfm"), w, h, c);
    write_ppm(basePath / (file_name + ".ppm"), w, h, c);
}

void UniformIntegrator<2>::exportEnergy(ExportImageChannel imageType, ToneMapping tone, const string &file_name)
{
    // reference util/film.h:107-145: luminance = dot(rgb, (0.299, 0.587, 0.114)), min/max normalisation
    const std::vector<float> &c = channels[(size_t)imageType];
    if (c.empty()) throw std::runtime_error(string("channel ") + channel_name(imageType) + " has not been produced");
    const int w = integratorSettings.frameSize.x, h = integratorSettings.frameSize.y;
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
