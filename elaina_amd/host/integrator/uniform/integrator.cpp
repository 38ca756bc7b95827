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

UniformIntegrator<2>::UniformIntegrator(Problem<2> &problem_, const IntegratorSettings &settings, const fs::path &basePath_,
                                        int device)
    : IntegratorOutputs(settings.frameSize, basePath_), problem(problem_), integratorSettings(settings)
{
    const wost_scene_desc sd = problem.scene_desc(settings.frameSize.x, settings.frameSize.y);
    wost_settings st{settings.frameSize.x, settings.frameSize.y, settings.samplesPerPixel, (int32_t)settings.maxWalkingDepth,
                     settings.epsilonShell};
    check_wost(wost_create(&sd, &st, device, &handle), "wost_create");
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
    // saveSppMetrics* / saveTimeMetrics* (reference integrator.cu:578-609): frames/<sampleId>.exr|png and
    // frames_time/<elapsed ms>.exr|png = solution / (sampleId + 1) after sample sampleId.  The pixels of this
    // integrator do not advance in lock step, but a pixel's first k samples do not depend on the total: each
    // frame is a solve with spp = k (extra work, a debug feature); the time frames are keyed by the milliseconds
    // since the start of solve(), as in the reference.
    const IntegratorSettings &s = integratorSettings;
    if (s.saveSppMetricsDuration > 0 || s.saveTimeMetricsDuration > 0) {
        if (s.saveSppMetricsDuration > 0) fs::create_directories(basePath / "frames");
        if (s.saveTimeMetricsDuration > 0) fs::create_directories(basePath / "frames_time");
        for (int sampleId = 0; sampleId < s.samplesPerPixel; ++sampleId) {
            const bool by_spp = s.saveSppMetricsDuration > 0 && sampleId % s.saveSppMetricsDuration == 0 && sampleId < s.saveSppMetricsUntil;
            const bool by_time = s.saveTimeMetricsDuration > 0 && sampleId % s.saveTimeMetricsDuration == 0;
            if (!by_spp && !by_time) continue;
            check_wost(wost_set_option(handle, "spp", sampleId + 1), "wost_set_option(spp)");
            check_wost(wost_solve(handle, 0, n, f.data(), &last_stats), "wost_solve");
            if (by_spp) {
                write_exr(basePath / "frames" / (std::to_string(sampleId) + ".exr"), s.frameSize.x, s.frameSize.y, f);
                write_png(basePath / "frames" / (std::to_string(sampleId) + ".png"), s.frameSize.x, s.frameSize.y, f);
            }
            if (by_time) {
                const auto elapsed = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - start).count();
                write_exr(basePath / "frames_time" / (std::to_string(elapsed) + ".exr"), s.frameSize.x, s.frameSize.y, f);
                write_png(basePath / "frames_time" / (std::to_string(elapsed) + ".png"), s.frameSize.x, s.frameSize.y, f);
            }
        }
        check_wost(wost_set_option(handle, "spp", s.samplesPerPixel), "wost_set_option(spp)");
    }
    check_wost(wost_solve(handle, 0, n, f.data(), &last_stats), "wost_solve");
    const auto end = std::chrono::high_resolution_clock::now();
    return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count();
}

void UniformIntegrator<2>::renderDirichletSDF() { render_sdf(handle, WOST_MESH_DIRICHLET, ExportImageChannel::DIRICHLET_SDF); }

void UniformIntegrator<2>::renderSilhouetteSDF() { render_sdf(handle, WOST_MESH_NEUMANN, ExportImageChannel::NEUMANN_SDF); }

void UniformIntegrator<2>::renderSource()
{
    std::vector<float> &c = channels[(size_t)ExportImageChannel::SOURCE];
    c.assign((size_t)frameSize_.x * frameSize_.y * 3, 0.0f);
    check_wost(wost_render_source(handle, c.data()), "wost_render_source");
}

UniformIntegrator<3>::UniformIntegrator(Problem<3> &problem_, const IntegratorSettings &settings, const fs::path &basePath_, int device)
    : IntegratorOutputs(settings.frameSize, basePath_), problem(problem_), integratorSettings(settings)
{
    const wost3_scene_desc sd = problem.scene_desc(settings.frameSize.x, settings.frameSize.y);
    wost_settings st{settings.frameSize.x, settings.frameSize.y, settings.samplesPerPixel, (int32_t)settings.maxWalkingDepth,
                     settings.epsilonShell};
    check_wost(wost3_create(&sd, &st, device, &handle), "wost3_create");
}

void UniformIntegrator<3>::renderDirichletSDF()
{
    std::vector<float> d((size_t)frameSize_.x * frameSize_.y);
    check_wost(wost3_render_sdf(handle, WOST_MESH_DIRICHLET, d.data()), "wost3_render_sdf");
    set_gray_channel(ExportImageChannel::DIRICHLET_SDF, d);
}

void UniformIntegrator<3>::renderSilhouetteSDF()
{
    std::vector<float> d((size_t)frameSize_.x * frameSize_.y);
    check_wost(wost3_render_sdf(handle, WOST_MESH_NEUMANN, d.data()), "wost3_render_sdf");
    set_gray_channel(ExportImageChannel::NEUMANN_SDF, d);
}

void UniformIntegrator<3>::renderSource()
{
    std::vector<float> &c = channels[(size_t)ExportImageChannel::SOURCE];
    c.assign((size_t)frameSize_.x * frameSize_.y * 3, 0.0f);
    check_wost(wost3_render_source(handle, c.data()), "wost3_render_source");
}

UniformIntegrator<3>::~UniformIntegrator()
{
    if (handle) wost3_destroy(handle);
}

uint64_t UniformIntegrator<3>::solve()
{
    const auto start = std::chrono::high_resolution_clock::now();
    const int n = integratorSettings.frameSize.x * integratorSettings.frameSize.y;
    std::vector<float> &f = channels[(size_t)ExportImageChannel::SOLUTION];
    f.assign((size_t)n * 3, 0.0f);
    check_wost(wost3_solve(handle, 0, n, f.data(), &last_stats), "wost3_solve");
    return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - start).count();
}

void UniformIntegrator<3>::queryNetwork(const VectorType &)
{
    throw std::runtime_error("queryNetwork: not implemented for the uniform integrator (reference integrator.cu:661-664)");
}

void UniformIntegrator<2>::queryNetwork(const VectorType &)
{
    throw std::runtime_error("queryNetwork: not implemented for the uniform integrator (reference integrator.cu:661-664)");
}

}  // namespace elaina
