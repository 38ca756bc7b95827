// integrator.h -- host mirror of the reference's UniformIntegrator<2> (integrator/uniform/
// integrator.h:55-131): same constructor, same public methods run_expr dispatches over
// (exec.cu:145-215), same settings struct.  All device work goes through the C-ABI
// (include/wost.h); this class owns a wost_handle instead of queues and films.
#pragma once
#include <vector>

#include "../../../../include/wost.h"
#include "core/common.h"
#include "core/problem.h"

namespace elaina {

// reference integrator/uniform/integrator.h:27-48 -- every key is required in the JSON
struct UniformIntegratorSettings {
    Vector2i frameSize{800, 800};
    unsigned debugPixel{0};
    int samplesPerPixel{512};
    unsigned maxWalkingDepth{32};
    int saveSppMetricsDuration{-1};
    int saveSppMetricsUntil{1024};
    int saveTimeMetricsDuration{-1};
    float epsilonShell{1e-5f};

    static UniformIntegratorSettings from_json(const json &j);
};

template <unsigned int DIM> class UniformIntegrator;

template <> class UniformIntegrator<2> {
public:
    using IntegratorSettings = UniformIntegratorSettings;
    using VectorType = Vector2f;
    using ProblemType = Problem<2>;

    UniformIntegrator(Problem<2> &problem, const IntegratorSettings &settings, const fs::path &basePath_, int device = 0);
    ~UniformIntegrator();
    UniformIntegrator(const UniformIntegrator &) = delete;
    UniformIntegrator &operator=(const UniformIntegrator &) = delete;

    uint64_t solve();  // wall milliseconds, like the reference (integrator.cu:666-672)
    void exportImage(ExportImageChannel imageType, const string &file_name);
    void exportEnergy(ExportImageChannel imageType, ToneMapping tone, const string &file_name);
    void renderDirichletSDF();
    void renderSilhouetteSDF();
    void renderSource();
    void queryNetwork(const VectorType &p);

    const IntegratorSettings &get_integratorSettings() const { return integratorSettings; }
    const fs::path &get_basePath() const { return basePath; }
    const Problem<2> &get_problem() const { return problem; }
    const wost_stats &get_last_stats() const { return last_stats; }
    // RGB per pixel of a channel (empty until that channel has been produced)
    const std::vector<float> &get_channel(ExportImageChannel c) const { return channels[(size_t)c]; }

private:
    Problem<2> &problem;
    IntegratorSettings integratorSettings;
    fs::path basePath;
    wost_handle handle{nullptr};
    wost_stats last_stats{};
    std::vector<float> channels[(size_t)ExportImageChannel::CHANNEL_COUNT];
};

// image writers for the raw field: binary PFM (fp32, what parity is measured on) and an
// 8-bit PPM preview.  EXR/PNG and the colormaps are "next" rows.
void write_pfm(const fs::path &path, int width, int height, const std::vector<float> &rgb);
void write_ppm(const fs::path &path, int width, int height, const std::vector<float> &rgb);

}  // namespace elaina
