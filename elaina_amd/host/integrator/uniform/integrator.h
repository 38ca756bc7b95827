// integrator.h -- host mirror of the reference's UniformIntegrator<2> (integrator/uniform/
// integrator.h:55-131): same constructor, same public methods run_expr dispatches over
// (exec.cu:145-215), same settings struct.  All device work goes through the C-ABI
// (include/wost.h); this class owns a wost_handle instead of queues and films.
#pragma once
#include <vector>

#include "../../../../include/wost.h"
#include "core/common.h"
#include "core/problem.h"
#include "integrator/common.h"

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

template <> class UniformIntegrator<2> : public IntegratorOutputs {
public:
    using IntegratorSettings = UniformIntegratorSettings;
    using VectorType = Vector2f;
    using ProblemType = Problem<2>;

    UniformIntegrator(Problem<2> &problem, const IntegratorSettings &settings, const fs::path &basePath_, int device = 0);
    ~UniformIntegrator();
    UniformIntegrator(const UniformIntegrator &) = delete;
    UniformIntegrator &operator=(const UniformIntegrator &) = delete;

    uint64_t solve();  // wall milliseconds, like the reference (integrator.cu:666-672)
    void renderDirichletSDF();
    void renderSilhouetteSDF();
    void renderSource();
    void queryNetwork(const VectorType &p);

    const IntegratorSettings &get_integratorSettings() const { return integratorSettings; }
    const Problem<2> &get_problem() const { return problem; }
    const wost_stats &get_last_stats() const { return last_stats; }

private:
    Problem<2> &problem;
    IntegratorSettings integratorSettings;
    wost_handle handle{nullptr};
    wost_stats last_stats{};
};

// UniformIntegrator<3> (reference integrator/uniform/integrator.h:55-131 with DIM = 3)
template <> class UniformIntegrator<3> : public IntegratorOutputs {
public:
    using IntegratorSettings = UniformIntegratorSettings;
    using VectorType = Vector3f;
    using ProblemType = Problem<3>;

    UniformIntegrator(Problem<3> &problem, const IntegratorSettings &settings, const fs::path &basePath_, int device = 0);
    ~UniformIntegrator();
    UniformIntegrator(const UniformIntegrator &) = delete;
    UniformIntegrator &operator=(const UniformIntegrator &) = delete;

    uint64_t solve();
    void renderDirichletSDF();
    void renderSilhouetteSDF();
    void renderSource();
    void queryNetwork(const VectorType &p);
    const wost_stats &get_last_stats() const { return last_stats; }

private:
    Problem<3> &problem;
    IntegratorSettings integratorSettings;
    wost3_handle handle{nullptr};
    wost_stats last_stats{};
};

}  // namespace elaina
