// integrator.h -- host mirror of the reference's GuidedIntegrator<2> (integrator/guided/
// integrator.h:77-256): same constructor + resetNetwork(json) pair run_expr uses
// (exec.cu:91-96), same public methods.  All device work goes through the C-ABI
// (include/wost.h, wost_guided_* and wost_net_*).
#pragma once
#include <vector>

#include "../../../../include/wost.h"
#include "core/common.h"
#include "core/problem.h"
#include "integrator/common.h"

namespace elaina {

// reference integrator/guided/integrator.h:54-75 -- every key is required in the JSON
struct GuidedIntegratorSettings {
    Vector2i frameSize{800, 800};
    unsigned debugPixel{0};
    int samplesPerPixel{512};
    unsigned trainSppCount{150};
    float uniformFractionInTrainingPhase{0.5f};
    float uniformFractionInGuidingPhase{0.5f};
    unsigned maxGuidedDepthInTrainingPhase{10};
    unsigned maxGuidedDepthInGuidingPhase{10};
    unsigned maxWalkingDepth{32};
    int saveSppMetricsDuration{-1};
    int saveSppMetricsUntil{1024};
    int saveTimeMetricsDuration{-1};
    float epsilonShell{1e-5f};

    static GuidedIntegratorSettings from_json(const json &j);
};

// "network" section of the configuration (reference data/ladybug/n.json:49-81) -> wost_net_config
wost_net_config network_config_from_json(const json &network_section);

template <unsigned int DIM> class GuidedIntegrator;

template <> class GuidedIntegrator<2> : public IntegratorOutputs {
public:
    using IntegratorSettings = GuidedIntegratorSettings;
    using VectorType = Vector2f;
    using ProblemType = Problem<2>;

    GuidedIntegrator(Problem<2> &problem, const IntegratorSettings &settings, const fs::path &basePath_, int device = 0);
    ~GuidedIntegrator();
    GuidedIntegrator(const GuidedIntegrator &) = delete;
    GuidedIntegrator &operator=(const GuidedIntegrator &) = delete;

    void resetNetwork(const json &config);   // builds the device objects (reference integrator.cu:1095-1131)
    uint64_t solve();                        // wall milliseconds (reference integrator.cu:1189-1195)
    void renderDirichletSDF();
    void renderSilhouetteSDF();
    void renderSource();
    void queryNetwork(const VectorType &p);  // logs the mixture at p (reference integrator.cu:566-615)

    const IntegratorSettings &get_integratorSettings() const { return integratorSettings; }
    const Problem<2> &get_problem() const { return problem; }
    const wost_guided_stats &get_last_stats() const { return last_stats; }

private:
    wost_handle scene_handle();
    Problem<2> &problem;
    IntegratorSettings integratorSettings;
    int device;
    wost_guided_handle handle{nullptr};
    wost_guided_stats last_stats{};
};

// GuidedIntegrator<3> (reference integrator/guided/integrator.h:77-256 with DIM = 3, dispatched by exec.cu:102-122): the same
// surface on Problem<3>; 41 network outputs (guided/parameters.h:26-33), queryNetwork(Vector3f) (exec.cu:175-186).
template <> class GuidedIntegrator<3> : public IntegratorOutputs {
public:
    using IntegratorSettings = GuidedIntegratorSettings;
    using VectorType = Vector3f;
    using ProblemType = Problem<3>;

    GuidedIntegrator(Problem<3> &problem, const IntegratorSettings &settings, const fs::path &basePath_, int device = 0);
    ~GuidedIntegrator();
    GuidedIntegrator(const GuidedIntegrator &) = delete;
    GuidedIntegrator &operator=(const GuidedIntegrator &) = delete;

    void resetNetwork(const json &config);
    uint64_t solve();
    void renderDirichletSDF();
    void renderSilhouetteSDF();
    void renderSource();
    void queryNetwork(const VectorType &p);
    const wost_guided_stats &get_last_stats() const { return last_stats; }

private:
    wost3_handle scene_handle();
    Problem<3> &problem;
    IntegratorSettings integratorSettings;
    int device;
    wost3_guided_handle handle{nullptr};
    wost_guided_stats last_stats{};
};

}  // namespace elaina
