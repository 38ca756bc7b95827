#include "integrator.h"

#include <chrono>
#include <cmath>

namespace elaina {

GuidedIntegratorSettings GuidedIntegratorSettings::from_json(const json &j)
{
    GuidedIntegratorSettings s;
    const auto fsz = json_get_or_throw<std::vector<int>>(j, "frameSize");
    if (fsz.size() != 2) throw std::runtime_error("frameSize must have 2 entries");
    s.frameSize = {fsz[0], fsz[1]};
    s.debugPixel = json_get_or_throw<unsigned>(j, "debugPixel");
    s.samplesPerPixel = json_get_or_throw<int>(j, "samplesPerPixel");
    s.trainSppCount = json_get_or_throw<unsigned>(j, "trainSppCount");
    s.uniformFractionInTrainingPhase = json_get_or_throw<float>(j, "uniformFractionInTrainingPhase");
    s.uniformFractionInGuidingPhase = json_get_or_throw<float>(j, "uniformFractionInGuidingPhase");
    s.maxGuidedDepthInTrainingPhase = json_get_or_throw<unsigned>(j, "maxGuidedDepthInTrainingPhase");
    s.maxGuidedDepthInGuidingPhase = json_get_or_throw<unsigned>(j, "maxGuidedDepthInGuidingPhase");
    s.maxWalkingDepth = json_get_or_throw<unsigned>(j, "maxWalkingDepth");
    s.saveSppMetricsDuration = json_get_or_throw<int>(j, "saveSppMetricsDuration");
    s.saveSppMetricsUntil = json_get_or_throw<int>(j, "saveSppMetricsUntil");
    s.saveTimeMetricsDuration = json_get_or_throw<int>(j, "saveTimeMetricsDuration");
    s.epsilonShell = json_get_or_throw<float>(j, "epsilonShell");
    return s;
}

wost_net_config network_config_from_json(const json &n)
{
    const json enc = json_get_or_throw<json>(n, "encoding");
    const json net = json_get_or_throw<json>(n, "network");
    const json opt = json_get_or_throw<json>(n, "optimizer");
    if (json_get_or_throw<string>(enc, "otype") != "DenseGrid" || json_get_optional<string>(enc, "interpolation", "Linear") != "Linear")
        throw std::runtime_error("network.encoding: this build implements DenseGrid with Linear interpolation");
    if (json_get_optional<string>(net, "activation", "ReLU") != "ReLU" ||
        json_get_optional<string>(net, "output_activation", "None") != "None")
        throw std::runtime_error("network.network: this build implements ReLU hidden / None output activations");
    if (json_get_or_throw<string>(opt, "otype") != "Ema")
        throw std::runtime_error("network.optimizer: this build implements Ema{Adam}");
    const json adam = json_get_or_throw<json>(opt, "nested");
    if (json_get_or_throw<string>(adam, "otype") != "Adam") throw std::runtime_error("network.optimizer.nested: Adam expected");
    // tiny-cuda-nn's Adam has an AdaBound variant behind this switch; the shipped configurations turn it off
    if (json_get_optional<bool>(adam, "adabound", false))
        throw std::runtime_error("network.optimizer.nested.adabound: the AdaBound variant is not implemented");
    wost_net_config c{};
    c.n_levels = json_get_or_throw<int>(enc, "n_levels");
    c.n_features_per_level = json_get_or_throw<int>(enc, "n_features_per_level");
    c.base_resolution = json_get_or_throw<int>(enc, "base_resolution");
    c.per_level_scale = json_get_or_throw<float>(enc, "per_level_scale");
    c.n_neurons = json_get_or_throw<int>(net, "n_neurons");
    c.n_hidden_layers = json_get_or_throw<int>(net, "n_hidden_layers");
    c.n_output = 33;   // 8 lobes x (lambda, kappa, mu.x, mu.y) + selection logit (guided/parameters.h:16-24)
    c.learning_rate = json_get_or_throw<float>(adam, "learning_rate");
    c.beta1 = json_get_or_throw<float>(adam, "beta1");
    c.beta2 = json_get_or_throw<float>(adam, "beta2");
    c.epsilon = json_get_or_throw<float>(adam, "epsilon");
    c.l2_reg = json_get_or_throw<float>(adam, "l2_reg");
    c.ema_decay = json_get_or_throw<float>(opt, "decay");
    return c;
}

GuidedIntegrator<2>::GuidedIntegrator(Problem<2> &problem_, const IntegratorSettings &settings, const fs::path &basePath_,
                                      int device_)
    : IntegratorOutputs(settings.frameSize, basePath_), problem(problem_), integratorSettings(settings), device(device_)
{
}

GuidedIntegrator<2>::~GuidedIntegrator()
{
    if (handle) wost_guided_destroy(handle);
}

void GuidedIntegrator<2>::resetNetwork(const json &config)
{
    if (handle) {
        wost_guided_destroy(handle);
        handle = nullptr;
    }
    const IntegratorSettings &s = integratorSettings;
    const wost_net_config nc = network_config_from_json(config);
    const wost_scene_desc sd = problem.scene_desc(s.frameSize.x, s.frameSize.y);
    wost_guided_settings gs{};
    gs.width = s.frameSize.x; gs.height = s.frameSize.y; gs.spp = s.samplesPerPixel; gs.max_depth = (int32_t)s.maxWalkingDepth;
    gs.eps_shell = s.epsilonShell;
    gs.train_spp_count = (int32_t)s.trainSppCount;
    gs.uniform_fraction_training = s.uniformFractionInTrainingPhase;
    gs.uniform_fraction_guiding = s.uniformFractionInGuidingPhase;
    gs.max_guided_depth_training = (int32_t)s.maxGuidedDepthInTrainingPhase;
    gs.max_guided_depth_guiding = (int32_t)s.maxGuidedDepthInGuidingPhase;
    const AABB2f &b = problem.getAABB();
    gs.aabb_min[0] = b.min.x; gs.aabb_min[1] = b.min.y; gs.aabb_max[0] = b.max.x; gs.aabb_max[1] = b.max.y;
    // the reference's compile-time training constants (parameters.h:7-14, integrator.h:237-239)
    gs.max_train_depth = 3; gs.batch_size = 65536 * 8; gs.min_batch_size = 65536; gs.batches_per_spp = 5;
    gs.train_pixel_stride = 1; gs.train_pixel_offset = -1; gs.loss_scale = 128.0f;   // offset drawn like the reference when the stride is > 1
    check_wost(wost_guided_create(&sd, &gs, &nc, /* ELAINA_DEFAULT_RNG_SEED */ 42, device, &handle), "wost_guided_create");
}

wost_handle GuidedIntegrator<2>::scene_handle()
{
    if (!handle) throw std::runtime_error("GuidedIntegrator: resetNetwork() has not been called");
    wost_handle scene = nullptr;
    check_wost(wost_guided_scene(handle, &scene), "wost_guided_scene");
    return scene;
}

// frames/<sampleId>.exr|png and frames_time/<ms>.exr|png (reference integrator.cu:1049-1081)
static int save_frame(void *user, int reason, int32_t sample_id, double elapsed_ms, const float *field)
{
    const GuidedIntegrator<2> *self = static_cast<const GuidedIntegrator<2> *>(user);
    const Vector2i fs_ = self->get_integratorSettings().frameSize;
    const std::vector<float> rgb(field, field + (size_t)fs_.x * fs_.y * 3);
    try {
        const fs::path dir = self->get_basePath() / (reason == 0 ? "frames" : "frames_time");
        fs::create_directories(dir);
        const string name = reason == 0 ? std::to_string(sample_id) : std::to_string((long long)elapsed_ms);
        write_exr(dir / (name + ".exr"), fs_.x, fs_.y, rgb);
        write_png(dir / (name + ".png"), fs_.x, fs_.y, rgb);
    } catch (const std::exception &e) {
        ELAINA_LOG(Error, "saving an intermediate frame failed: %s", e.what());
        return 1;
    }
    return 0;
}

uint64_t GuidedIntegrator<2>::solve()
{
    if (!handle) throw std::runtime_error("GuidedIntegrator: resetNetwork() has not been called");
    const IntegratorSettings &s = integratorSettings;
    if (s.saveSppMetricsDuration > 0 || s.saveTimeMetricsDuration > 0)
        check_wost(wost_guided_set_frame_callback(handle, save_frame, this, s.saveSppMetricsDuration, s.saveSppMetricsUntil,
                                                  s.saveTimeMetricsDuration), "wost_guided_set_frame_callback");
    const auto start = std::chrono::high_resolution_clock::now();
    std::vector<float> &f = channels[(size_t)ExportImageChannel::SOLUTION];
    f.assign((size_t)frameSize_.x * frameSize_.y * 3, 0.0f);
    check_wost(wost_guided_solve(handle, f.data(), &last_stats), "wost_guided_solve");
    const auto end = std::chrono::high_resolution_clock::now();
    return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count();
}

void GuidedIntegrator<2>::renderDirichletSDF() { render_sdf(scene_handle(), WOST_MESH_DIRICHLET, ExportImageChannel::DIRICHLET_SDF); }

void GuidedIntegrator<2>::renderSilhouetteSDF() { render_sdf(scene_handle(), WOST_MESH_NEUMANN, ExportImageChannel::NEUMANN_SDF); }

void GuidedIntegrator<2>::renderSource()
{
    std::vector<float> &c = channels[(size_t)ExportImageChannel::SOURCE];
    c.assign((size_t)frameSize_.x * frameSize_.y * 3, 0.0f);
    check_wost(wost_render_source(scene_handle(), c.data()), "wost_render_source");
}

void GuidedIntegrator<2>::queryNetwork(const VectorType &p)
{
    if (!handle) throw std::runtime_error("GuidedIntegrator: resetNetwork() has not been called");
    float raw[33];
    const float xy[2] = {p.x, p.y};
    check_wost(wost_guided_query_network(handle, xy, 1, raw), "wost_guided_query_network");
    ELAINA_LOG(Info, "VMM @ (%f, %f): ", p.x, p.y);
    float total = 0.0f;
    float lambda[8];
    for (int k = 0; k < 8; ++k) total += lambda[k] = std::exp(std::fmin(std::fmax(raw[4 * k], -10.0f), 15.0f));
    for (int k = 0; k < 8; ++k) {
        const float nn = std::sqrt(raw[4 * k + 2] * raw[4 * k + 2] + raw[4 * k + 3] * raw[4 * k + 3]);
        ELAINA_LOG(Info, "  lobe %d: weight %.4f kappa %.4f mu (%.4f, %.4f)", k, lambda[k] / total,
                   std::exp(std::fmin(std::fmax(raw[4 * k + 1], -10.0f), 15.0f)), raw[4 * k + 2] / nn, raw[4 * k + 3] / nn);
    }
    ELAINA_LOG(Info, "  selection probability %.4f", 1.0f / (1.0f + std::exp(-raw[32])));
}

// ---- GuidedIntegrator<3> ---------------------------------------------------------------------------------------------
GuidedIntegrator<3>::GuidedIntegrator(Problem<3> &problem_, const IntegratorSettings &settings, const fs::path &basePath_, int device_)
    : IntegratorOutputs(settings.frameSize, basePath_), problem(problem_), integratorSettings(settings), device(device_)
{
}

GuidedIntegrator<3>::~GuidedIntegrator()
{
    if (handle) wost3_guided_destroy(handle);
}

void GuidedIntegrator<3>::resetNetwork(const json &config)
{
    if (handle) {
        wost3_guided_destroy(handle);
        handle = nullptr;
    }
    if (!problem.hasAABB()) throw std::runtime_error("scene.aabb (min / max with three entries) is required by the guided integrator");
    const IntegratorSettings &s = integratorSettings;
    wost_net_config nc = network_config_from_json(config);
    nc.n_output = 41;       // 8 lobes x (lambda, kappa, mean vector) + selection logit (guided/parameters.h:26-33)
    const wost3_scene_desc sd = problem.scene_desc(s.frameSize.x, s.frameSize.y);
    wost3_guided_settings gs{};
    gs.width = s.frameSize.x; gs.height = s.frameSize.y; gs.spp = s.samplesPerPixel; gs.max_depth = (int32_t)s.maxWalkingDepth;
    gs.eps_shell = s.epsilonShell;
    gs.train_spp_count = (int32_t)s.trainSppCount;
    gs.uniform_fraction_training = s.uniformFractionInTrainingPhase;
    gs.uniform_fraction_guiding = s.uniformFractionInGuidingPhase;
    gs.max_guided_depth_training = (int32_t)s.maxGuidedDepthInTrainingPhase;
    gs.max_guided_depth_guiding = (int32_t)s.maxGuidedDepthInGuidingPhase;
    const AABB3f &b = problem.getAABB();
    gs.aabb_min[0] = b.min.x; gs.aabb_min[1] = b.min.y; gs.aabb_min[2] = b.min.z;
    gs.aabb_max[0] = b.max.x; gs.aabb_max[1] = b.max.y; gs.aabb_max[2] = b.max.z;
    gs.max_train_depth = 3; gs.batch_size = 65536 * 8; gs.min_batch_size = 65536; gs.batches_per_spp = 5;
    gs.train_pixel_stride = 1; gs.train_pixel_offset = -1; gs.loss_scale = 128.0f;
    check_wost(wost3_guided_create(&sd, &gs, &nc, /* ELAINA_DEFAULT_RNG_SEED */ 42, device, &handle), "wost3_guided_create");
}

wost3_handle GuidedIntegrator<3>::scene_handle()
{
    if (!handle) throw std::runtime_error("GuidedIntegrator: resetNetwork() has not been called");
    wost3_handle scene = nullptr;
    check_wost(wost3_guided_scene(handle, &scene), "wost3_guided_scene");
    return scene;
}

uint64_t GuidedIntegrator<3>::solve()
{
    if (!handle) throw std::runtime_error("GuidedIntegrator: resetNetwork() has not been called");
    const auto start = std::chrono::high_resolution_clock::now();
    std::vector<float> &f = channels[(size_t)ExportImageChannel::SOLUTION];
    f.assign((size_t)frameSize_.x * frameSize_.y * 3, 0.0f);
    check_wost(wost3_guided_solve(handle, f.data(), &last_stats), "wost3_guided_solve");
    return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - start).count();
}

void GuidedIntegrator<3>::renderDirichletSDF()
{
    std::vector<float> d((size_t)frameSize_.x * frameSize_.y);
    check_wost(wost3_render_sdf(scene_handle(), WOST_MESH_DIRICHLET, d.data()), "wost3_render_sdf");
    set_gray_channel(ExportImageChannel::DIRICHLET_SDF, d);
}

void GuidedIntegrator<3>::renderSilhouetteSDF()
{
    std::vector<float> d((size_t)frameSize_.x * frameSize_.y);
    check_wost(wost3_render_sdf(scene_handle(), WOST_MESH_NEUMANN, d.data()), "wost3_render_sdf");
    set_gray_channel(ExportImageChannel::NEUMANN_SDF, d);
}

void GuidedIntegrator<3>::renderSource()
{
    std::vector<float> &c = channels[(size_t)ExportImageChannel::SOURCE];
    c.assign((size_t)frameSize_.x * frameSize_.y * 3, 0.0f);
    check_wost(wost3_render_source(scene_handle(), c.data()), "wost3_render_source");
}

void GuidedIntegrator<3>::queryNetwork(const VectorType &p)
{
    if (!handle) throw std::runtime_error("GuidedIntegrator: resetNetwork() has not been called");
    float raw[41];
    const float xyz[3] = {p.x, p.y, p.z};
    check_wost(wost3_guided_query_network(handle, xyz, 1, raw), "wost3_guided_query_network");
    ELAINA_LOG(Info, "VMM @ (%f, %f, %f): ", p.x, p.y, p.z);
    float total = 0.0f;
    float lambda[8];
    for (int k = 0; k < 8; ++k) total += lambda[k] = std::exp(std::fmin(std::fmax(raw[5 * k], -10.0f), 15.0f));
    for (int k = 0; k < 8; ++k) {
        const float nn = std::sqrt(raw[5 * k + 2] * raw[5 * k + 2] + raw[5 * k + 3] * raw[5 * k + 3] + raw[5 * k + 4] * raw[5 * k + 4]);
        ELAINA_LOG(Info, "  lobe %d: weight %.4f kappa %.4f mu (%.4f, %.4f, %.4f)", k, lambda[k] / total,
                   std::exp(std::fmin(std::fmax(raw[5 * k + 1], -10.0f), 15.0f)), raw[5 * k + 2] / nn, raw[5 * k + 3] / nn, raw[5 * k + 4] / nn);
    }
    ELAINA_LOG(Info, "  selection probability %.4f", 1.0f / (1.0f + std::exp(-raw[40])));
}

}  // namespace elaina
