// main.cpp -- `elaina-exec <conf.json>` (reference main.cpp:9-19).  `--selftest` runs the
// host-only checks (JSON, OBJ, settings binding) that need no GPU.
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>

#include "core/problem.h"
#include "exec.h"
#include "integrator/guided/integrator.h"
#include "integrator/uniform/integrator.h"

using namespace elaina;

static int selftest()
{
    int failures = 0;
    auto expect = [&](bool ok, const char *what) {
        if (!ok) { std::cerr << "selftest FAILED: " << what << std::endl; ++failures; }
    };
    const json j = json::parse(R"({"a": {"b": [1, 2.5, -3e2], "s": "x\nyé", "t": true, "n": null}, "k": 7})");
    expect(json_get_or_throw<int>(j, "k") == 7, "int");
    expect(json_get_or_throw<std::vector<float>>(j, "a/b")[2] == -300.0f, "array");
    expect(json_get_or_throw<string>(j, "a/s") == "x\ny\xc3\xa9", "string escapes");
    expect(json_get_or_throw<bool>(j, "a/t"), "bool");
    expect(!json_get_optional<int>(j, "a/n").has_value(), "null is absent");
    expect(json_get_optional<float>(j, "missing/path", 3.0f) == 3.0f, "optional default");
    bool threw = false;
    try { json_get_or_throw<int>(j, "nope"); } catch (const std::runtime_error &) { threw = true; }
    expect(threw, "missing key throws");
    threw = false;
    try { json::parse("{\"a\": }"); } catch (const std::runtime_error &) { threw = true; }
    expect(threw, "syntax error throws");
    expect(json::parse(j.dump(2)).dump() == j.dump(), "dump/parse round trip");
    // settings binding: every key required
    const json st = json::parse(R"({"frameSize":[64,32],"debugPixel":0,"samplesPerPixel":4,"maxWalkingDepth":16,
        "saveSppMetricsDuration":-1,"saveSppMetricsUntil":-1,"saveTimeMetricsDuration":-1,"epsilonShell":1})");
    const UniformIntegratorSettings s = UniformIntegratorSettings::from_json(st);
    expect(s.frameSize.x == 64 && s.frameSize.y == 32 && s.samplesPerPixel == 4 && s.epsilonShell == 1.0f, "settings");
    threw = false;
    try { UniformIntegratorSettings::from_json(json::parse("{\"frameSize\":[1,1]}")); } catch (const std::runtime_error &) { threw = true; }
    expect(threw, "missing setting throws");
    // guided settings + network section binding
    const json gst = json::parse(R"({"frameSize":[64,32],"debugPixel":0,"samplesPerPixel":4,"maxWalkingDepth":16,
        "saveSppMetricsDuration":-1,"saveSppMetricsUntil":-1,"saveTimeMetricsDuration":-1,"epsilonShell":1,
        "trainSppCount":2,"uniformFractionInTrainingPhase":0.5,"uniformFractionInGuidingPhase":0.25,
        "maxGuidedDepthInTrainingPhase":10,"maxGuidedDepthInGuidingPhase":7})");
    const GuidedIntegratorSettings g2 = GuidedIntegratorSettings::from_json(gst);
    expect(g2.trainSppCount == 2 && g2.uniformFractionInGuidingPhase == 0.25f && g2.maxGuidedDepthInGuidingPhase == 7, "guided settings");
    threw = false;
    try { GuidedIntegratorSettings::from_json(st); } catch (const std::runtime_error &) { threw = true; }
    expect(threw, "missing guided setting throws");
    const json net = json::parse(R"({"encoding":{"otype":"DenseGrid","interpolation":"Linear","n_levels":8,
        "n_features_per_level":4,"base_resolution":8,"per_level_scale":1.405},"loss":{"otype":"L2"},
        "network":{"otype":"FullyFusedMLP","activation":"ReLU","output_activation":"None","n_neurons":64,"n_hidden_layers":3},
        "optimizer":{"otype":"Ema","decay":0.95,"nested":{"otype":"Adam","beta1":0.9,"beta2":0.99,"epsilon":1e-15,
        "l2_reg":1e-6,"learning_rate":0.008}}})");
    const wost_net_config nc = network_config_from_json(net);
    expect(nc.n_levels == 8 && nc.n_neurons == 64 && nc.n_hidden_layers == 3 && nc.n_output == 33 && nc.ema_decay == 0.95f &&
               nc.learning_rate == 0.008f, "network section");
    threw = false;
    try { network_config_from_json(json::parse(R"({"encoding":{"otype":"HashGrid"},"network":{},"optimizer":{}})")); }
    catch (const std::runtime_error &) { threw = true; }
    expect(threw, "unsupported encoding throws");
    // source grid binding: nanovdb paths are refused with a pointer to the dense format
    {
        Problem<2> pr(false);
        threw = false;
        try {
            pr.loadConfig(json::parse(R"({"aabb":{"min":[0,0],"max":[1,1]},"evaluation_grid":{"mData":{"scale":1,"pos":[0,0],"up":[0,1]}},
                "mesh":{},"source_path":"x.nvdb"})"));
        } catch (const std::runtime_error &e) { threw = string(e.what()).find("source_grid") != string::npos; }
        expect(threw, "source_path is refused with a hint");
        pr.set_source(2, 1, {1, 2, 3, 4, 5, 6}, {2.0f, 2.0f}, {0.5f, 0.0f});
        expect(pr.isSourceEnabled() && pr.scene_desc(4, 4).source.nx == 2 && pr.scene_desc(4, 4).source.rgb[5] == 6.0f, "set_source");
    }
    // OBJ polylines
    const fs::path tmp = fs::temp_directory_path() / "elaina_selftest.obj";
    { std::ofstream f(tmp); f << "# c\no P\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nl 1 2 3\nl -1 1\n"; }
    SceneLoader2 ld(tmp.string());
    expect(ld.vertices.size() == 8 && ld.indices.size() == 6, "obj counts");
    expect(ld.indices[2] == 1 && ld.indices[3] == 2 && ld.indices[4] == 3 && ld.indices[5] == 0, "obj polyline + relative index");
    fs::remove(tmp);
    EvaluationGrid<2> g(json::parse(R"({"mData":{"scale":250,"pos":[250,250],"up":[-1,0]}})"));
    const Vector2f p = g.getEvaluationPoint({0, 0}, {1024, 1024});
    expect(std::fabs(p.x - 500.0f) < 1e-3f && std::fabs(p.y - 0.0f) < 1e-3f, "evaluation grid");
    std::cout << (failures ? "selftest failed" : "selftest ok") << std::endl;
    return failures ? 1 : 0;
}

// writes a small known image in both export formats and every colormap at a few abscissae
static int imagetest(const fs::path &dir)
{
    const int w = 5, h = 3;
    std::vector<float> rgb((size_t)w * h * 3);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float *p = &rgb[3 * ((size_t)y * w + x)];
            p[0] = (float)x / 4.0f;          // 0 .. 1
            p[1] = (float)y - 0.5f;          // negative, in range, above 1
            p[2] = 0.1f * (float)(x + y * w);
        }
    fs::create_directories(dir);
    write_png(dir / "grad.png", w, h, rgb);
    write_exr(dir / "grad.exr", w, h, rgb);
    for (ToneMapping t : {ToneMapping::MATLAB_JET, ToneMapping::MATLAB_PARULA, ToneMapping::IDL_RDBU, ToneMapping::NONE_NORMALIZED})
        for (float x : {0.0f, 0.25f, 0.5f, 0.75f, 1.0f}) {
            float c[3];
            tone_map(t, x, c);
            std::cout << "tone " << (int)t << " " << x << " " << c[0] << " " << c[1] << " " << c[2] << std::endl;
        }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::cerr << "Usage: " << argv[0] << " <conf_path> | --selftest" << std::endl;
        return 1;
    }
    if (std::strcmp(argv[1], "--selftest") == 0) return selftest();
    if (std::strcmp(argv[1], "--imagetest") == 0 && argc > 2) return imagetest(argv[2]);
    if (std::strcmp(argv[1], "--readpng") == 0 && argc > 3) {   // decode a PNG to raw RGBA8 (tests of the mask reader)
        try {
            int w = 0, h = 0;
            std::vector<uint8_t> rgba;
            read_png(argv[2], &w, &h, &rgba);
            std::ofstream f(argv[3], std::ios::binary);
            f.write(reinterpret_cast<const char *>(rgba.data()), (std::streamsize)rgba.size());
            std::cout << w << " " << h << std::endl;
            return 0;
        } catch (const std::exception &e) {
            std::cerr << e.what() << std::endl;
            return 1;
        }
    }
    if (std::strcmp(argv[1], "--readmask") == 0 && argc > 3) {   // a mask image of any readable format to raw bytes (tests)
        try {
            int w = 0, h = 0;
            std::vector<uint8_t> mask;
            read_mask_image(argv[2], &w, &h, &mask);
            std::ofstream f(argv[3], std::ios::binary);
            f.write(reinterpret_cast<const char *>(mask.data()), (std::streamsize)mask.size());
            std::cout << w << " " << h << std::endl;
            return 0;
        } catch (const std::exception &e) {
            std::cerr << e.what() << std::endl;
            return 1;
        }
    }
    try {
        run_expr(fs::path(argv[1]));
    } catch (const std::exception &e) {
        ELAINA_LOG(Error, "%s", e.what());
        return 1;
    }
    return 0;
}
