// exec.cpp -- run_expr: parse the config, build the Problem, build the integrator, run the
// requested channels, export, write result.json (reference exec.cu:39-221).  Differences,
// on purpose: the output directory is created (the reference writes conf.json into a
// directory it never makes, exec.cu:69); unknown channels / export entries are skipped
// instead of dereferencing an empty optional (exec.cu:193-199); dimensionality 3 is built for the
// uniform integrator only (exec.cu:102-122: the guided one exits like an unknown type).
#include "exec.h"

#include <chrono>
#include <ctime>
#include <fstream>
#include <set>
#include <type_traits>

#include "core/problem.h"
#include "integrator/guided/integrator.h"
#include "integrator/uniform/integrator.h"

using namespace elaina;

static std::string get_current_time()
{
    const std::time_t t = std::chrono::system_clock::to_time_t(std::chrono::system_clock::now());
    char buffer[20];
    std::strftime(buffer, sizeof(buffer), "%Y-%m-%d %H:%M:%S", std::localtime(&t));
    return buffer;
}

// the channel + export loops of run_expr (reference exec.cu:145-215), for either integrator
template <class Integrator>
static void run_channels(Integrator &obj, const json &conf_json, const json &integrator_section, json &result_json)
{
    const json export_section = json_get_or_throw<json>(conf_json, "export");
    const json integrator_channels = json_get_or_throw<json>(integrator_section, "channels");
    std::set<ExportImageChannel> channels;
    for (const json &c : integrator_channels.items()) {
        ExportImageChannel ch;
        if (!parse_channel(c.get<string>(), &ch)) {
            ELAINA_LOG(Error, "Unrecognized integrator channel, skipping...");
            continue;
        }
        channels.insert(ch);
    }
    for (const ExportImageChannel ch : channels) {
        switch (ch) {
        case ExportImageChannel::SOLUTION: result_json["duration"] = json((uint64_t)obj.solve()); break;
        case ExportImageChannel::DIRICHLET_SDF: obj.renderDirichletSDF(); break;
        case ExportImageChannel::NEUMANN_SDF: obj.renderSilhouetteSDF(); break;
        case ExportImageChannel::SOURCE: obj.renderSource(); break;
        default: break;
        }
    }
    if (json_get_optional<bool>(conf_json, "print_network", false)) {
        try {
            // exec.cu:175-186: the 3-D run asks for the mixture at (0, -0.21, 0), the 2-D run at the origin
            if constexpr (std::is_same_v<typename Integrator::VectorType, Vector3f>) obj.queryNetwork(Vector3f{0.0f, -0.21f, 0.0f});
            else obj.queryNetwork(typename Integrator::VectorType{});
        } catch (const std::exception &e) {
            ELAINA_LOG(Warning, "print_network: %s", e.what());
        }
    }
    for (const json &m : export_section.items()) {
        const string type = json_get_or_throw<string>(m, "type");
        const string channel_string = json_get_or_throw<string>(m, "channel");
        const string file_name = json_get_or_throw<string>(m, "file_name");
        ExportImageChannel ch;
        if (!parse_channel(channel_string, &ch)) {
            ELAINA_LOG(Error, "Unrecognized export channel, skipping...");
            continue;
        }
        if (type == "image") {
            obj.exportImage(ch, file_name);
        } else if (type == "energy") {
            ToneMapping tone;
            if (parse_tone(json_get_or_throw<string>(m, "tone"), &tone)) obj.exportEnergy(ch, tone, file_name);
            else ELAINA_LOG(Error, "Unrecognized tone mapping method, skipping...");
        }
    }
}

void run_expr(fs::path conf_path)
{
    if (!fs::exists(conf_path)) {
        ELAINA_LOG(Error, "Configuration file does not exist: %s", conf_path.c_str());
        return;
    }
    json conf_json, result_json = json::object();
    try {
        conf_json = load_json_file(conf_path.string());
    } catch (const std::exception &e) {
        ELAINA_LOG(Error, "Failed to parse JSON: %s", e.what());
        return;
    }
    const int dimensionality = json_get_or_throw<int>(conf_json, "dimensionality");
    const fs::path basePath = json_get_or_throw<string>(conf_json, "base_path");
    const std::string expName = json_get_or_throw<string>(conf_json, "exp_name");
    const fs::path outDir = basePath / expName;
    fs::create_directories(outDir);
    {
        std::ofstream confFileCopy(outDir / "conf.json");
        confFileCopy << conf_json.dump(4) << std::endl;
    }
    ELAINA_LOG(Success, "Configuration file copied to %s", (outDir / "conf.json").c_str());

    const json scene_section = json_get_or_throw<json>(conf_json, "scene");
    const json integrator_section = json_get_or_throw<json>(conf_json, "integrator");
    const string integrator_type = json_get_or_throw<string>(integrator_section, "type");
    const json integrator_setting = json_get_or_throw<json>(integrator_section, "setting");
    if (dimensionality == 3) {
        // exec.cu:102-122: the same two integrator types on Problem<3>
        Problem<3> scene3;
        scene3.loadConfig(scene_section, conf_path.parent_path());
        if (integrator_type == "uniform") {
            UniformIntegrator<3> obj(scene3, UniformIntegratorSettings::from_json(integrator_setting), outDir);
            run_channels(obj, conf_json, integrator_section, result_json);
            const wost_stats &s = obj.get_last_stats();
            if (s.walk_steps) {
                result_json["walk_steps"] = json((uint64_t)s.walk_steps);
                result_json["walk_steps_per_second"] = json(s.solve_ms > 0 ? (double)s.walk_steps / (s.solve_ms * 1e-3) : 0.0);
            }
        } else if (integrator_type == "guided") {
            const json network_section = json_get_or_throw<json>(conf_json, "network");
            GuidedIntegrator<3> obj(scene3, GuidedIntegratorSettings::from_json(integrator_setting), outDir);
            obj.resetNetwork(network_section);
            run_channels(obj, conf_json, integrator_section, result_json);
            const wost_guided_stats &s = obj.get_last_stats();
            if (s.walk_steps) {
                result_json["walk_steps"] = json((uint64_t)s.walk_steps);
                result_json["guided_steps"] = json((uint64_t)s.guided_steps);
                result_json["optimizer_steps"] = json((uint64_t)s.optimizer_steps);
                result_json["walk_steps_per_second"] = json(s.solve_ms > 0 ? (double)s.walk_steps / (s.solve_ms * 1e-3) : 0.0);
            }
        } else {
            ELAINA_LOG(Error, "Unrecognized integrator type.");
            exit(1);
        }
        result_json["timestamp"] = json(get_current_time());
        std::ofstream resultFile(outDir / "result.json");
        resultFile << result_json.dump(4) << std::endl;
        ELAINA_LOG(Success, "Result file written to %s", (outDir / "result.json").c_str());
        return;
    }
    if (dimensionality != 2) {
        ELAINA_LOG(Error, "Unsupported dimensionality.");
        exit(1);
    }
    Problem<2> scene;
    scene.loadConfig(scene_section, conf_path.parent_path());
    if (integrator_type == "uniform") {
        UniformIntegrator<2> obj(scene, UniformIntegratorSettings::from_json(integrator_setting), outDir);
        run_channels(obj, conf_json, integrator_section, result_json);
        const wost_stats &s = obj.get_last_stats();
        if (s.walk_steps) {
            result_json["walk_steps"] = json((uint64_t)s.walk_steps);
            result_json["walk_steps_per_second"] = json(s.solve_ms > 0 ? (double)s.walk_steps / (s.solve_ms * 1e-3) : 0.0);
        }
    } else if (integrator_type == "guided") {
        const json network_section = json_get_or_throw<json>(conf_json, "network");
        GuidedIntegrator<2> obj(scene, GuidedIntegratorSettings::from_json(integrator_setting), outDir);
        obj.resetNetwork(network_section);
        run_channels(obj, conf_json, integrator_section, result_json);
        const wost_guided_stats &s = obj.get_last_stats();
        if (s.walk_steps) {
            result_json["walk_steps"] = json((uint64_t)s.walk_steps);
            result_json["guided_steps"] = json((uint64_t)s.guided_steps);
            result_json["optimizer_steps"] = json((uint64_t)s.optimizer_steps);
            result_json["walk_steps_per_second"] = json(s.solve_ms > 0 ? (double)s.walk_steps / (s.solve_ms * 1e-3) : 0.0);
        }
    } else {
        ELAINA_LOG(Error, "Unrecognized integrator type.");
        exit(1);
    }
    result_json["timestamp"] = json(get_current_time());
    std::ofstream resultFile(outDir / "result.json");
    resultFile << result_json.dump(4) << std::endl;
    ELAINA_LOG(Success, "Result file written to %s", (outDir / "result.json").c_str());
}
