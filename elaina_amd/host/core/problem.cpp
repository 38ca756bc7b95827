#include "problem.h"

#include "util/image_io.h"

#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace elaina {

// ---- common.h helpers ---------------------------------------------------------------------
bool parse_channel(const string &n, ExportImageChannel *out)
{
    if (n == "DIRICHLET_SDF") *out = ExportImageChannel::DIRICHLET_SDF;
    else if (n == "NEUMANN_SDF") *out = ExportImageChannel::NEUMANN_SDF;
    else if (n == "SOURCE") *out = ExportImageChannel::SOURCE;
    else if (n == "SOLUTION") *out = ExportImageChannel::SOLUTION;
    else return false;
    return true;
}

bool parse_tone(const string &n, ToneMapping *out)
{
    if (n == "NONE") *out = ToneMapping::NONE;
    else if (n == "NONE_NORMALIZED") *out = ToneMapping::NONE_NORMALIZED;
    else if (n == "MATLAB_JET") *out = ToneMapping::MATLAB_JET;
    else if (n == "MATLAB_PARULA") *out = ToneMapping::MATLAB_PARULA;
    else if (n == "IDL_RDBU") *out = ToneMapping::IDL_RDBU;
    else return false;
    return true;
}

const char *channel_name(ExportImageChannel c)
{
    switch (c) {
    case ExportImageChannel::DIRICHLET_SDF: return "DIRICHLET_SDF";
    case ExportImageChannel::NEUMANN_SDF: return "NEUMANN_SDF";
    case ExportImageChannel::SOURCE: return "SOURCE";
    case ExportImageChannel::SOLUTION: return "SOLUTION";
    default: return "?";
    }
}

void log_message(LogLevel level, const char *fmt, ...)
{
    static const char *tag[] = {"[debug] ", "[info] ", "[ok] ", "[warning] ", "[error] "};
    std::fputs(tag[(int)level], stderr);
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fputc('\n', stderr);
}

// ---- OBJ polylines -------------------------------------------------------------------------
SceneLoader2::SceneLoader2(const string &path)
{
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("Failed to open model file: " + path);
    string line;
    while (std::getline(f, line)) {
        const char *p = line.c_str();
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char *e = nullptr;
            const double x = std::strtod(p + 1, &e);
            const double y = std::strtod(e, &e);
            vertices.push_back((float)x);
            vertices.push_back((float)y);
        } else if (p[0] == 'l' && (p[1] == ' ' || p[1] == '\t')) {
            std::vector<long> idx;
            const char *q = p + 1;
            while (true) {
                char *e = nullptr;
                const long v = std::strtol(q, &e, 10);
                if (e == q) break;
                idx.push_back(v);
                q = e;
                while (*q == '/' || (*q >= '0' && *q <= '9')) ++q;  // skip "/vt" suffixes
            }
            const long nv = (long)(vertices.size() / 2);
            for (size_t i = 0; i + 1 < idx.size(); ++i) {
                long a = idx[i] > 0 ? idx[i] - 1 : nv + idx[i];
                long b = idx[i + 1] > 0 ? idx[i + 1] - 1 : nv + idx[i + 1];
                indices.push_back((int32_t)a);
                indices.push_back((int32_t)b);
            }
        }
    }
    const long nv = (long)(vertices.size() / 2);
    for (int32_t i : indices)
        if (i < 0 || i >= nv) throw std::runtime_error("OBJ segment references a missing vertex: " + path);
}

// ---- vertex colours ------------------------------------------------------------------------
std::vector<float> parseVertexColorFile(const string &path)
{
    const json conf = load_json_file(path);
    const json cc = json_get_or_throw<json>(conf, "ColorConfigurations");
    if (!cc.is_array()) throw std::runtime_error("The ColorConfigurations item is not an array.");
    std::vector<float> out(cc.size() * 6);
    for (size_t i = 0; i < cc.size(); ++i) {
        const json &c = cc[i];
        if (json_get_or_throw<int>(c, "vertexID") != (int)i + 1)
            throw std::runtime_error("The configurations should be sorted.");
        out[6 * i + 0] = json_get_or_throw<float>(c, "leftColor/R");
        out[6 * i + 1] = json_get_or_throw<float>(c, "leftColor/G");
        out[6 * i + 2] = json_get_or_throw<float>(c, "leftColor/B");
        out[6 * i + 3] = json_get_or_throw<float>(c, "rightColor/R");
        out[6 * i + 4] = json_get_or_throw<float>(c, "rightColor/G");
        out[6 * i + 5] = json_get_or_throw<float>(c, "rightColor/B");
    }
    return out;
}

// ---- Problem<2> ----------------------------------------------------------------------------
static string resolve(const string &p, const fs::path &search_dir)
{
    if (fs::exists(p) || search_dir.empty()) return p;
    const fs::path alt = search_dir / p;
    return fs::exists(alt) ? alt.string() : p;
}

void Problem<2>::loadConfig(const json &config, const fs::path &search_dir)
{
    // reference core/problem.cu:152-181
    const json probeConfig = json_get_or_throw<json>(config, "evaluation_grid");
    const auto amin = json_get_or_throw<std::vector<float>>(config, "aabb/min");
    const auto amax = json_get_or_throw<std::vector<float>>(config, "aabb/max");
    if (amin.size() != 2 || amax.size() != 2) throw std::runtime_error("aabb/min and aabb/max must have 2 entries");
    mAABB = AABB2f{{amin[0], amin[1]}, {amax[0], amax[1]}};
    mpProbe = std::make_shared<SceneProbe>(probeConfig);

    const json meshConfig = json_get_or_throw<json>(config, "mesh");
    const auto dirichlet_path = json_get_optional<string>(meshConfig, "dirichlet_path");
    const auto neumann_path = json_get_optional<string>(meshConfig, "neumann_path");
    if (dirichlet_path) {
        scene_dirichlet_loader = std::make_unique<SceneLoader2>(resolve(*dirichlet_path, search_dir));
        enable_dirichlet = true;
        scene_stat.dirichlet_vertices_size = scene_dirichlet_loader->vertices.size() / 2;
        scene_stat.dirichlet_primitives_size = scene_dirichlet_loader->indices.size() / 2;
    }
    if (neumann_path) {
        scene_neumann_loader = std::make_unique<SceneLoader2>(resolve(*neumann_path, search_dir));
        enable_neumann = true;
        scene_stat.neumann_vertices_size = scene_neumann_loader->vertices.size() / 2;
        scene_stat.neumann_primitives_size = scene_neumann_loader->indices.size() / 2;
    }
    // reference core/problem.cu:99-133: absent file => all-zero colours
    const auto cd = json_get_optional<string>(meshConfig, "vertex_color_dirichlet_path");
    const auto cn = json_get_optional<string>(meshConfig, "vertex_color_neumann_path");
    if (enable_dirichlet) {
        if (cd) vertex_color_dirichlet = parseVertexColorFile(resolve(*cd, search_dir));
        else vertex_color_dirichlet.assign(scene_stat.dirichlet_vertices_size * 6, 0.0f);
        if (vertex_color_dirichlet.size() != scene_stat.dirichlet_vertices_size * 6)
            throw std::runtime_error("Dirichlet colour file does not have one entry per vertex");
    }
    if (enable_neumann) {
        if (cn) vertex_color_neumann = parseVertexColorFile(resolve(*cn, search_dir));
        else vertex_color_neumann.assign(scene_stat.neumann_vertices_size * 6, 0.0f);
        if (vertex_color_neumann.size() != scene_stat.neumann_vertices_size * 6)
            throw std::runtime_error("Neumann colour file does not have one entry per vertex");
    }
    // The reference reads a nanovdb Vec3f grid ("source_path", core/problem.cu:136-149); nanovdb is
    // not available here, so the source term comes as a dense grid:
    //   "source_grid": {"path": raw little-endian float32 [ny][nx][3], "nx", "ny",
    //                   "index_scale": [sx, sy], "index_offset": [ox, oy]}   index = world * scale + offset
    if (json_get_optional<string>(config, "source_path"))
        throw std::runtime_error("source_path: nanovdb grids cannot be read by this build; export the grid as \"source_grid\" "
                                 "(dense float32 RGB, see core/problem.cpp)");
    if (const auto sg = json_get_optional<json>(config, "source_grid")) {
        const int nx = json_get_or_throw<int>(*sg, "nx"), ny = json_get_or_throw<int>(*sg, "ny");
        const auto sc = json_get_or_throw<std::vector<float>>(*sg, "index_scale");
        const auto of = json_get_or_throw<std::vector<float>>(*sg, "index_offset");
        if (nx <= 0 || ny <= 0 || sc.size() != 2 || of.size() != 2) throw std::runtime_error("source_grid: bad shape");
        const string path = resolve(json_get_or_throw<string>(*sg, "path"), search_dir);
        std::ifstream f(path, std::ios::binary);
        if (!f.is_open()) throw std::runtime_error("cannot open source grid " + path);
        std::vector<float> rgb((size_t)nx * ny * 3);
        f.read(reinterpret_cast<char *>(rgb.data()), (std::streamsize)(rgb.size() * sizeof(float)));
        if ((size_t)f.gcount() != rgb.size() * sizeof(float)) throw std::runtime_error("source grid file is too short: " + path);
        set_source(nx, ny, std::move(rgb), {sc[0], sc[1]}, {of[0], of[1]});
    }
    // reference core/problem.cu:216-242: the image is loaded flipped vertically; a pixel is on when any of its R, G, B
    // values is non-zero.  PNG, OpenEXR and PFM are read (util/image_io.cpp); the reference goes through stb_image / tinyexr.
    if (const auto mp = json_get_optional<string>(config, "mask_path")) {
        int w = 0, h = 0;
        read_mask_image(resolve(*mp, search_dir), &w, &h, &mask);
    }
    source_intensity = json_get_optional<float>(config, "source_intensity", 1.0f);
    dirichlet_intensity = json_get_optional<float>(config, "dirichlet_intensity", 1.0f);
    neumann_intensity = json_get_optional<float>(config, "neumann_intensity", 1.0f);
    if (verbose) {
        ELAINA_LOG(Success, "Problem: loadConfig is completed.");
        if (enable_dirichlet)
            ELAINA_LOG(Info, "Dirichlet: %zu vertices, %zu primitives, intensity %f", scene_stat.dirichlet_vertices_size,
                       scene_stat.dirichlet_primitives_size, dirichlet_intensity);
        if (enable_neumann)
            ELAINA_LOG(Info, "Neumann: %zu vertices, %zu primitives, intensity %f", scene_stat.neumann_vertices_size,
                       scene_stat.neumann_primitives_size, neumann_intensity);
    }
}

void Problem<2>::set_source(int nx, int ny, std::vector<float> rgb, Vector2f index_scale, Vector2f index_offset)
{
    if (nx <= 0 || ny <= 0 || rgb.size() != (size_t)nx * ny * 3) throw std::runtime_error("set_source: size mismatch");
    source_nx = nx; source_ny = ny;
    source_rgb = std::move(rgb);
    source_index_scale = index_scale;
    source_index_offset = index_offset;
    enable_source = true;
}

// ---- OBJ triangles + Problem<3> ------------------------------------------------------------------
SceneLoader3::SceneLoader3(const string &path)
{
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("Failed to open model file: " + path);
    string line;
    while (std::getline(f, line)) {
        const char *p = line.c_str();
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char *e = nullptr;
            const double x = std::strtod(p + 1, &e), y = std::strtod(e, &e), z = std::strtod(e, &e);
            vertices.push_back((float)x); vertices.push_back((float)y); vertices.push_back((float)z);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            std::vector<long> idx;
            const char *q = p + 1;
            while (true) {
                char *e = nullptr;
                const long v = std::strtol(q, &e, 10);
                if (e == q) break;
                idx.push_back(v);
                q = e;
                while (*q == '/' || (*q >= '0' && *q <= '9') || *q == '-') ++q;  // skip "/vt/vn" suffixes
            }
            const long nv = (long)(vertices.size() / 3);
            auto fix = [nv](long i) { return (int32_t)(i > 0 ? i - 1 : nv + i); };
            for (size_t i = 1; i + 1 < idx.size(); ++i) {
                indices.push_back(fix(idx[0])); indices.push_back(fix(idx[i])); indices.push_back(fix(idx[i + 1]));
            }
        }
    }
    const long nv = (long)(vertices.size() / 3);
    for (int32_t i : indices)
        if (i < 0 || i >= nv) throw std::runtime_error("OBJ face references a missing vertex: " + path);
}

void Problem<3>::loadConfig(const json &config, const fs::path &search_dir)
{
    // reference core/problem.cu:152-181 with DIM == 3 (the aabb only serves the guided integrator: optional here, required there)
    mpProbe = std::make_shared<SceneProbe>(json_get_or_throw<json>(config, "evaluation_grid"));
    {
        const auto amin = json_get_optional<std::vector<float>>(config, "aabb/min"), amax = json_get_optional<std::vector<float>>(config, "aabb/max");
        if (amin && amax) {
            if (amin->size() != 3 || amax->size() != 3) throw std::runtime_error("aabb/min and aabb/max must have 3 entries");
            mAABB = AABB3f{{(*amin)[0], (*amin)[1], (*amin)[2]}, {(*amax)[0], (*amax)[1], (*amax)[2]}};
            has_aabb = true;
        }
    }
    const json meshConfig = json_get_or_throw<json>(config, "mesh");
    if (const auto dp = json_get_optional<string>(meshConfig, "dirichlet_path")) {
        scene_dirichlet_loader = std::make_unique<SceneLoader3>(resolve(*dp, search_dir));
        enable_dirichlet = true;
        scene_stat.dirichlet_vertices_size = scene_dirichlet_loader->vertices.size() / 3;
        scene_stat.dirichlet_primitives_size = scene_dirichlet_loader->indices.size() / 3;
    }
    if (const auto np_ = json_get_optional<string>(meshConfig, "neumann_path")) {
        scene_neumann_loader = std::make_unique<SceneLoader3>(resolve(*np_, search_dir));
        enable_neumann = true;
        scene_stat.neumann_vertices_size = scene_neumann_loader->vertices.size() / 3;
        scene_stat.neumann_primitives_size = scene_neumann_loader->indices.size() / 3;
    }
    const auto cd = json_get_optional<string>(meshConfig, "vertex_color_dirichlet_path");
    const auto cn = json_get_optional<string>(meshConfig, "vertex_color_neumann_path");
    if (enable_dirichlet) {
        if (cd) vertex_color_dirichlet = parseVertexColorFile(resolve(*cd, search_dir));
        else vertex_color_dirichlet.assign(scene_stat.dirichlet_vertices_size * 6, 0.0f);
        if (vertex_color_dirichlet.size() != scene_stat.dirichlet_vertices_size * 6)
            throw std::runtime_error("Dirichlet colour file does not have one entry per vertex");
    }
    if (enable_neumann) {
        if (cn) vertex_color_neumann = parseVertexColorFile(resolve(*cn, search_dir));
        else vertex_color_neumann.assign(scene_stat.neumann_vertices_size * 6, 0.0f);
        if (vertex_color_neumann.size() != scene_stat.neumann_vertices_size * 6)
            throw std::runtime_error("Neumann colour file does not have one entry per vertex");
    }
    // as in 2-D: "source_grid": {"path": raw little-endian float32 [nz][ny][nx][3], "nx", "ny", "nz",
    //                            "index_scale": [sx, sy, sz], "index_offset": [ox, oy, oz]}
    if (json_get_optional<string>(config, "source_path"))
        throw std::runtime_error("source_path: nanovdb grids cannot be read by this build; export the grid as \"source_grid\" "
                                 "(dense float32 RGB, see core/problem.cpp)");
    if (const auto sg = json_get_optional<json>(config, "source_grid")) {
        const int nx = json_get_or_throw<int>(*sg, "nx"), ny = json_get_or_throw<int>(*sg, "ny"), nz = json_get_or_throw<int>(*sg, "nz");
        const auto sc = json_get_or_throw<std::vector<float>>(*sg, "index_scale");
        const auto of = json_get_or_throw<std::vector<float>>(*sg, "index_offset");
        if (nx <= 0 || ny <= 0 || nz <= 0 || sc.size() != 3 || of.size() != 3) throw std::runtime_error("source_grid: bad shape");
        const string path = resolve(json_get_or_throw<string>(*sg, "path"), search_dir);
        std::ifstream f(path, std::ios::binary);
        if (!f.is_open()) throw std::runtime_error("cannot open source grid " + path);
        std::vector<float> rgb((size_t)nx * ny * nz * 3);
        f.read(reinterpret_cast<char *>(rgb.data()), (std::streamsize)(rgb.size() * sizeof(float)));
        if ((size_t)f.gcount() != rgb.size() * sizeof(float)) throw std::runtime_error("source grid file is too short: " + path);
        set_source(nx, ny, nz, std::move(rgb), {sc[0], sc[1], sc[2]}, {of[0], of[1], of[2]});
    }
    source_intensity = json_get_optional<float>(config, "source_intensity", 1.0f);
    dirichlet_intensity = json_get_optional<float>(config, "dirichlet_intensity", 1.0f);
    neumann_intensity = json_get_optional<float>(config, "neumann_intensity", 1.0f);
    if (verbose) {
        ELAINA_LOG(Success, "Problem<3>: loadConfig is completed.");
        if (enable_dirichlet)
            ELAINA_LOG(Info, "Dirichlet: %zu vertices, %zu triangles", scene_stat.dirichlet_vertices_size, scene_stat.dirichlet_primitives_size);
        if (enable_neumann)
            ELAINA_LOG(Info, "Neumann: %zu vertices, %zu triangles", scene_stat.neumann_vertices_size, scene_stat.neumann_primitives_size);
    }
}

void Problem<3>::set_source(int nx, int ny, int nz, std::vector<float> rgb, Vector3f index_scale, Vector3f index_offset)
{
    if (nx <= 0 || ny <= 0 || nz <= 0 || rgb.size() != (size_t)nx * ny * nz * 3) throw std::runtime_error("set_source: size mismatch");
    source_nx = nx; source_ny = ny; source_nz = nz;
    source_rgb = std::move(rgb);
    source_index_scale = index_scale;
    source_index_offset = index_offset;
    enable_source = true;
}

wost3_scene_desc Problem<3>::scene_desc(int width, int height) const
{
    wost3_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    if (enable_dirichlet) {
        d.dirichlet.n_verts = (int32_t)scene_stat.dirichlet_vertices_size;
        d.dirichlet.n_tris = (int32_t)scene_stat.dirichlet_primitives_size;
        d.dirichlet.verts = scene_dirichlet_loader->vertices.data();
        d.dirichlet.tris = scene_dirichlet_loader->indices.data();
        d.dirichlet.colors = vertex_color_dirichlet.data();
    }
    if (enable_neumann) {
        d.neumann.n_verts = (int32_t)scene_stat.neumann_vertices_size;
        d.neumann.n_tris = (int32_t)scene_stat.neumann_primitives_size;
        d.neumann.verts = scene_neumann_loader->vertices.data();
        d.neumann.tris = scene_neumann_loader->indices.data();
        d.neumann.colors = vertex_color_neumann.data();
    }
    d.dirichlet_intensity = dirichlet_intensity;
    d.neumann_intensity = neumann_intensity;
    const SceneProbe::ProbeData &p = mpProbe->mData;
    d.probe_scale = p.scale;
    d.probe_pos[0] = p.pos.x; d.probe_pos[1] = p.pos.y; d.probe_pos[2] = p.pos.z;
    d.probe_up[0] = p.up.x; d.probe_up[1] = p.up.y; d.probe_up[2] = p.up.z;
    d.probe_right[0] = p.right.x; d.probe_right[1] = p.right.y; d.probe_right[2] = p.right.z;
    if (!mask.empty()) {
        if (mask.size() != (size_t)width * height) throw std::runtime_error("mask size does not match the frame");
        d.mask = mask.data();
    }
    if (enable_source) {
        d.source.nx = source_nx; d.source.ny = source_ny; d.source.nz = source_nz;
        d.source.rgb = source_rgb.data();
        d.source.index_scale[0] = source_index_scale.x; d.source.index_scale[1] = source_index_scale.y; d.source.index_scale[2] = source_index_scale.z;
        d.source.index_offset[0] = source_index_offset.x; d.source.index_offset[1] = source_index_offset.y; d.source.index_offset[2] = source_index_offset.z;
        d.source.intensity = source_intensity;
    }
    return d;
}

wost_scene_desc Problem<2>::scene_desc(int width, int height) const
{
    wost_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    if (enable_dirichlet) {
        d.dirichlet.n_verts = (int32_t)scene_stat.dirichlet_vertices_size;
        d.dirichlet.n_segs = (int32_t)scene_stat.dirichlet_primitives_size;
        d.dirichlet.verts = scene_dirichlet_loader->vertices.data();
        d.dirichlet.segs = scene_dirichlet_loader->indices.data();
        d.dirichlet.colors = vertex_color_dirichlet.data();
    }
    if (enable_neumann) {
        d.neumann.n_verts = (int32_t)scene_stat.neumann_vertices_size;
        d.neumann.n_segs = (int32_t)scene_stat.neumann_primitives_size;
        d.neumann.verts = scene_neumann_loader->vertices.data();
        d.neumann.segs = scene_neumann_loader->indices.data();
        d.neumann.colors = vertex_color_neumann.data();
    }
    d.dirichlet_intensity = dirichlet_intensity;
    d.neumann_intensity = neumann_intensity;
    d.probe_scale = mpProbe->mData.scale;
    d.probe_pos[0] = mpProbe->mData.pos.x; d.probe_pos[1] = mpProbe->mData.pos.y;
    d.probe_up[0] = mpProbe->mData.up.x; d.probe_up[1] = mpProbe->mData.up.y;
    if (!mask.empty()) {
        if (mask.size() != (size_t)width * height) throw std::runtime_error("mask size does not match the frame");
        d.mask = mask.data();
    }
    if (enable_source) {
        d.source.nx = source_nx; d.source.ny = source_ny;
        d.source.rgb = source_rgb.data();
        d.source.index_scale[0] = source_index_scale.x; d.source.index_scale[1] = source_index_scale.y;
        d.source.index_offset[0] = source_index_offset.x; d.source.index_offset[1] = source_index_offset.y;
        d.source.intensity = source_intensity;
    }
    return d;
}

}  // namespace elaina
