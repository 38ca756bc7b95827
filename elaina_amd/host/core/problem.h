// problem.h -- scene description, mirror of the reference's Problem<2> surface
// (core/problem.h:54-194): loadConfig(json) plus the getters the integrator and run_expr
// touch.  Where the reference hands out snch-lbvh device handles
// (get_problem_*_bvh_device / get_problem_*_ptr), this class hands out the host arrays as a
// wost_scene_desc: the LBVH lives behind the C-ABI (include/wost.h).
#pragma once
#include <memory>
#include <optional>
#include <vector>

#include "../../../include/wost.h"
#include "common.h"
#include "evaluation_grid.h"

namespace elaina {

struct ProblemStatistics {
    std::size_t dirichlet_vertices_size{0};
    std::size_t dirichlet_primitives_size{0};
    std::size_t neumann_vertices_size{0};
    std::size_t neumann_primitives_size{0};
};

// OBJ polyline loader: replaces lbvh::scene_loader<2> (reference core/problem.cu:29,46).
// Reads "v x y [z]" and "l i j k ..." (1-based, negative = relative) records.
struct SceneLoader2 {
    explicit SceneLoader2(const string &path);
    std::vector<float> vertices;   // x,y per vertex
    std::vector<int32_t> indices;  // i0,i1 per segment, 0-based
};

template <unsigned int DIM> class Problem;

template <> class Problem<2> {
public:
    explicit Problem(const bool verbose = true) : verbose(verbose) {}
    using SceneProbe = EvaluationGrid<2>;

    void loadConfig(const json &config, const fs::path &search_dir = {});

    // reference core/problem.h:104-171
    bool isDirichletEnabled() const { return enable_dirichlet; }
    bool isNeumannEnabled() const { return enable_neumann; }
    bool isSourceEnabled() const { return enable_source; }
    const SceneProbe &getProbe() const { return *mpProbe; }
    const AABB2f &getAABB() const { return mAABB; }
    const ProblemStatistics &get_problem_stat() const { return scene_stat; }
    float get_source_intensity() const { return source_intensity; }
    float get_dirichlet_intensity() const { return dirichlet_intensity; }
    float get_neumann_intensity() const { return neumann_intensity; }
    const std::vector<float> &get_vertex_color_dirichlet() const { return vertex_color_dirichlet; }
    const std::vector<float> &get_vertex_color_neumann() const { return vertex_color_neumann; }
    const std::vector<uint8_t> &get_mask() const { return mask; }
    void set_mask(std::vector<uint8_t> m) { mask = std::move(m); }
    // dense source grid (stands in for the reference's nanovdb grid, core/problem.cu:136-149):
    // rgb = ny*nx*3 floats, index = world * scale + offset
    void set_source(int nx, int ny, std::vector<float> rgb, Vector2f index_scale, Vector2f index_offset);

    // what crosses the C-ABI (valid while this Problem is alive and unchanged)
    wost_scene_desc scene_desc(int width, int height) const;

private:
    std::shared_ptr<SceneProbe> mpProbe;
    AABB2f mAABB;
    std::unique_ptr<SceneLoader2> scene_dirichlet_loader, scene_neumann_loader;
    std::vector<float> vertex_color_dirichlet, vertex_color_neumann;  // 6 floats per vertex
    bool enable_dirichlet{false}, enable_neumann{false}, enable_source{false};
    int source_nx{0}, source_ny{0};
    std::vector<float> source_rgb;
    Vector2f source_index_scale{1.0f, 1.0f}, source_index_offset{0.0f, 0.0f};
    bool verbose{false};
    ProblemStatistics scene_stat;
    float source_intensity{1.0f}, dirichlet_intensity{1.0f}, neumann_intensity{1.0f};
    std::vector<uint8_t> mask;  // empty = all pixels on
};

// OBJ triangle loader: replaces lbvh::scene_loader<3> (reference core/problem.h:207).
// Reads "v x y z" and "f i j k ..." (1-based, negative = relative, "i/vt/vn" accepted; polygons are fanned).
struct SceneLoader3 {
    explicit SceneLoader3(const string &path);
    std::vector<float> vertices;   // x,y,z per vertex
    std::vector<int32_t> indices;  // i0,i1,i2 per triangle, 0-based
};

// Problem<3> (reference core/problem.h:197-260): triangle meshes, EvaluationGrid<3>.  The source term (a nanovdb
// volume in the reference) comes as a dense grid: "source_grid" with "nz" and three-component scale / offset.
template <> class Problem<3> {
public:
    explicit Problem(const bool verbose = true) : verbose(verbose) {}
    using SceneProbe = EvaluationGrid<3>;

    void loadConfig(const json &config, const fs::path &search_dir = {});
    bool isDirichletEnabled() const { return enable_dirichlet; }
    bool isNeumannEnabled() const { return enable_neumann; }
    bool isSourceEnabled() const { return enable_source; }
    const SceneProbe &getProbe() const { return *mpProbe; }
    const AABB3f &getAABB() const { return mAABB; }      // scene.aabb, three components (the guided integrator's box)
    bool hasAABB() const { return has_aabb; }
    const ProblemStatistics &get_problem_stat() const { return scene_stat; }
    float get_source_intensity() const { return source_intensity; }
    // dense source grid [nz][ny][nx][3] (stands in for the reference's nanovdb grid, core/problem.cu:136-149):
    // index = world * index_scale + index_offset per axis, trilinear
    void set_source(int nx, int ny, int nz, std::vector<float> rgb, Vector3f index_scale, Vector3f index_offset);
    float get_dirichlet_intensity() const { return dirichlet_intensity; }
    float get_neumann_intensity() const { return neumann_intensity; }
    const std::vector<uint8_t> &get_mask() const { return mask; }
    void set_mask(std::vector<uint8_t> m) { mask = std::move(m); }
    wost3_scene_desc scene_desc(int width, int height) const;

private:
    std::shared_ptr<SceneProbe> mpProbe;
    AABB3f mAABB;
    bool has_aabb{false};
    std::unique_ptr<SceneLoader3> scene_dirichlet_loader, scene_neumann_loader;
    std::vector<float> vertex_color_dirichlet, vertex_color_neumann;
    bool enable_dirichlet{false}, enable_neumann{false}, enable_source{false};
    int source_nx{0}, source_ny{0}, source_nz{0};
    std::vector<float> source_rgb;
    Vector3f source_index_scale{1.0f, 1.0f, 1.0f}, source_index_offset{0.0f, 0.0f, 0.0f};
    bool verbose{false};
    ProblemStatistics scene_stat;
    float source_intensity{1.0f}, dirichlet_intensity{1.0f}, neumann_intensity{1.0f};
    std::vector<uint8_t> mask;
};

// reference core/problem.cu:63-96: {"ColorConfigurations":[{"vertexID":i+1,"leftColor":{R,G,B},"rightColor":{R,G,B}}]}
std::vector<float> parseVertexColorFile(const string &path);

}  // namespace elaina
