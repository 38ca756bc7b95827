/*
 * wost_oracle.h -- CPU oracle for the wavefront Walk-on-Stars hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under elaina_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the reported
 * host baseline.
 *
 * What it restates (all paths relative to /root/reference):
 *   integrator/uniform/integrator.cu:65-100,103-212,215-232,319-445,448-526,529-623
 *   integrator/uniform/workitem.h:12-61, workqueue.h:25-29,99-110
 *   core/sampler.h:20-98, util/hash.h:13-28, core/evaluation_grid.h:27-33
 *   util/sampling.h:29-42,80-94,112-115, util/green.h:15-74
 *   util/transformation.h:30-55, util/math_utils.h:153-156,319-322
 *   integrator/common.h:242-260, core/math/include/krrmath/functors.h:60-92
 *
 * PARITY STATUS: the geometric queries of the path live in the third-party
 * submodule ext/lbvh = tyanyuy3125/snch-lbvh (.gitmodules:16-18, pinned commit
 * unknown, directory empty in /root/reference), and the reference has no test
 * or golden vector for the integrators.  The pieces that ARE pinned here:
 * PCG32 against the canonical pcg32 known-answer vector (the reference's
 * sampler is the canonical generator, core/sampler.h:20-27,65-72) and the
 * per-pixel seeding vectors of SURVEY.md 8(c).  For the lbvh boundary the
 * oracle implements the mathematical definition of each query (exact closest
 * point / closest silhouette vertex / first hit) and is checked against an
 * O(N) brute force.  => "parity unpinned" at the snch-lbvh boundary.
 *
 * Deterministic arithmetic (so that the HIP path can be compared bit for bit):
 *   - fp32 everywhere except PCG32 (uint64), compiled with -ffp-contract=off;
 *   - fmaf() only where written explicitly;
 *   - sin/cos/log are the polynomial kernels specified in DESIGN.md
 *     ("deterministic math"), not libm (build with -DWOST_ORACLE_LIBM to get
 *     the literal std::cos/std::sin/std::log restatement for statistical
 *     cross-checks).
 */
#ifndef WOST_ORACLE_H
#define WOST_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wo_mesh {
    int n_verts;
    int n_segs;
    const float *verts;   /* n_verts * 2 (x,y)                                  */
    const int *segs;      /* n_segs  * 2 (i0,i1), 0-based                       */
    const float *colors;  /* n_verts * 6 (left rgb, right rgb) or NULL = zeros  */
} wo_mesh;

/* Source term f of  laplace(u) = -f  as a dense 2-D grid of RGB samples at integer index
 * coordinates, index = world * index_scale + index_offset, bilinear, zero outside: what the
 * reference reads from a nanovdb Vec3f grid at z = 0 (worldToIndex + order-1 SampleFromVoxels,
 * integrator/uniform/integrator.cu:303-306; nanovdb is absent => PARITY UNPINNED). */
typedef struct wo_source {
    int nx, ny;               /* nx == 0 -> no source term */
    const float *rgb;         /* ny * nx * 3, x fastest */
    float index_scale[2], index_offset[2];
    float intensity;          /* source_intensity (core/problem.cu:179) */
} wo_source;

typedef struct wo_scene {
    wo_mesh dirichlet;    /* n_segs == 0  -> Dirichlet disabled */
    wo_mesh neumann;      /* n_segs == 0  -> Neumann disabled   */
    float dirichlet_intensity;
    float neumann_intensity;
    float probe_scale;
    float probe_pos[2];
    float probe_up[2];
    const unsigned char *mask; /* width*height bytes (0 = masked out) or NULL   */
    wo_source source;
} wo_scene;

typedef struct wo_settings {
    int width;
    int height;
    int spp;
    int max_depth;
    float eps_shell;
} wo_settings;

typedef struct wo_stats {
    uint64_t walk_steps;      /* sum over depths of the size of the eval queue  */
    uint64_t walks_started;
    uint64_t walks_absorbed;  /* terminated in the epsilon shell                */
    uint64_t walks_truncated; /* reached max_depth                              */
    uint64_t neumann_hits;    /* steps that landed on the Neumann boundary      */
    double seconds;
} wo_stats;

/* ---- PCG32 (core/sampler.h) -------------------------------------------- */
typedef struct wo_pcg { uint64_t state, inc; } wo_pcg;
void wo_pcg_set_seed(wo_pcg *r, uint64_t initstate, uint64_t initseq);
uint32_t wo_pcg_next_uint(wo_pcg *r);
float wo_pcg_next_float(wo_pcg *r);
double wo_pcg_next_double(wo_pcg *r);
void wo_pcg_advance(wo_pcg *r, int64_t delta);
uint32_t wo_interleave_32bit(uint32_t x, uint32_t y);
/* prepareSolve seeding of one pixel (integrator.cu:71-77) */
void wo_pcg_seed_pixel(wo_pcg *r, int pixel_id, int width);

/* ---- deterministic math ------------------------------------------------ */
void wo_sincos_2pi(float u, float *c, float *s);   /* cos/sin(2*pi*u), u in [0,1) */
float wo_logf(float x);                            /* natural log, x > 0 finite   */

/* ---- evaluation grid --------------------------------------------------- */
void wo_eval_point(const wo_scene *sc, int px, int py, int width, int height,
                   float *x, float *y);

/* ---- geometric queries (definition of the lbvh boundary) ---------------- */
/* closest point: returns per query the winning segment (original index),
 * the distance, the unclamped projection ratio and the side.
 * mode 0 = brute force O(N), 1 = oracle's own BVH. */
int wo_closest_point_batch(const wo_mesh *mesh, const float *pts, int n, int mode,
                           int *out_idx, float *out_dist, float *out_uv, int *out_side);
/* closest silhouette vertex distance, search radius limited to rmax
 * (pass INFINITY for unbounded); returns INFINITY when there is none. */
int wo_closest_silhouette_batch(const wo_mesh *mesh, const float *pts, const float *rmax,
                                int n, float *out_dist);
/* closest-hit ray query: hit flag, t, segment index. */
int wo_ray_intersect_batch(const wo_mesh *mesh, const float *origins, const float *dirs,
                           const float *tmax, int n, int *out_hit, float *out_t, int *out_idx);

/* ---- the solver --------------------------------------------------------- */
/* Solves pixels [pixel_begin, pixel_end) of the width*height frame with
 * n_threads host threads.  field_rgb: (pixel_end-pixel_begin)*3 floats,
 * steps_per_pixel (optional): (pixel_end-pixel_begin) uint32,
 * depth_hist (optional): max_depth uint64 (queue size per depth, summed over spp). */
int wo_solve(const wo_scene *sc, const wo_settings *st, int pixel_begin, int pixel_end,
             int n_threads, float *field_rgb, uint32_t *steps_per_pixel,
             uint64_t *depth_hist, wo_stats *stats);

/* SOURCE channel (renderSource, integrator/common.h:126-163): intensity * f at every pixel. */
int wo_render_source(const wo_scene *sc, const wo_settings *st, float *out_rgb);

/* Dirichlet SDF channel (integrator/common.h:52-85): distance per pixel. */
int wo_render_dirichlet_sdf(const wo_scene *sc, const wo_settings *st, int n_threads,
                            float *out_dist);

/* ---- 3-D uniform path (oracle/wost_oracle3d.c) --------------------------- */
typedef struct wo3_mesh {
    int n_verts;
    int n_tris;
    const float *verts;   /* n_verts * 3 */
    const int *tris;      /* n_tris * 3, 0-based */
    const float *colors;  /* n_verts * 6 (colour on the side the normal points to, colour on the other side) or NULL */
} wo3_mesh;
/* Source term f of  laplace(u) = -f  in 3-D: a dense grid of RGB samples at integer index coordinates,
 * index = world * index_scale + index_offset per axis, trilinear (order-1) interpolation, zero outside -- what
 * nanovdb's worldToIndex + SampleFromVoxels<.., 1> compute on the reference's Vec3fGrid (core/problem.cu:136-149,
 * integrator/uniform/integrator.cu:296-304); the nanovdb file format itself cannot be read here. */
typedef struct wo3_source {
    int nx, ny, nz;                       /* nx == 0 -> no source term */
    const float *rgb;                     /* [nz][ny][nx][3] */
    float index_scale[3], index_offset[3];
    float intensity;
} wo3_source;
typedef struct wo3_scene {
    wo3_mesh dirichlet, neumann;          /* n_tris == 0 -> disabled */
    float dirichlet_intensity, neumann_intensity;
    float probe_scale;                    /* EvaluationGrid<3>::ProbeData (core/evaluation_grid.h:48-55) */
    float probe_pos[3], probe_up[3], probe_right[3];
    const unsigned char *mask;
    wo3_source source;
} wo3_scene;
void wo3_source_eval(const wo3_source *src, float x, float y, float z, float out[3]);
int wo3_solve(const wo3_scene *sc, const wo_settings *st, int pixel_begin, int pixel_end, int n_threads, float *field_rgb,
              wo_stats *stats);
/* closest triangle (lowest index on ties), distance, barycentric (u, v) of the projection, side */
int wo3_closest_point_batch(const wo3_mesh *mesh, const float *pts, int n, int *out_idx, float *out_dist, float *out_uv,
                            int *out_side);
int wo3_closest_silhouette_batch(const wo3_mesh *mesh, const float *pts, const float *rmax, int n, float *out_dist);
int wo3_ray_intersect_batch(const wo3_mesh *mesh, const float *origins, const float *dirs, const float *tmax, int n,
                            int *out_hit, float *out_t, int *out_idx);
void wo3_green_ball(float R, float r, float *eval, float *norm, float *pdf_radius);
/* debug channels of the 3-D integrator at the evaluation points of the frame (integrator/common.h:52-163 with DIM = 3):
 * which = 0 distance to the Dirichlet mesh, 1 distance to the closest silhouette edge of the Neumann mesh
 * (+inf without that mesh); the source channel = intensity * f (zeros without a source term) */
/* von Mises-Fisher lobe (util/vmf.h): density by cos(theta); directions about mu[n*3] from PCG32 streams setSeed(seed[i], 1) */
float wo3_vmf_eval(float kappa, float cos_theta);
void wo3_vmf_sample(float kappa, const float mu[3], wo_pcg *rng, float out[3]);
int wo3_vmf_eval_batch(const float *kappa, const float *cos_theta, int n, float *pdf);
int wo3_vmf_sample_batch(const float *kappa, const float *mu, const uint64_t *seed, int n, int per_point, float *dirs);
/* VMM<3,8> (distribution.h:279-436): raw = 40 floats per point (8 x (lambda, kappa, mean vector)); loss gradients over 41
 * (with the selection logit), reference record dir[3], Li, dirPdf, onNeumann, normal[3] */
int wo3_vmm_pdf_sample(const float *raw, const float *wi, const uint64_t *seed, int n, float *pdf, float *dir);
int wo3_vmm_loss_gradients(const float *raw, const float *dir, const float *li, const float *dir_pdf,
                           const unsigned char *on_neumann, const float *normal, int n, float loss_scale,
                           float *dl_draw, float *likelihood);
int wo3_render_sdf(const wo3_scene *sc, const wo_settings *st, int which, float *out_dist);
int wo3_render_source(const wo3_scene *sc, const wo_settings *st, float *out_rgb);

/* ---- guided path, deterministic distribution layer (oracle/wost_vmm.c) ---- */
float wo_eval_poly_large0(float y);
float wo_log_bessel(float x, int order);
int wo_vonmises_eval(const float *kappa, const float *cos_theta, int n, float *log_i0, float *log_i1,
                     float *log_pdf, float *dlog_dkappa);
int wo_vonmises_sample(const float *kappa, const uint64_t *seed, int n, int per_point, float *theta);
int wo_vmm_pdf_sample(const float *raw, const float *wi, const uint64_t *seed, int n, float *pdf, float *dir);
int wo_vmm_loss_gradients(const float *raw, const float *dir, const float *li, const float *dir_pdf,
                          const unsigned char *on_neumann, const float *normal, int n, float loss_scale,
                          float *dl_draw, float *likelihood);

/* ---- guiding network (oracle/wost_net.c) -------------------------------------------- */
typedef struct wo_net_config {
    int n_levels, n_features, base_resolution;
    float per_level_scale;
    int n_neurons, n_hidden_layers, n_output, n_output_padded;
    float learning_rate, beta1, beta2, epsilon, l2_reg, ema_decay;
} wo_net_config;
uint64_t wo_net_n_params(const wo_net_config *c);
int wo_net_levels(const wo_net_config *c, int *res, float *scale);
int wo_net_forward(const wo_net_config *c, const float *params, const float *xy, int n, float *out, float *acts);
int wo_net_backward(const wo_net_config *c, const float *params, const float *xy, const float *dl_dout, int n,
                    float *grad);
int wo_net_optimizer_step(const wo_net_config *c, float *params, float *m1, float *m2, float *ema_raw,
                          float *inference_params, const float *grad, int step, float loss_scale, uint32_t *param_steps);
/* the same network with THREE inputs (GuidedIntegrator<3>, guided/parameters.h:26-33): trilinear DenseGrid, res^3 entries per level */
uint64_t wo_net3_n_params(const wo_net_config *c);
int wo_net3_forward(const wo_net_config *c, const float *params, const float *xyz, int n, float *out, float *acts);
int wo_net3_backward(const wo_net_config *c, const float *params, const float *xyz, const float *dl_dout, int n, float *grad);
int wo_net3_optimizer_step(const wo_net_config *c, float *params, float *m1, float *m2, float *ema_raw,
                           float *inference_params, const float *grad, int step, float loss_scale, uint32_t *param_steps);

/* ---- guided integrator (oracle/wost_guided.c) ------------------------------------------ */
typedef struct wo_guided_settings {
    int width, height, spp, max_depth;
    float eps_shell;
    int train_spp_count;                    /* GuidedIntegratorSettings, integrator/guided/integrator.h:54-75 */
    float uniform_fraction_training, uniform_fraction_guiding;
    int max_guided_depth_training, max_guided_depth_guiding;
    float aabb_min[2], aabb_max[2];         /* scene.aabb of the JSON configuration */
    int max_train_depth;                    /* 3       integrator.h:237 */
    int batch_size;                         /* 524288  parameters.h:11  */
    int min_batch_size;                     /* 65536   parameters.h:12  */
    int batches_per_spp;                    /* 5       integrator.h:238 */
    int train_pixel_stride, train_pixel_offset; /* 1, 0 guided.h:104-121 */
    float loss_scale;                       /* 128     parameters.h:14  */
} wo_guided_settings;

typedef struct wo_guided_stats {
    uint64_t walk_steps, walks_started, walks_absorbed, walks_truncated, neumann_hits;
    uint64_t guided_steps;                  /* steps whose direction came from the mixture */
    uint64_t train_samples, optimizer_steps;
} wo_guided_stats;

/* training set of one sample pass, in (pixel, record) order; arrays may be NULL */
typedef struct wo_train_dump {
    int capacity, n;
    float *xy, *dir, *solution, *dir_pdf, *normal;
    unsigned char *on_neumann;
} wo_train_dump;

/* params: n_params floats, initial weights in, trained weights out. */
int wo_solve_guided(const wo_scene *sc, const wo_guided_settings *gs, const wo_net_config *nc, float *params,
                    int n_threads, float *field_rgb, wo_guided_stats *stats, int dump_spp, wo_train_dump *dump);

/* ---- GuidedIntegrator<3> (oracle/wost_oracle3d.c) ---------------------------------------------- */
typedef struct wo3_guided_settings {
    int width, height, spp, max_depth;
    float eps_shell;
    int train_spp_count;
    float uniform_fraction_training, uniform_fraction_guiding;
    int max_guided_depth_training, max_guided_depth_guiding;
    float aabb_min[3], aabb_max[3];
    int max_train_depth, batch_size, min_batch_size, batches_per_spp, train_pixel_stride, train_pixel_offset;
    float loss_scale;
} wo3_guided_settings;
typedef struct wo3_train_dump {
    int capacity, n;
    float *xyz, *dir, *solution, *dir_pdf, *normal;      /* xyz: normalised network inputs; dir / normal: 3 floats per sample */
    unsigned char *on_neumann;
} wo3_train_dump;
/* nc: n_output 41 (guided/parameters.h:26-33), params: wo_net3_n_params floats (in: initial, out: trained).  Returns -3 for a
 * scene with a source term (not restated for the 3-D guided solve). */
int wo3_solve_guided(const wo3_scene *sc, const wo3_guided_settings *gs, const wo_net_config *nc, float *params, int n_threads,
                     float *field_rgb, wo_guided_stats *stats, int dump_spp, wo3_train_dump *dump);

const char *wo_version(void);

#ifdef __cplusplus
}
#endif
#endif
