/*
 * wost.h -- C-ABI of the MI355X-native Walk-on-Stars hot path (libwost_hip.so).
 *
 * This is the drop-in boundary for the ONE path this repository accelerates: the
 * wavefront walk loop of Elaina's uniform integrator and the snch-lbvh queries it
 * calls.  The reference has no FFI; what `run_expr` touches is the C++ class shape
 * of `UniformIntegrator<2>` / `Problem<2>` (reference exec.cu:77-78,145-215).  The
 * host-side mirror of those classes (elaina_amd/host/) calls only the functions
 * declared here.  Plain pointers and sizes, no C++/torch types, no exceptions
 * across the boundary, caller owns every host buffer, callee owns device memory,
 * one handle per GPU, handles are independent.
 *
 * Every entry point returns WOST_OK (0) or a negative error code; the message of
 * the last error on the calling thread is available from wost_last_error().
 * There is NO CPU fallback: without a usable HIP device wost_create() fails.
 */
#ifndef WOST_H
#define WOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WOST_OK 0
#define WOST_ERR_INVALID -1      /* bad argument                                 */
#define WOST_ERR_DEVICE -2       /* HIP runtime error / no device                */
#define WOST_ERR_UNSUPPORTED -3  /* valid request this build does not cover      */
#define WOST_ERR_NOMEM -4

/* which mesh a query refers to */
#define WOST_MESH_DIRICHLET 0
#define WOST_MESH_NEUMANN 1

/* Host description of one boundary mesh (2-D polylines).
 * Replaces: lbvh::scene_loader<2> + lbvh::scene<2>(vb,ve,ib,ie) + compute_silhouettes()
 * + build_bvh() (reference core/problem.cu:27-60) and the per-vertex colour pairs of
 * loadVertexColorFileImpl (core/problem.cu:99-133). */
typedef struct wost_mesh_desc {
    int32_t n_verts;
    int32_t n_segs;        /* 0 => this boundary type is disabled (problem.h:104-111)   */
    const float *verts;    /* n_verts * 2, (x, y)                                        */
    const int32_t *segs;   /* n_segs * 2, 0-based (i0, i1); direction p0 -> p1           */
    const float *colors;   /* n_verts * 6: (left r,g,b, right r,g,b); NULL => all zero   */
} wost_mesh_desc;

/* Source term f of  laplace(u) = -f  (SURVEY 8f.2).  Replaces the nanovdb Vec3f grid of
 * Problem<2>::loadSource (core/problem.cu:136-149) as the integrator reads it
 * (worldToIndex + order-1 SampleFromVoxels at z = 0, integrator/uniform/integrator.cu:303-306): a
 * dense 2-D grid of RGB samples at integer index coordinates, index = world * index_scale +
 * index_offset, bilinear, zero outside.  nx == 0 disables the source term. */
typedef struct wost_source_desc {
    int32_t nx, ny;
    const float *rgb;            /* ny * nx * 3, x fastest                                      */
    float index_scale[2], index_offset[2];
    float intensity;             /* source_intensity (core/problem.cu:179)                      */
} wost_source_desc;

/* Replaces Problem<2> as seen by the integrator (core/problem.h:104-171) and the
 * probe EvaluationGrid<2>::ProbeData (core/evaluation_grid.h:16-23). */
typedef struct wost_scene_desc {
    wost_mesh_desc dirichlet;
    wost_mesh_desc neumann;
    float dirichlet_intensity;   /* problem.h:155 */
    float neumann_intensity;     /* problem.h:159 */
    float probe_scale;           /* evaluation_grid.h:18 */
    float probe_pos[2];          /* evaluation_grid.h:19 */
    float probe_up[2];           /* evaluation_grid.h:20 */
    const uint8_t *mask;         /* width*height bytes, 0 = pixel masked out; NULL = all on
                                    (problem.h:163; the reference's fixed 1024^2 default mask,
                                    core/problem.cu:245-247, is sized to the frame here)     */
    wost_source_desc source;     /* zero-initialised = no source term                          */
} wost_scene_desc;

/* Replaces UniformIntegratorSettings (integrator/uniform/integrator.h:27-48); the
 * metric-dump keys are host-side concerns and do not cross the boundary. */
typedef struct wost_settings {
    int32_t width;         /* frameSize[0] */
    int32_t height;        /* frameSize[1] */
    int32_t spp;           /* samplesPerPixel */
    int32_t max_depth;     /* maxWalkingDepth */
    float eps_shell;       /* epsilonShell */
} wost_settings;

typedef struct wost_stats {
    uint64_t walk_steps;       /* sum over depths of the evaluation-queue size   */
    uint64_t walks_started;
    uint64_t walks_absorbed;   /* ended in the epsilon shell                                  */
    uint64_t walks_truncated;  /* reached max_depth                                           */
    uint64_t neumann_hits;     /* steps that landed on the Neumann boundary                   */
    uint64_t inner_visits;     /* LBVH inner nodes expanded by the walk's closest-point queries */
    uint64_t leaf_visits;      /* LBVH leaves (4 segments each) evaluated by those queries    */
    uint64_t trav_trips;       /* wave-level scheduler diagnostics: traversal-phase trips ...  */
    uint64_t step_trips;       /* ... and step-phase trips, summed over waves                 */
    double solve_ms;           /* host wall time of the call, like UniformIntegrator::solve() */
    double kernel_ms;          /* sum of HIP-event durations of the walk kernel launches      */
    uint32_t kernel_launches;  /* number of walk-kernel launches (rounds)                     */
    uint32_t reserved;
} wost_stats;

typedef struct wost_context *wost_handle;

/* Upload the scene, build both LBVHs, allocate walk-state queues for the frame.
 * Replaces Problem<2>::loadConfig geometry upload + UniformIntegrator<2> ctor
 * (integrator/uniform/integrator.cu:626-633 -> initializeImpl :18-62). */
int wost_create(const wost_scene_desc *scene, const wost_settings *settings, int device,
                wost_handle *out);

/* UniformIntegrator<2>::solve() (integrator/uniform/integrator.cu:666-672 -> solveImpl
 * :529-623) for the pixels [pixel_begin, pixel_end) of the frame, row-major pixelId as in
 * the reference.  field_rgb receives (pixel_end - pixel_begin) * 3 floats = solution / spp,
 * i.e. the RGB of the Film after the resolve pass (:614-621).  Results depend only on
 * (pixelId, frame width, geometry, settings), never on the range, so shards concatenate
 * to exactly the full-frame result. */
int wost_solve(wost_handle h, int32_t pixel_begin, int32_t pixel_end, float *field_rgb,
               wost_stats *stats);

/* Same walk, sharded by 64-pixel tiles for multi-GPU load balance: this call owns the tiles
 * t with t % shard_count == shard_index.  field_rgb_dev is a DEVICE buffer of
 * width*height*3 floats that the caller has zero-filled; only owned pixels are written, so
 * a sum-reduce over ranks (RCCL) yields the full field.  `stream` is a hipStream_t (NULL =
 * default stream); the call returns after the stream work has completed. */
int wost_solve_sharded(wost_handle h, int32_t shard_index, int32_t shard_count,
                       float *field_rgb_dev, void *stream, wost_stats *stats);

/* renderDirichletSDF / renderSilhouetteSDF (integrator/common.h:52-123): one query per
 * pixel of the frame, out receives width*height distances. */
int wost_render_sdf(wost_handle h, int which_mesh, float *out_dist);

/* renderSource (integrator/common.h:126-163): intensity * f at every pixel of the frame, out
 * receives width*height*3 floats (zeros without a source term). */
int wost_render_source(wost_handle h, float *out_rgb);

/* lbvh::query_device(bvh, lbvh::nearest(q), distance_calculator()) + checkPointSide +
 * computeProjectionRatio for a batch of host points (call sites
 * integrator/uniform/integrator.cu:138,148-149).  out_idx = original segment index. */
int wost_closest_point(wost_handle h, int which_mesh, const float *pts, int32_t n,
                       int32_t *out_idx, float *out_dist, float *out_uv, int32_t *out_side);

/* lbvh::query_device(bvh, nearest_silhouette(q,false), silhouette_distance_calculator())
 * (integrator.cu:189); rmax (optional, per point) bounds the search radius. */
int wost_closest_silhouette(wost_handle h, int which_mesh, const float *pts, const float *rmax,
                            int32_t n, float *out_dist);

/* lbvh::query_device(bvh, ray_intersect(ray(o,d), tmax), intersect_test()) closest hit
 * (integrator.cu:500-503). */
int wost_ray_intersect(wost_handle h, int which_mesh, const float *origins, const float *dirs,
                       const float *tmax, int32_t n, int32_t *out_hit, float *out_t,
                       int32_t *out_idx);

/* ---- guided path, deterministic distribution layer (first slice of SURVEY 8a row a24) -------
 * Batch entry points over HOST arrays; no handle, `device` selects the GPU. */

/* logModifiedBesselFn(x, 0|1), VonMises::log_eval(cos), VonMises::d_log_eval_d_kappa(cos)
 * (reference util/vonmises.h:75-93,128-163) for n (kappa, cos_theta) pairs; any output may
 * be NULL. */
int wost_vonmises_eval(int device, const float *kappa, const float *cos_theta, int32_t n,
                       float *log_i0, float *log_i1, float *log_pdf, float *dlogpdf_dkappa);

/* rejectionSample(kappa, proposalR, sampler) (util/vonmises.h:95-118): per point a PCG32
 * stream setSeed(seed[i], 1) and per_point consecutive angles in theta[n*per_point]. */
int wost_vonmises_sample(int device, const float *kappa, const uint64_t *seed, int32_t n,
                         int32_t per_point, float *theta);

/* VMM<2,8> built from 32 raw network outputs per point (integrator/guided/distribution.h:
 * 146-168, train.h:50-79): mixture pdf at direction wi[n*2] (pdf may be NULL) and one sampled
 * direction per point with the stream setSeed(seed[i], 1) (distribution.h:186-198;
 * sample_dir may be NULL). */
int wost_vmm_pdf_sample(int device, const float *raw, const float *wi, const uint64_t *seed,
                        int32_t n, float *pdf, float *sample_dir);

/* Training-side gradients: VMM<2,8>::gradients_probability
 * (distribution.h:201-264) chained with compute_dL_doutput_divergence (train.h:492-553) for n
 * samples: raw[n*33] network outputs (8 lobes + selection logit), reference record per sample
 * (dir[n*2], li = mean |solution/thp|, dir_pdf, on_neumann flags (may be NULL), normal[n*2]),
 * loss_scale (the reference uses 128, divided by n inside).  Outputs dL/draw[n*33] and the
 * per-sample likelihood term (may be NULL). */
int wost_vmm_loss_gradients(int device, const float *raw, const float *dir, const float *li,
                            const float *dir_pdf, const uint8_t *on_neumann, const float *normal,
                            int32_t n, float loss_scale, float *dl_draw, float *likelihood);

/* ---- guiding network (SURVEY 8a rows a22/a23 + the optimizer of a27) -------------------------
 * Replaces the tiny-cuda-nn objects of the guided integrator: the NetworkWithInputEncoding
 * variant of util/network.h:21-196 (inference 39-47, forward 49-60, backward 62-92, parameter
 * order 94-136), built in integrator/guided/integrator.cu:1095-1131 from the configuration of
 * data/ladybug/n.json:49-81: DenseGrid encoding (n_levels x n_features_per_level, linear
 * interpolation) -> bias-free ReLU MLP -> n_output raw values per point; Adam nested in a
 * debiased EMA.  Parameter vector order as in util/network.h:99-117 (network first, encoding
 * second): [W1 n_neurons x enc][W2..][Wout pad16(n_output) x n_neurons][grid levels].
 * fp32 arithmetic by default (the bit-exact mode); wost_net_set_option("precision" / "train_precision", 16) selects the
 * reference's half precision for the inference and for the training passes (f16 MFMAs, fp32 master weights). */
typedef struct wost_net_config {
    int32_t n_levels, n_features_per_level, base_resolution;
    float per_level_scale;
    int32_t n_neurons, n_hidden_layers, n_output;
    float learning_rate, beta1, beta2, epsilon, l2_reg, ema_decay;
} wost_net_config;
typedef struct wost_net *wost_net_handle;

/* integrator/guided/integrator.cu:1095-1131 (network + optimizer + trainer construction):
 * allocates parameters on `device` and initialises them from `seed` (util/network.h:113-136;
 * MLP xavier-uniform, grid uniform(-1e-4, 1e-4)). */
int wost_net_create(int device, const wost_net_config *cfg, uint64_t seed, wost_net_handle *out);
int wost_net_destroy(wost_net_handle h);
int wost_net_n_params(wost_net_handle h, uint64_t *n_total, uint64_t *n_mlp);
/* which: 0 = training parameters, 1 = inference (EMA) parameters, 2 = gradients of the last
 * wost_net_train_step (scaled by loss_scale). */
int wost_net_get_params(wost_net_handle h, int which, float *host);
/* Gradients are accumulated as 64-bit fixed point (value * 2^36) with integer atomics, so they do
 * not depend on the order of summation.  A multi-GPU caller may supply the accumulation buffer
 * itself (device memory, n_total int64; NULL = internal again), e.g. memory it can hand to a
 * collective. */
int wost_net_set_gradient_buffer(wost_net_handle h, void *dev_int64);
/* sets training and inference parameters, resets the optimizer state */
int wost_net_set_params(wost_net_handle h, const float *host);
/* network->inference (integrator/guided/integrator.cu:560,597; util/network.h:39-47): xy[n*2] in [0,1]^2 -> out[n*n_output];
 * use_inference_params = 1 evaluates the EMA weights (what rendering uses), 0 the training ones. */
int wost_net_inference(wost_net_handle h, const float *xy, int32_t n, float *out, int use_inference_params);
/* "precision": 32 (default) = fp32 everywhere, bit-exact against the CPU restatement; 16 = the reference's own
 * network precision for inference (tiny-cuda-nn FullyFusedMLP + grid in half, util/network.h:21-196,
 * data/ladybug/n.json:61-67): f16 weights / activations / grid values, fp32 accumulation on
 * v_mfma_f32_16x16x16_f16; the guided solve then runs a whole sample per launch with the network evaluated inside
 * the walk kernel (same field and records as the per-depth launches).
 * "train_precision": 32 (default) or 16 = forward / backward / weight-gradient passes of a training step in that
 * half-precision arithmetic (tiny-cuda-nn trains in half with loss scale 128, guided/parameters.h:13); with 16,
 * wost_net_inference(use_inference_params = 0) evaluates the training weights the way a training step does.
 * Master weights, Adam and the EMA are fp32 in every mode; gradient sums are 64-bit fixed point (reproducible). */
int wost_net_set_option(wost_net_handle h, const char *key, double value);
/* One training step (integrator/guided/integrator.cu:655-662: network->forward, ->backward,
 * trainer->optimizer_step(TRAIN_LOSS_SCALE)): forward with the training parameters, backward of
 * sum_p <dl_dout[p], out[p]>, then (apply_update != 0) one Adam+EMA step on gradient/loss_scale. */
int wost_net_train_step(wost_net_handle h, const float *xy, const float *dl_dout, int32_t n, float loss_scale,
                        int apply_update);

/* ---- guided integrator -------------------------------
 * Replaces GuidedIntegrator<2> as run_expr drives it (reference exec.cu:145-215): the ctor
 * (integrator/guided/integrator.cu:1148-1161), resetNetwork (:1095-1131) and solve()
 * (:1189-1195 -> solveImpl :968-1094).  The first block mirrors GuidedIntegratorSettings
 * (integrator/guided/integrator.h:54-75) plus scene.aabb of the JSON; the second block holds the
 * reference's compile-time constants, exposed so that small frames can be trained in tests. */
typedef struct wost_guided_settings {
    int32_t width, height, spp, max_depth;      /* frameSize, samplesPerPixel, maxWalkingDepth   */
    float eps_shell;                            /* epsilonShell                                  */
    int32_t train_spp_count;                    /* trainSppCount                                 */
    float uniform_fraction_training;            /* uniformFractionInTrainingPhase                */
    float uniform_fraction_guiding;             /* uniformFractionInGuidingPhase                 */
    int32_t max_guided_depth_training;          /* maxGuidedDepthInTrainingPhase                 */
    int32_t max_guided_depth_guiding;           /* maxGuidedDepthInGuidingPhase                  */
    float aabb_min[2], aabb_max[2];             /* scene.aabb (core/problem.cu loadConfig)       */
    int32_t max_train_depth;                    /* 3       integrator.h:237 (<= 4, parameters.h:7) */
    int32_t batch_size;                         /* 524288  parameters.h:11                       */
    int32_t min_batch_size;                     /* 65536   parameters.h:12                       */
    int32_t batches_per_spp;                    /* 5       integrator.h:238                      */
    int32_t train_pixel_stride;                 /* 1       guided.h:104-121                      */
    int32_t train_pixel_offset;                 /* 0; -1 = drawn per solve from the integrator's host sampler when the
                                                   stride is > 1, as the reference does (integrator.cu:126) */
    float loss_scale;                           /* 128     parameters.h:14                       */
} wost_guided_settings;

typedef struct wost_guided_stats {
    uint64_t walk_steps, walks_started, walks_absorbed, walks_truncated, neumann_hits;
    uint64_t guided_steps;       /* steps whose direction was drawn from the mixture              */
    uint64_t train_samples;      /* training records collected over all training passes           */
    uint64_t optimizer_steps;
    double solve_ms;             /* host wall time of wost_guided_solve                           */
    double train_ms;             /* part of it spent building training sets and training          */
    uint32_t kernel_launches;
    uint32_t reserved;           /* trainPixelOffset used by the solve                                */
    uint64_t net_points;         /* network evaluations made for walkers (out-of-shell entries of guided depths) */
    double net_infer_ms;         /* GPU time of the launches that evaluate the network for walkers (HIP events)  */
} wost_guided_stats;

typedef struct wost_guided *wost_guided_handle;

/* Scene upload + LBVH build + network construction (initialised from net_seed). */
int wost_guided_create(const wost_scene_desc *scene, const wost_guided_settings *settings,
                       const wost_net_config *net, uint64_t net_seed, int device, wost_guided_handle *out);
/* The integrator's network, borrowed (valid until wost_guided_destroy): get/set parameters,
 * queryNetwork-style inference (integrator.cu:566-615). */
int wost_guided_network(wost_guided_handle h, wost_net_handle *net);
/* The uploaded scene, borrowed: SDF renders and the batched geometric queries of the guided
 * integrator's scene go through the wost_* entry points above. */
int wost_guided_scene(wost_guided_handle h, wost_handle *scene);
/* queryNetwork (integrator.cu:566-615): raw[n*33] mixture parameters of the inference (EMA)
 * network at world positions pts[n*2] (normalizeSpatialCoord applied inside). */
int wost_guided_query_network(wost_guided_handle h, const float *pts, int32_t n, float *raw);
/* GuidedIntegrator<2>::solve(): all samples, training passes included; field_rgb receives
 * width*height*3 floats = solution / spp.  Starts from the network's current state. */
int wost_guided_solve(wost_guided_handle h, float *field_rgb, wost_guided_stats *stats);
/* Same solve for the 8x8-pixel tiles t with t % shard_count == shard_index (the tiling of
 * wost_solve_sharded).  Every shard trains its OWN copy of the guiding network on the records of
 * its own pixels -- the estimator is unbiased for any network state, so no collective is needed
 * on the data path.  field_rgb_dev: DEVICE buffer of width*height*3 floats; pixels of other
 * shards are written as 0, so a sum-reduce over ranks (RCCL) yields the full field.  Returns
 * after the work has completed. */
int wost_guided_solve_sharded(wost_guided_handle h, int32_t shard_index, int32_t shard_count,
                              float *field_rgb_dev, wost_guided_stats *stats);
/* Shared network across shards (optional).  By default every shard trains its own network; with
 * a sync callback the shards train ONE network: before every Adam step the callback must sum the
 * fixed-point gradient buffer over all ranks (WOST_SYNC_SUM_I64_DEVICE: data = device int64[count],
 * e.g. ncclAllReduce(.., ncclInt64, ncclSum, ..) followed by a stream synchronisation), and once per
 * training pass it must reduce the number of usable batches to the minimum over the ranks
 * (WOST_SYNC_MIN_I64_HOST: data = host int64[1]).  Integer sums make the result independent of
 * the reduction order: all ranks hold the same network bit for bit.  At the start of a solve the
 * callback is asked once for the number of ranks (WOST_SYNC_RANKS_I64_HOST: it writes host
 * int64[1]); the summed gradient is divided by it before the Adam step, so that a shared step is
 * the step of ONE batch of ranks x batch_size samples (each rank normalises its loss gradient by
 * its own batch) and the L2 term and epsilon keep their weight for any rank count.  (The op exists since library
 * version 0.2, wost_version(); a callback written against 0.1 answers it with WOST_SYNC_UNSUPPORTED and gets the summed
 * gradient undivided, with a warning on stderr.)
 * The divisor belongs to the solve: afterwards wost_net_train_step on the same network is a plain single-rank step.
 * Return 0 on success, WOST_SYNC_UNSUPPORTED for an op the callback does not know, any other value for a failure:
 * a failure of ANY op -- or a rank count below 1 -- fails the solve with WOST_ERR_DEVICE (a rank that trained on while
 * another one failed would let the "shared" weights diverge silently). */
#define WOST_SYNC_SUM_I64_DEVICE 0
#define WOST_SYNC_MIN_I64_HOST 1
#define WOST_SYNC_RANKS_I64_HOST 2
#define WOST_SYNC_UNSUPPORTED 2      /* return value of a callback for an unknown op */
typedef int (*wost_sync_fn)(void *user, int op, void *data, uint64_t count);
int wost_guided_set_sync(wost_guided_handle h, wost_sync_fn fn, void *user);
/* Intermediate frames (saveSppMetrics* / saveTimeMetrics* of GuidedIntegratorSettings, reference
 * integrator/guided/integrator.cu:1049-1081): during wost_guided_solve the callback receives
 * solution / (sample_id + 1) after sample sample_id when  spp_every > 0 && sample_id % spp_every
 * == 0 && sample_id < spp_until  (reason 0), and when  time_every > 0 && sample_id % time_every
 * == 0  (reason 1, with the milliseconds since the start of the solve).  Return 0 to continue. */
typedef int (*wost_frame_fn)(void *user, int reason, int32_t sample_id, double elapsed_ms, const float *field_rgb);
int wost_guided_set_frame_callback(wost_guided_handle h, wost_frame_fn fn, void *user, int32_t spp_every,
                                   int32_t spp_until, int32_t time_every);
/* The training set built by the most recent training pass, in (pixel, record) order
 * (generate_training_data, train.h:423-471): xy[n*2] normalised positions, dir[n*2],
 * solution[n*3] = |record.solution / record.thp|, dir_pdf[n], normal[n*2], on_neumann[n].
 * Copies min(n, capacity) entries; any output array may be NULL. */
int wost_guided_train_set(wost_guided_handle h, int32_t capacity, int32_t *n, float *xy, float *dir,
                          float *solution, float *dir_pdf, float *normal, uint8_t *on_neumann);
/* Options of a guided handle; unknown keys -> WOST_ERR_INVALID.
 * "pipeline" (0 default / 1): the PIPELINED training order.  The reference trains between the samples
 * (integrator/guided/integrator.cu:968-1094: walk of sample k, trainStep on its records, walk of sample k + 1 with the new
 * weights); with "pipeline" 1 sample k + 1 walks with a frozen copy of the weights that the training pass of sample k - 1
 * left while the pass of sample k runs on a second stream -- no barrier between walking and training.  The estimator
 * stays unbiased for any network state (direction and one-sample-MIS density of a sample come from the same copy, the
 * training records carry the density they were drawn with); the field differs from the exact order's statistically, so
 * this is never the parity mode: default off, and a solve with intermediate frames falls back to the exact order.
 * "train_group" (1 default .. 16): a training launch walks up to that many samples of every pixel, each with its own record
 * set, and their training passes follow the launch (with "pipeline" 1: on the second stream, while the next group walks): the
 * drain of a sample's longest walks is paid once per group.  The groups grow with the training -- the launch that starts at
 * sample s covers min(S, max(1, s / 2)) samples, never more than half of what has been trained before it -- so the first
 * samples, from which the network learns fastest, keep the reference's order.  The same statistical contract as "pipeline";
 * 1 = the reference's order throughout (a training pass between any two samples). */
int wost_guided_set_option(wost_guided_handle h, const char *key, double value);
int wost_guided_destroy(wost_guided_handle h);

/* The launches of the last wost_solve / wost_solve_sharded of this handle, in order (bench.py prices the dominant one against
 * the roofline; the reference has no counterpart: its solveImpl issues 2 + 5 * depth full-frame launches per sample,
 * integrator/uniform/integrator.cu:529-623).  Writes min(*count, capacity) records. */
typedef struct wost_launch_info {
    int32_t kind;             /* WOST_LAUNCH_* */
    uint32_t walkers;         /* walkers (pixels in flight) the launch started with                                      */
    uint32_t walkers_beside;  /* walkers started with it on other streams: long remainders, strayed walkers               */
    uint32_t grid;            /* workgroups                                                                              */
    double ms;                /* HIP events around it on the solve's stream (what ran beside it ends inside later spans) */
    uint64_t walk_steps_done; /* walk steps of the solve counted when the launch had ended (cumulative, all streams)      */
} wost_launch_info;
enum {
    WOST_LAUNCH_ROUND = 0,       /* walk_round_kernel: every walker up to steps_per_round steps, then compaction          */
    WOST_LAUNCH_QUAD = 1,        /* walk_quad_kernel: the same with four lanes per walker (under-filled launches)         */
    WOST_LAUNCH_ONE = 2,         /* few samples per pixel: resident lanes drain the queue, the whole solve in one launch   */
    WOST_LAUNCH_PERSISTENT = 3,  /* many samples per pixel: resident lanes take whole pixels, longest expected chain first,
                                    until none is unread; what they hold then goes to rounds                              */
    WOST_LAUNCH_WAIT = 4         /* no launch: the time the solve waited at its end for what ran beside the rounds         */
};
int wost_last_launches(wost_handle h, wost_launch_info *out, int32_t capacity, int32_t *count);

/* Tuning knobs ("steps_per_round", "block_size", "refill", "thin_waves", ...; scheduling only,
 * never the result); unknown keys -> WOST_ERR_INVALID.
 * "persist" (-1 automatic / 0 / 1): the persistent first launch of a solve with many samples per pixel and more walkers than
 * resident lanes (WOST_LAUNCH_PERSISTENT); "persist_order" 0: in queue order instead of longest-first; "long_steps" (a multiple
 * of 8, default 1024; 0 = none): pixels that launch hands over with at least that many walk steps expected still run to their
 * end beside the rounds of the others, the first "long_thin" (2048) of them four to a wave, at most "long_cap" (32768);
 * "tail_sort" (1): the first round after it takes its walkers in the order of their expected remainders; "few_order" (1): the
 * one-launch path of few samples per pixel takes the pixels longest-first too; "resident_blocks": workgroups of those launches
 * (0 = what the chip holds).
 * "spp" changes samplesPerPixel of an existing handle (a pixel's first k samples do not depend on
 * the total, so solving with spp = k reproduces the state of a longer solve after k samples: the
 * host mirror uses this for saveSppMetrics frames). */
int wost_set_option(wost_handle h, const char *key, double value);

int wost_destroy(wost_handle h);

/* ---- 3-D: UniformIntegrator<3> on triangle meshes ---------------------------------
 * Replaces Problem<3> geometry upload (core/problem.h:197-260, core/problem.cu:262-270) and
 * UniformIntegrator<3>::solve() -- the DIM == 3 branches of integrator/uniform/integrator.cu
 * (:150-168 triangle side / barycentric uv and the in-shell test, :343-365 three Neumann draws,
 * :465-525 oneStepWalk), EvaluationGrid<3> (core/evaluation_grid.h:43-70), HarmonicGreenBall<3>
 * (util/green.h:77-119), uniformSampleSphere<3> / Hemisphere<3> (util/sampling.h:20-27,57-66) and
 * frameFromNormal(Vector3f) (util/transformation.h:62-67).  colors: per vertex 6 floats, rgb on the
 * side the triangle normal (p1-p0) x (p2-p0) points to, then rgb on the other side
 * (thrust::pair first / second, integrator/common.h:250-257).  Neumann meshes of any size (flat loops up to 64
 * triangles, the LBVH with normal cones beyond), optional source term (wost3_source_desc). */
typedef struct wost3_mesh_desc {
    int32_t n_verts, n_tris;
    const float *verts;       /* n_verts * 3 */
    const int32_t *tris;      /* n_tris * 3, 0-based */
    const float *colors;      /* n_verts * 6 or NULL = zeros */
} wost3_mesh_desc;
/* Source term f of laplace(u) = -f in 3-D (Problem<3>::source_vdb_ptr, core/problem.cu:136-149; sampled at
 * integrator/uniform/integrator.cu:296-304): dense grid of RGB samples at integer index coordinates,
 * index = world * index_scale + index_offset per axis, trilinear, zero outside.  nx == 0 disables the source term. */
typedef struct wost3_source_desc {
    int32_t nx, ny, nz;
    const float *rgb;                        /* nz * ny * nx * 3, x fastest */
    float index_scale[3], index_offset[3];
    float intensity;                         /* source_intensity (core/problem.cu:179) */
} wost3_source_desc;
typedef struct wost3_scene_desc {
    wost3_mesh_desc dirichlet, neumann;      /* n_tris == 0 -> disabled */
    float dirichlet_intensity, neumann_intensity;
    float probe_scale;                       /* EvaluationGrid<3>::ProbeData: point = scale (ndc.x right + ndc.y up) + pos */
    float probe_pos[3], probe_up[3], probe_right[3];
    const uint8_t *mask;                     /* width*height bytes (0 = masked out) or NULL */
    wost3_source_desc source;                /* zero-initialised = no source term */
} wost3_scene_desc;
typedef struct wost3_context *wost3_handle;
int wost3_create(const wost3_scene_desc *scene, const wost_settings *settings, int device, wost3_handle *out);
/* UniformIntegrator<3>::solve() for the pixels [pixel_begin, pixel_end): field_rgb = (n, 3) host floats */
int wost3_solve(wost3_handle h, int32_t pixel_begin, int32_t pixel_end, float *field_rgb, wost_stats *stats);
/* the 8x8 pixel tiles t % shard_count == shard_index into a zero-filled full-frame DEVICE buffer (as wost_solve_sharded) */
int wost3_solve_sharded(wost3_handle h, int32_t shard_index, int32_t shard_count, float *field_rgb_dev, void *stream,
                        wost_stats *stats);
/* lbvh::nearest + checkPointSide + computeProjectionRatio for triangles (call sites integrator.cu:138,154-155):
 * winning triangle (lowest index on ties), distance, barycentric (u, v) of the projection, side */
int wost3_closest_point(wost3_handle h, int which_mesh, const float *pts, int32_t n, int32_t *out_idx, float *out_dist,
                        float *out_uv, int32_t *out_side);
/* lbvh::nearest_silhouette in 3-D: distance to the closest silhouette EDGE within rmax (NULL = unbounded) */
int wost3_closest_silhouette(wost3_handle h, int which_mesh, const float *pts, const float *rmax, int32_t n, float *out_dist);
/* lbvh::ray_intersect on triangles: closest hit (flag, t, triangle) */
int wost3_ray_intersect(wost3_handle h, int which_mesh, const float *origins, const float *dirs, const float *tmax, int32_t n,
                        int32_t *out_hit, float *out_t, int32_t *out_idx);
/* Problem<3>::build_bvh (core/problem.cu:31-37, 48-54: the reference builds its trees on the device).  wost3_create builds
 * every mesh with HIP kernels (csrc/wost_build3.hip); this developer / test entry builds `mesh` with those kernels AND with the
 * host builder kept as their checker, `repeat` times, and compares the two uploaded meshes byte for byte:
 * mismatch[0..13] = differing bytes of nodes, tri, triOrig, slotOfOrig, triVerts, colors, flat, flatVerts, edges, slotEdges,
 * cones, obox, areas, sampTri; [14] = differing scalars (the arrays are not compared then); [15] = bytes compared.
 * host_ms / device_ms (either may be NULL) = the fastest wall-clock time of each build, uploads and the final wait included. */
int wost3_mesh_build_check(const wost3_mesh_desc *mesh, int device, int32_t repeat, double *host_ms, double *device_ms,
                           int64_t *mismatch);
/* VMF (reference util/vmf.h:21-70), the lobe of the 3-D guided integrator's mixture (that integrator is not built; this is
 * its distribution layer, batch entry points over HOST arrays like wost_vonmises_*): eval(cosTheta) for n (kappa,
 * cos_theta) pairs; sample(sampler, mu): per point a PCG32 stream setSeed(seed[i], 1) and per_point consecutive unit
 * directions about mu[n*3] in dirs[n*per_point*3]. */
int wost3_vmf_eval(int device, const float *kappa, const float *cos_theta, int32_t n, float *pdf);
int wost3_vmf_sample(int device, const float *kappa, const float *mu, const uint64_t *seed, int32_t n, int32_t per_point,
                     float *dirs);
/* VMM<3,8> built from 40 raw network outputs per point (integrator/guided/distribution.h:279-345, train.h:50-79 with
 * common3d: 8 x (lambda, kappa, mean vector)): mixture pdf at direction wi[n*3] (pdf may be NULL) and one sampled
 * direction per point with the stream setSeed(seed[i], 1) (sample_dir may be NULL). */
int wost3_vmm_pdf_sample(int device, const float *raw, const float *wi, const uint64_t *seed, int32_t n, float *pdf,
                         float *sample_dir);
/* compute_dL_doutput_divergence with common3d::GuidedOutput (train.h:492-553) around VMM<3,N>::gradients_probability
 * (distribution.h:348-421): raw and dl_draw 41 floats per sample (the selection logit last); record dir[n*3], li, dir_pdf,
 * on_neumann (may be NULL), normal[n*3]; likelihood may be NULL. */
int wost3_vmm_loss_gradients(int device, const float *raw, const float *dir, const float *li, const float *dir_pdf,
                             const uint8_t *on_neumann, const float *normal, int32_t n, float loss_scale, float *dl_draw,
                             float *likelihood);
/* renderDirichletSDF / renderSilhouetteSDF / renderSource with DIM = 3 (integrator/common.h:52-163): one query per pixel
 * of the frame at its evaluation point; which_mesh as above; out_dist width*height floats (+inf without that mesh),
 * out_rgb width*height*3 floats (zeros without a source term) */
int wost3_render_sdf(wost3_handle h, int which_mesh, float *out_dist);
int wost3_render_source(wost3_handle h, float *out_rgb);
int wost3_destroy(wost3_handle h);

/* ---- 3-D: GuidedIntegrator<3> (SURVEY.md 8a rows a21-a27 with DIM == 3) -----------------------------------------------
 * Replaces the GuidedIntegrator<3> alternative of run_expr's variant (exec.cu:102-122) and its solve() -- the DIM == 3
 * branches of integrator/guided/integrator.cu (:181-215 triangle side / barycentric uv, :347 three Neumann draws, :508-511,
 * :690-693, :800-803 VMM<3,8> and the reflection about the Neumann normal), guided/parameters.h:26-33 (3 network inputs,
 * 8 x (lambda, kappa, mean vector) + selection logit = 41 outputs padded to 48), train.h:289-353 (3-D records) -- on the
 * scene types of the 3-D uniform integrator above.  The network is the one of wost3_net_create: the DenseGrid encoding
 * with three inputs (trilinear, res^3 entries per level), otherwise the configuration of data/ladybug/n.json:49-81; fp32.
 * Scenes with a source term are solved like the others (sampleSourceImpl with DIM == 3, integrator.cu:277-364; the
 * dense-grid stand-in for nanovdb of wost3_source_desc): tests/test_guided_3d.py::test_gpu_guided3_source_term_matches_oracle. */
int wost3_net_create(int device, const wost_net_config *cfg, uint64_t seed, wost_net_handle *out);   /* inputs: 3 floats per point */
typedef struct wost3_guided_settings {
    int32_t width, height, spp, max_depth;
    float eps_shell;
    int32_t train_spp_count;
    float uniform_fraction_training, uniform_fraction_guiding;
    int32_t max_guided_depth_training, max_guided_depth_guiding;
    float aabb_min[3], aabb_max[3];             /* scene.aabb */
    int32_t max_train_depth, batch_size, min_batch_size, batches_per_spp, train_pixel_stride, train_pixel_offset;
    float loss_scale;                           /* meanings and reference defaults as in wost_guided_settings */
} wost3_guided_settings;
typedef struct wost3_guided *wost3_guided_handle;
/* integrator/guided/integrator.h:127,175 (ctor + resetNetwork) with DIM == 3; net->n_output must be 41 */
int wost3_guided_create(const wost3_scene_desc *scene, const wost3_guided_settings *settings, const wost_net_config *net,
                        uint64_t net_seed, int device, wost3_guided_handle *out);
int wost3_guided_destroy(wost3_guided_handle h);
int wost3_guided_network(wost3_guided_handle h, wost_net_handle *net);            /* borrowed: owned by the integrator */
int wost3_guided_scene(wost3_guided_handle h, wost3_handle *scene);               /* borrowed: for the SDF / source channels */
/* solve() (integrator.cu:1189-1195): field_rgb = width*height*3 floats on the host / in device memory (tile shard as in
 * wost3_solve_sharded, every shard its own network) */
int wost3_guided_solve(wost3_guided_handle h, float *field_rgb, wost_guided_stats *stats);
int wost3_guided_solve_sharded(wost3_guided_handle h, int32_t shard_index, int32_t shard_count, float *field_rgb_dev,
                               wost_guided_stats *stats);
/* queryNetwork(Vector3f) (exec.cu:175-186): raw = n * 41 mixture parameters of the inference weights at pts (n * 3, world) */
int wost3_guided_query_network(wost3_guided_handle h, const float *pts, int32_t n, float *raw);
/* the ordered training set of the last training pass (tests): xyz = normalised inputs, dir / normal 3 floats per sample */
int wost3_guided_train_set(wost3_guided_handle h, int32_t capacity, int32_t *n, float *xyz, float *dir, float *solution,
                           float *dir_pdf, float *normal, uint8_t *on_neumann);

const char *wost_last_error(void);
const char *wost_version(void);
/* Problem<2>::build_bvh (core/problem.cu:31-37, 48-54: the reference builds its trees on the device).  wost_create builds every
 * mesh of 512 segments or more with HIP kernels (csrc/wost_build2.hip); this developer / test entry builds `mesh` with those kernels
 * AND with the host builder kept as their checker (csrc/lbvh_build.cpp), `repeat` times, and compares the two uploaded trees byte
 * for byte: mismatch[0..13] = differing bytes of nodes, cones, segA, segInv, segOrig, segCol, segVerts, flat, flatCol, sil, silN,
 * scanBox, scanHl, scanId; [14] = differing scalars (the arrays are not compared then); [15] = bytes compared.
 * host_ms / device_ms (either may be NULL) = the fastest wall-clock time of each build, uploads and the final wait included. */
int wost_mesh_build_check(const wost_mesh_desc *mesh, int device, int32_t repeat, double *host_ms, double *device_ms, int64_t *mismatch);

#ifdef __cplusplus
}
#endif
#endif
