/*
 * wost_guided.c -- CPU oracle of the GUIDED Walk-on-Stars integrator (SURVEY.md 8a rows a21,
 * a22, a25, a26, a27), built on the uniform oracle's geometric queries (wost_oracle.c), the von
 * Mises mixture (wost_vmm.c) and the guiding network (wost_net.c).  TEST INFRASTRUCTURE ONLY.
 *
 * What it restates (paths relative to /root/reference):
 *   integrator/guided/integrator.cu:112-128 (prepareSolve), :131-150 (generateEvaluationPoints),
 *     :153-249 (separateEvaluationPoint, R_B without the 0.99 factor :238-239), :252-274
 *     (handleBoundary + recordSolution), :367-494 (sampleNeumann + recordSourceContribution),
 *     :497-526 (handleOutShellPoint routing), :529-563 (inferenceStep), :618-668 (trainStep),
 *     :671-779 (handleUniformSampling), :782-880 (handleGuidedSampling), :883-965 (oneStepWalk),
 *     :968-1094 (solveImpl, phase switch :991-996), ctor :1158-1160
 *   integrator/guided/guided.h:12-69 (records), :104-121 (TrainState), guideditem.h:21-36
 *   integrator/guided/train.h:149-155 (normalizeSpatialCoord), :423-471 (generate_training_data),
 *     :474-486 (generate_inference_data), :492-553 (loss gradient, in wost_vmm.c)
 *   integrator/guided/parameters.h:7-16, integrator.h:232-239 (batch constants)
 *
 * The reference is not bit-reproducible here (its training set is ordered by atomics, its
 * network runs in half precision, and tiny-cuda-nn is absent: PARITY UNPINNED, see wost_net.c).
 * This restatement fixes the free choices so that it IS deterministic: the training set is
 * ordered by (pixel id, record index), the network is the fp32 one of wost_net.c.  The solve is
 * synchronous per sample and per depth like the reference's wavefront, because every pixel
 * shares one network that changes between samples.
 *
 * Reference behaviours kept on purpose:
 *   - the routing draw happens before the AABB test and only when uniform fraction != 0 (:518);
 *   - uniform fraction >= 1 never launches the guided kernel, so walks routed to it end (:1031);
 *   - a record's solution slot [curDepth] collects Neumann contributions and is wiped when the
 *     record is created (guided.h:59-68 against :33).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "wost_internal.h"
#include "wost_detmath.h"

typedef struct {
    float sol[3];
    float px, py, dx, dy, pdf, thp;
    int on_n;
    float nx, ny;
} g_record;

#define G_MAX_TRAIN_DEPTH 4   /* parameters.h:7 */

typedef struct {
    wo_pcg rng;
    float sol[3];
    /* walker of the current sample */
    int state;            /* 0 = none, 1 = evaluation point queued, 2 = out of shell (has R_B) */
    float x, y, thp;
    int on_n;
    float nx, ny;
    float R_B;
    /* training records of the current sample */
    g_record rec[G_MAX_TRAIN_DEPTH + 1];
    unsigned cur_depth;
} g_pixel;

typedef struct {
    float min[2], max[2];
} g_aabb;

static int aabb_contains(const g_aabb *b, float x, float y)
{
    return b->min[0] <= x && x <= b->max[0] && b->min[1] <= y && y <= b->max[1];
}

/* train.h:149-155: inflate by 0.5 % of the diagonal length, then 0.5 + (p - centre) / extent */
static void normalize_coord(const g_aabb *b, float x, float y, float *ox, float *oy)
{
    const float ex = b->max[0] - b->min[0], ey = b->max[1] - b->min[1];
    const float infl = sqrtf(ex * ex + ey * ey) * 0.005f;
    const float lox = b->min[0] - infl, loy = b->min[1] - infl, hix = b->max[0] + infl, hiy = b->max[1] + infl;
    const float cx = (lox + hix) / 2.0f, cy = (loy + hiy) / 2.0f;      /* Eigen AlignedBox::center */
    *ox = 0.5f + (x - cx) / (hix - lox);
    *oy = 0.5f + (y - cy) / (hiy - loy);
}

static void record_solution(g_pixel *p, const float c[3], int inclusive)
{
    unsigned depth = p->cur_depth < G_MAX_TRAIN_DEPTH ? p->cur_depth : G_MAX_TRAIN_DEPTH;
    unsigned end = inclusive ? depth + 1 : depth;      /* guided.h:48-57 against :59-68 */
    for (unsigned i = 0; i < end; ++i)
        for (int ch = 0; ch < 3; ++ch) p->rec[i].sol[ch] = p->rec[i].sol[ch] + c[ch];
}

static void increment_depth(g_pixel *p, float dx, float dy, float pdf)
{
    unsigned d = p->cur_depth;
    if (d >= G_MAX_TRAIN_DEPTH) return;
    g_record *r = &p->rec[d];
    r->sol[0] = r->sol[1] = r->sol[2] = 0.0f;
    r->px = p->x; r->py = p->y; r->dx = dx; r->dy = dy; r->pdf = pdf; r->thp = p->thp;
    r->on_n = p->on_n; r->nx = p->nx; r->ny = p->ny;
    p->cur_depth = d + 1;
}

/* the shared tail of the three sampling kernels: intersect, advance, record */
static void advance_walker(const pmesh *nm, g_pixel *p, float eps, float dirx, float diry, float pdf, float alpha,
                           int record, uint64_t *nhits)
{
    float cxp = p->x, cyp = p->y;
    if (p->on_n) { cxp += eps * p->nx; cyp += eps * p->ny; }
    float nxt_x = p->x + p->R_B * dirx, nxt_y = p->y + p->R_B * diry;
    int hit = 0;
    float hnx = 0.0f, hny = 0.0f;
    if (nm->n_segs > 0) {
        float t; int hi;
        hit = ray_closest(nm, cxp, cyp, dirx, diry, p->R_B, &t, &hi);
        if (hit) {
            hnx = nm->segs[hi].nx; hny = nm->segs[hi].ny;
            if (wo_dot2(hnx, hny, dirx, diry) > 0) { hnx = -hnx; hny = -hny; }
            nxt_x = cxp + t * dirx; nxt_y = cyp + t * diry;
            if (nhits) __atomic_fetch_add(nhits, 1, __ATOMIC_RELAXED);
        }
    }
    if (record) increment_depth(p, dirx, diry, pdf);      /* records the state BEFORE the step */
    p->thp = p->thp / pdf / alpha / WO_2PI;
    p->x = nxt_x; p->y = nxt_y;
    p->on_n = hit; p->nx = hnx; p->ny = hny;
    p->state = 1;
}

static void uniform_direction(g_pixel *p, float *dirx, float *diry, float *pdf, float *alpha)
{
    float u = wo_pcg_next_float(&p->rng);
    if (p->on_n) {
        float lc, ls;
        wo_sincos_2pi(u * 0.5f, &lc, &ls);
        float qx = -p->ny, qy = p->nx;
        float ql = sqrtf(wo_dot2(qx, qy, qx, qy));
        float tx = -(qx / ql), ty = -(qy / ql);
        *dirx = tx * lc + p->nx * ls;
        *diry = ty * lc + p->ny * ls;
        *pdf = (float)(1.0 / WO_PI_D);
        *alpha = 0.5f;
    } else {
        wo_sincos_2pi(u, dirx, diry);
        *pdf = 1.0f / WO_2PI;
        *alpha = 1.0f;
    }
}

int wo_solve_guided(const wo_scene *sc, const wo_guided_settings *gs, const wo_net_config *nc, float *params,
                    int n_threads, float *field_rgb, wo_guided_stats *stats, int dump_spp, wo_train_dump *dump)
{
    if (!sc || !gs || !nc || !params || !field_rgb) return -1;
    const int W = gs->width, H = gs->height, N = W * H;
    pmesh dm, nm;
    if (pmesh_prepare(&dm, &sc->dirichlet)) return -2;
    if (pmesh_prepare(&nm, &sc->neumann)) { pmesh_free(&dm); return -2; }
    const int has_d = dm.n_segs > 0, has_n = nm.n_segs > 0;
    const float eps = gs->eps_shell;
    g_aabb box = { { gs->aabb_min[0], gs->aabb_min[1] }, { gs->aabb_max[0], gs->aabb_max[1] } };
    const uint64_t n_params = wo_net_n_params(nc);
    const int NO = nc->n_output_padded;

    g_pixel *px = calloc((size_t)N, sizeof(g_pixel));
    float *inf_params = malloc(sizeof(float) * n_params);
    float *m1 = calloc(n_params, sizeof(float)), *m2 = calloc(n_params, sizeof(float)), *ema = calloc(n_params, sizeof(float));
    uint32_t *param_steps = calloc(n_params, sizeof(uint32_t));
    float *grad = malloc(sizeof(float) * n_params);
    float *net_in = malloc(sizeof(float) * 2 * (size_t)N);
    float *net_out = malloc(sizeof(float) * (size_t)NO * N);
    int *slot_of = malloc(sizeof(int) * (size_t)N);
    /* training set: at most max_train_depth records per pixel */
    const size_t max_samples = (size_t)N * G_MAX_TRAIN_DEPTH;
    float *t_xy = malloc(sizeof(float) * 2 * max_samples), *t_dir = malloc(sizeof(float) * 2 * max_samples);
    float *t_li = malloc(sizeof(float) * max_samples), *t_pdf = malloc(sizeof(float) * max_samples);
    float *t_nrm = malloc(sizeof(float) * 2 * max_samples), *t_sol = malloc(sizeof(float) * 3 * max_samples);
    unsigned char *t_onn = malloc(max_samples);
    float *t_out = NULL, *t_dl = NULL;
    memcpy(inf_params, params, sizeof(float) * n_params);
    int opt_step = 0;
    uint64_t steps = 0, nhits = 0, guided_steps = 0, train_samples = 0, absorbed = 0, truncated = 0, started = 0;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#else
    (void)n_threads;
#endif

    /* prepareSolve (integrator.cu:112-128) */
    for (int p = 0; p < N; ++p) wo_pcg_seed_pixel(&px[p].rng, p, W);

    int training = 1;
    float uniform_fraction = gs->uniform_fraction_training;
    int max_guided_depth = gs->max_guided_depth_training;

    for (int sample = 0; sample < gs->spp; ++sample) {
        if (sample == gs->train_spp_count) {            /* integrator.cu:991-996 */
            training = 0;
            uniform_fraction = gs->uniform_fraction_guiding;
            max_guided_depth = gs->max_guided_depth_guiding;
        }
        for (int p = 0; p < N; ++p) {
            g_pixel *q = &px[p];
            q->cur_depth = 0;
            q->state = 0;
            if (sc->mask && sc->mask[p] == 0) continue;
            wo_eval_point(sc, p % W, p / W, W, H, &q->x, &q->y);
            q->thp = 1.0f; q->on_n = 0; q->nx = 0.0f; q->ny = 0.0f;
            q->state = 1;
            started++;
        }
        for (int depth = 0; depth < gs->max_depth; ++depth) {
            const int guiding = depth < max_guided_depth;      /* enableGuiding is always on (:126) */
            /* ---- separate + handleBoundary + sampleNeumann ---- */
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads) reduction(+ : steps, absorbed)
            for (int p = 0; p < N; ++p) {
                g_pixel *q = &px[p];
                if (q->state != 1) continue;
                steps++;
                const int train_px = training && ((unsigned)(p - gs->train_pixel_offset) % (unsigned)gs->train_pixel_stride == 0);
                float R_D = INFINITY;
                if (has_d) {
                    cp_result cp = closest_bvh(&dm, q->x, q->y);
                    const pseg *s = &dm.segs[cp.idx];
                    int side = seg_side(s, q->x, q->y);
                    float uv = seg_proj_ratio(s, q->x, q->y);
                    R_D = sqrtf(cp.d2);
                    if ((R_D < eps) && (uv > 0.0f && uv < 1.0f)) {
                        float col[3];
                        surface_color(dm.colors, s->i0, s->i1, side, uv, col);
                        for (int c = 0; c < 3; ++c) {
                            col[c] *= sc->dirichlet_intensity;
                            col[c] *= q->thp;
                            q->sol[c] = col[c] + q->sol[c];
                        }
                        if (train_px) record_solution(q, col, 0);
                        q->state = 0;
                        absorbed++;
                        continue;
                    }
                }
                float R_N = INFINITY;
                if (has_n) R_N = closest_silhouette(&nm, q->x, q->y, R_D);
                float R_B = fmaxf(WO_R_B_FLOOR, fminf(R_D, R_N));      /* no 0.99 here (:238-239) */
                if (isinf(R_B)) { q->state = 0; continue; }            /* no boundary at all: nothing to walk to */
                q->R_B = R_B;
                q->state = 2;
                if (sc->source.nx > 0) {          /* sampleSource (guided/integrator.cu:277-364) */
                    float col[3];
                    if (wo_sample_source(&sc->source, &nm, eps, q->x, q->y, R_B, q->on_n, q->nx, q->ny, q->thp, &q->rng, col)) {
                        for (int c = 0; c < 3; ++c) q->sol[c] = col[c] + q->sol[c];
                        if (train_px) record_solution(q, col, 1);
                    }
                }
                if (has_n) {
                    float u0 = wo_pcg_next_float(&q->rng);
                    float u1 = wo_pcg_next_float(&q->rng);
                    float pdf;
                    int oi = sample_in_sphere(&nm, q->x, q->y, R_B, u0, &pdf);
                    if (oi != -1 && pdf > 0) {
                        const pseg *so = &nm.segs[oi];
                        float spx = fmaf(u1, so->ex, so->ax), spy = fmaf(u1, so->ey, so->ay);
                        float rx = spx - q->x, ry = spy - q->y;
                        float r = sqrtf(wo_dot2(rx, ry, rx, ry));
                        if (r < R_B && r > 0) {
                            float ox = q->x, oy = q->y;
                            if (q->on_n) { ox += eps * q->nx; oy += eps * q->ny; }
                            float dx = spx - ox, dy = spy - oy;
                            float cd = sqrtf(wo_dot2(dx, dy, dx, dy));
                            if (cd > 0) { dx /= cd; dy /= cd; }
                            if (!ray_any(&nm, ox, oy, dx, dy, cd - eps)) {
                                int side = seg_side(so, q->x, q->y);
                                float uv = seg_proj_ratio(so, spx, spy);
                                if (q->on_n) {
                                    float dn = wo_dot2(so->nx, so->ny, q->nx, q->ny);
                                    side = (0.0f < dn) - (dn < 0.0f);
                                }
                                if (side != 0) {
                                    float col[3];
                                    surface_color(nm.colors, so->i0, so->i1, side, uv, col);
                                    float alpha = q->on_n ? 0.5f : 1.0f;
                                    float G = wo_logf(R_B / r) / WO_2PI;
                                    for (int c = 0; c < 3; ++c) {
                                        col[c] *= sc->neumann_intensity;
                                        col[c] *= q->thp * G / alpha / pdf;
                                        col[c] = -col[c];
                                        q->sol[c] = col[c] + q->sol[c];
                                    }
                                    if (train_px) record_solution(q, col, 1);
                                }
                            }
                        }
                    }
                }
            }
            /* ---- inferenceStep: batch of the out-of-shell points, EMA weights ---- */
            int n_out = 0;
            for (int p = 0; p < N; ++p) {
                slot_of[p] = -1;
                if (px[p].state == 2) {
                    slot_of[p] = n_out;
                    normalize_coord(&box, px[p].x, px[p].y, &net_in[2 * n_out], &net_in[2 * n_out + 1]);
                    n_out++;
                }
            }
            if (n_out == 0) break;
            if (guiding) {
                const int chunk = 256;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
                for (int b = 0; b < n_out; b += chunk) {
                    int cnt = n_out - b < chunk ? n_out - b : chunk;
                    wo_net_forward(nc, inf_params, net_in + 2 * (size_t)b, cnt, net_out + (size_t)NO * b, NULL);
                }
            }
            /* ---- handleOutShellPoint + guided / uniform sampling, or oneStepWalk ---- */
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads) reduction(+ : guided_steps)
            for (int p = 0; p < N; ++p) {
                g_pixel *q = &px[p];
                if (q->state != 2) continue;
                const int train_px = training && ((unsigned)(p - gs->train_pixel_offset) % (unsigned)gs->train_pixel_stride == 0);
                const int record = train_px && depth < gs->max_train_depth;
                float dirx, diry, pdf, alpha;
                if (!guiding) {                                          /* oneStepWalk (:883-965) */
                    uniform_direction(q, &dirx, &diry, &pdf, &alpha);
                    advance_walker(&nm, q, eps, dirx, diry, pdf, alpha, record, &nhits);
                    continue;
                }
                const float *raw = net_out + (size_t)NO * slot_of[p];
                const float sel = 1 / (1.f + wo_expf(-raw[32]));            /* logistic, functors.h:182 */
                const int inside = aabb_contains(&box, q->x, q->y);
                int to_guided = (uniform_fraction == 0) || (wo_pcg_next_float(&q->rng) < sel);
                to_guided = to_guided && inside;
                if (to_guided) {
                    if (!(uniform_fraction < 1.0f)) { q->state = 0; continue; }   /* kernel not launched (:1031) */
                    wv_vmm m;
                    wo_vmm_build(&m, raw);
                    wo_vmm_sample(&m, &q->rng, &dirx, &diry);
                    float guided_pdf = wo_vmm_pdf(&m, dirx, diry);
                    float uniform_pdf = 1.0f / WO_2PI;
                    alpha = 1.0f;
                    if (q->on_n) {
                        uniform_pdf = (float)(1.0 / WO_PI_D);
                        alpha = 0.5f;
                        const float dd = 2 * (dirx * q->nx + diry * q->ny);
                        const float rx = dirx - dd * q->nx, ry = diry - dd * q->ny;
                        if (q->nx * dirx + q->ny * diry <= 0) { dirx = rx; diry = ry; }
                        guided_pdf += wo_vmm_pdf(&m, rx, ry);
                    }
                    pdf = sel * guided_pdf + (1.0f - sel) * uniform_pdf;
                    guided_steps++;
                } else {                                                 /* handleUniformSampling (:671-779) */
                    uniform_direction(q, &dirx, &diry, &pdf, &alpha);
                    if (inside) {
                        wv_vmm m;
                        wo_vmm_build(&m, raw);
                        float guided_pdf = wo_vmm_pdf(&m, dirx, diry);
                        if (q->on_n) {
                            const float dd = 2 * (dirx * q->nx + diry * q->ny);
                            guided_pdf += wo_vmm_pdf(&m, dirx - dd * q->nx, diry - dd * q->ny);
                        }
                        pdf = sel * guided_pdf + (1.0f - sel) * pdf;
                    }
                }
                advance_walker(&nm, q, eps, dirx, diry, pdf, alpha, record, &nhits);
            }
            if (depth == gs->max_depth - 1)
                for (int p = 0; p < N; ++p) truncated += px[p].state == 1;
        }

        /* ---- trainStep (:618-668) ---- */
        if (training) {
            size_t n = 0;
            for (unsigned p = (unsigned)gs->train_pixel_offset; p < (unsigned)N; p += (unsigned)gs->train_pixel_stride) {
                const g_pixel *q = &px[p];
                for (unsigned k = 0; k < q->cur_depth; ++k) {
                    const g_record *r = &q->rec[k];
                    if (!aabb_contains(&box, r->px, r->py)) continue;
                    float s3[3];
                    for (int ch = 0; ch < 3; ++ch) {
                        float v = 0.0f;
                        if (fabsf(r->thp) > 1e-5f) v = r->sol[ch] / r->thp;      /* M_EPSILON */
                        s3[ch] = fabsf(v);
                    }
                    float ix, iy;
                    normalize_coord(&box, r->px, r->py, &ix, &iy);
                    if (isnan(ix) || isnan(iy) || isnan(r->dx) || isnan(r->dy) || isnan(r->pdf) || r->pdf == 0 ||
                        isnan(s3[0]) || isnan(s3[1]) || isnan(s3[2]))
                        continue;
                    t_xy[2 * n] = ix; t_xy[2 * n + 1] = iy;
                    t_dir[2 * n] = r->dx; t_dir[2 * n + 1] = r->dy;
                    t_sol[3 * n] = s3[0]; t_sol[3 * n + 1] = s3[1]; t_sol[3 * n + 2] = s3[2];
                    t_li[n] = (s3[0] + s3[1] + s3[2]) / 3.0f;            /* Color::mean() */
                    t_pdf[n] = r->pdf;
                    t_onn[n] = (unsigned char)r->on_n;
                    t_nrm[2 * n] = r->nx; t_nrm[2 * n + 1] = r->ny;
                    n++;
                }
            }
            train_samples += n;
            if (dump && sample == dump_spp) {
                dump->n = (int)n;
                size_t m = n < (size_t)dump->capacity ? n : (size_t)dump->capacity;
                if (dump->xy) memcpy(dump->xy, t_xy, sizeof(float) * 2 * m);
                if (dump->dir) memcpy(dump->dir, t_dir, sizeof(float) * 2 * m);
                if (dump->solution) memcpy(dump->solution, t_sol, sizeof(float) * 3 * m);
                if (dump->dir_pdf) memcpy(dump->dir_pdf, t_pdf, sizeof(float) * m);
                if (dump->on_neumann) memcpy(dump->on_neumann, t_onn, m);
                if (dump->normal) memcpy(dump->normal, t_nrm, sizeof(float) * 2 * m);
            }
            const size_t bs = (size_t)gs->batch_size;
            size_t n_batches = n / bs + 1;
            if (n_batches > (size_t)gs->batches_per_spp) n_batches = (size_t)gs->batches_per_spp;
            for (size_t it = 0; it < n_batches; ++it) {
                size_t local = n - it * bs < bs ? n - it * bs : bs;
                local -= local % 128;
                if (local < (size_t)gs->min_batch_size) break;
                const size_t o = it * bs;
                t_out = realloc(t_out, sizeof(float) * (size_t)NO * local);
                t_dl = realloc(t_dl, sizeof(float) * (size_t)NO * local);
                const int chunk = 256;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
                for (size_t b = 0; b < local; b += chunk) {
                    int cnt = local - b < (size_t)chunk ? (int)(local - b) : chunk;
                    wo_net_forward(nc, params, t_xy + 2 * (o + b), cnt, t_out + (size_t)NO * b, NULL);
                }
                /* loss gradient on the first 33 outputs of every sample */
                float *raw33 = malloc(sizeof(float) * 33 * local), *dl33 = malloc(sizeof(float) * 33 * local);
                for (size_t i = 0; i < local; ++i) memcpy(raw33 + 33 * i, t_out + (size_t)NO * i, sizeof(float) * 33);
                wo_vmm_loss_gradients(raw33, t_dir + 2 * o, t_li + o, t_pdf + o, t_onn + o, t_nrm + 2 * o, (int)local,
                                      gs->loss_scale, dl33, NULL);
                memset(t_dl, 0, sizeof(float) * (size_t)NO * local);
                for (size_t i = 0; i < local; ++i) memcpy(t_dl + (size_t)NO * i, dl33 + 33 * i, sizeof(float) * 33);
                free(raw33); free(dl33);
                wo_net_backward(nc, params, t_xy + 2 * o, t_dl, (int)local, grad);
                opt_step++;
                wo_net_optimizer_step(nc, params, m1, m2, ema, inf_params, grad, opt_step, gs->loss_scale, param_steps);
            }
        }
    }
    for (int p = 0; p < N; ++p)
        for (int c = 0; c < 3; ++c) field_rgb[3 * (size_t)p + c] = px[p].sol[c] / (float)gs->spp;
    if (stats) {
        stats->walk_steps = steps; stats->walks_started = started; stats->walks_absorbed = absorbed;
        stats->walks_truncated = truncated; stats->neumann_hits = nhits; stats->guided_steps = guided_steps;
        stats->train_samples = train_samples; stats->optimizer_steps = (uint64_t)opt_step;
    }
    free(px); free(inf_params); free(m1); free(m2); free(ema); free(param_steps); free(grad); free(net_in); free(net_out); free(slot_of);
    free(t_xy); free(t_dir); free(t_li); free(t_pdf); free(t_nrm); free(t_sol); free(t_onn); free(t_out); free(t_dl);
    pmesh_free(&dm); pmesh_free(&nm);
    return 0;
}
