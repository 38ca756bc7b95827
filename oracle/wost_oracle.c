/*
 * wost_oracle.c -- CPU oracle (plain C11 + pthreads) for the Walk-on-Stars
 * wavefront hot path of the uniform integrator.  See wost_oracle.h for scope,
 * the list of reference lines restated and the parity status.
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into the product.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fPIC -shared \
 *            wost_oracle.c -o libwost_oracle.so -lm -lpthread
 */
#define _GNU_SOURCE
#include "wost_oracle.h"
#include "wost_internal.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
/* PCG32: core/sampler.h:10-103                                              */
/* ------------------------------------------------------------------------ */
#define WO_PCG32_MULT 0x5851f42d4c957f2dULL

uint32_t wo_pcg_next_uint(wo_pcg *r)
{
    /* core/sampler.h:65-72 */
    uint64_t oldstate = r->state;
    r->state = oldstate * WO_PCG32_MULT + r->inc;
    uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

void wo_pcg_set_seed(wo_pcg *r, uint64_t initstate, uint64_t initseq)
{
    /* core/sampler.h:20-27 */
    r->state = 0U;
    r->inc = (initseq << 1u) | 1u;
    wo_pcg_next_uint(r);
    r->state += initstate;
    wo_pcg_next_uint(r);
}

float wo_pcg_next_float(wo_pcg *r)
{
    /* core/sampler.h:87-98 */
    union { uint32_t u; float f; } x;
    x.u = (wo_pcg_next_uint(r) >> 9) | 0x3f800000u;
    return x.f - 1.0f;
}

double wo_pcg_next_double(wo_pcg *r)
{
    /* core/sampler.h:74-85 */
    union { uint64_t u; double d; } x;
    x.u = ((uint64_t)wo_pcg_next_uint(r) << 20) | 0x3ff0000000000000ULL;
    return x.d - 1.0;
}

void wo_pcg_advance(wo_pcg *r, int64_t delta)
{
    /* core/sampler.h:46-62 */
    uint64_t cur_mult = WO_PCG32_MULT, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta /= 2;
    }
    r->state = acc_mult * r->state + acc_plus;
}

uint32_t wo_interleave_32bit(uint32_t vx, uint32_t vy)
{
    /* util/hash.h:13-28 */
    uint32_t x = vx & 0x0000ffff;
    uint32_t y = vy & 0x0000ffff;
    x = (x | (x << 8)) & 0x00FF00FF;
    x = (x | (x << 4)) & 0x0F0F0F0F;
    x = (x | (x << 2)) & 0x33333333;
    x = (x | (x << 1)) & 0x55555555;
    y = (y | (y << 8)) & 0x00FF00FF;
    y = (y | (y << 4)) & 0x0F0F0F0F;
    y = (y | (y << 2)) & 0x33333333;
    y = (y | (y << 1)) & 0x55555555;
    return x | (y << 1);
}

void wo_pcg_seed_pixel(wo_pcg *r, int pixel_id, int width)
{
    /* integrator.cu:71-77: pixelCoord = {id % W, id / W};
     * setPixelSample(coord, 0) -> setSeed(interleave(coord), 0) (sampler.h:29-34);
     * advance(256 * pixelId) -- the product is an int in the reference. */
    uint32_t px = (uint32_t)(pixel_id % width);
    uint32_t py = (uint32_t)(pixel_id / width);
    wo_pcg_set_seed(r, (uint64_t)wo_interleave_32bit(px, py), 0);
    int delta = 256 * pixel_id;
    wo_pcg_advance(r, (int64_t)delta);
}

/* ------------------------------------------------------------------------ */
/* deterministic math (specified in DESIGN.md, "deterministic math")         */
/* ------------------------------------------------------------------------ */
void wo_sincos_2pi(float u, float *c, float *s)
{
#ifdef WOST_ORACLE_LIBM
    /* literal: util/sampling.h:29-33 */
    const float theta = u * WO_2PI;
    *c = cosf(theta);
    *s = sinf(theta);
#else
    /* octant reduction is exact for u = k * 2^-23 */
    float r = u * 8.0f;
    int j = (int)r;
    float f = r - (float)j;
    float g = (j & 1) ? (1.0f - f) : f;
    float z = g * g;
    /* sin(pi/4 * g), Taylor coefficients (pi/4)^(2k+1)/(2k+1)! rounded to fp32 */
    float ps = 0x1.507834p-22f;
    ps = fmaf(ps, z, -0x1.32d2ccp-15f);
    ps = fmaf(ps, z, 0x1.466bc6p-9f);
    ps = fmaf(ps, z, -0x1.4abbcep-4f);
    ps = fmaf(ps, z, 0x1.921fb6p-1f);
    float sg = ps * g;
    /* cos(pi/4 * g) */
    float pc = -0x1.a6d1f2p-26f;
    pc = fmaf(pc, z, 0x1.e1f506p-19f);
    pc = fmaf(pc, z, -0x1.55d3c8p-12f);
    pc = fmaf(pc, z, 0x1.03c1f0p-6f);
    pc = fmaf(pc, z, -0x1.3bd3ccp-2f);
    pc = fmaf(pc, z, 1.0f);
    float cg = pc;
    /* octant j: angle = (j + f) * pi/4 */
    float cc, ss;
    switch (j & 7) {
    case 0: cc = cg;  ss = sg;  break;
    case 1: cc = sg;  ss = cg;  break;
    case 2: cc = -sg; ss = cg;  break;
    case 3: cc = -cg; ss = sg;  break;
    case 4: cc = -cg; ss = -sg; break;
    case 5: cc = -sg; ss = -cg; break;
    case 6: cc = sg;  ss = -cg; break;
    default: cc = cg; ss = -sg; break;
    }
    *c = cc;
    *s = ss;
#endif
}

float wo_logf(float x)
{
#ifdef WOST_ORACLE_LIBM
    return logf(x);
#else
    union { float f; uint32_t u; } v;
    v.f = x;
    int e = 0;
    if (v.u < 0x00800000u) { /* subnormal */
        v.f = x * 8388608.0f;
        e = -23;
    }
    e += (int)((v.u >> 23) & 0xff) - 127;
    v.u = (v.u & 0x007fffffu) | 0x3f800000u;
    float m = v.f;
    if (m > 0x1.6a09e6p+0f) { /* sqrt(2) */
        m = m * 0.5f;
        e += 1;
    }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = 0x1.c71c72p-3f;            /* 2/9 */
    p = fmaf(p, z, 0x1.24924ap-2f);      /* 2/7 */
    p = fmaf(p, z, 0x1.99999ap-2f);      /* 2/5 */
    p = fmaf(p, z, 0x1.555556p-1f);      /* 2/3 */
    p = p * z;
    float r = fmaf(s, p, s + s);
    return fmaf((float)e, 0x1.62e430p-1f, r);
#endif
}

/* ------------------------------------------------------------------------ */
/* evaluation grid: core/evaluation_grid.h:27-33                             */
/* ------------------------------------------------------------------------ */
void wo_eval_point(const wo_scene *sc, int px, int py, int width, int height,
                   float *x, float *y)
{
    float ndcx = 2.0f * (float)px / (float)width + -1.0f;
    float ndcy = 2.0f * (float)py / (float)height + -1.0f;
    float ux = sc->probe_up[1], uy = -sc->probe_up[0];
    float vx = sc->probe_up[0], vy = sc->probe_up[1];
    *x = sc->probe_scale * (ndcx * ux + ndcy * vx) + sc->probe_pos[0];
    *y = sc->probe_scale * (ndcx * uy + ndcy * vy) + sc->probe_pos[1];
}

/* ------------------------------------------------------------------------ */
/* prepared mesh                                                             */
/* ------------------------------------------------------------------------ */
void pmesh_free(pmesh *m)
{
    free(m->segs); free(m->v_prev); free(m->v_next); free(m->nodes); free(m->order);
    memset(m, 0, sizeof(*m));
}

static void seg_prepare(pseg *s, const float *verts, int i0, int i1)
{
    s->i0 = i0; s->i1 = i1;
    s->ax = verts[2 * i0]; s->ay = verts[2 * i0 + 1];
    float bx = verts[2 * i1], by = verts[2 * i1 + 1];
    s->ex = bx - s->ax; s->ey = by - s->ay;
    float len2 = wo_dot2(s->ex, s->ey, s->ex, s->ey);
    s->inv_len2 = (len2 > 0.0f) ? 1.0f / len2 : 0.0f;
    s->len = sqrtf(len2);
    /* unit normal (e.y, -e.x)/|e|: outward for a CCW polygon (FCPW convention) */
    if (s->len > 0.0f) { s->nx = s->ey / s->len; s->ny = -s->ex / s->len; }
    else { s->nx = 0.0f; s->ny = 0.0f; }
    /* distance form (DESIGN.md "segment distance"): centre, unit axis, half length */
    s->cx = fmaf(0.5f, s->ex, s->ax); s->cy = fmaf(0.5f, s->ey, s->ay);
    if (s->len > 0.0f) { s->ux = s->ex / s->len; s->uy = s->ey / s->len; }
    else { s->ux = 1.0f; s->uy = 0.0f; }
    s->hl = 0.5f * s->len;
}

/* ---- BVH build (median split; layout does not influence results) -------- */
typedef struct { float cx, cy; int idx; } cent;
static int cmp_cx(const void *a, const void *b)
{
    const cent *p = a, *q = b;
    if (p->cx < q->cx) return -1; if (p->cx > q->cx) return 1;
    return (p->idx > q->idx) - (p->idx < q->idx);
}
static int cmp_cy(const void *a, const void *b)
{
    const cent *p = a, *q = b;
    if (p->cy < q->cy) return -1; if (p->cy > q->cy) return 1;
    return (p->idx > q->idx) - (p->idx < q->idx);
}

static int bvh_build_rec(pmesh *m, cent *c, int first, int count, float pad)
{
    int id = m->n_nodes++;
    bnode *nd = &m->nodes[id];
    float lox = INFINITY, loy = INFINITY, hix = -INFINITY, hiy = -INFINITY;
    for (int k = first; k < first + count; ++k) {
        const pseg *s = &m->segs[c[k].idx];
        float bx = m->verts[2 * s->i1], by = m->verts[2 * s->i1 + 1];
        lox = fminf(lox, fminf(s->ax, bx)); hix = fmaxf(hix, fmaxf(s->ax, bx));
        loy = fminf(loy, fminf(s->ay, by)); hiy = fmaxf(hiy, fmaxf(s->ay, by));
    }
    nd->lox = lox - pad; nd->loy = loy - pad; nd->hix = hix + pad; nd->hiy = hiy + pad;
    nd->first = first; nd->count = count; nd->left = nd->right = -1;
    if (count > 4) {
        if ((hix - lox) >= (hiy - loy)) qsort(c + first, count, sizeof(cent), cmp_cx);
        else qsort(c + first, count, sizeof(cent), cmp_cy);
        int half = count / 2;
        int l = bvh_build_rec(m, c, first, half, pad);
        int r = bvh_build_rec(m, c, first + half, count - half, pad);
        nd = &m->nodes[id];
        nd->left = l; nd->right = r;
    }
    return id;
}

int pmesh_prepare(pmesh *m, const wo_mesh *in)
{
    memset(m, 0, sizeof(*m));
    m->n_segs = in->n_segs; m->n_verts = in->n_verts;
    m->colors = in->colors; m->verts = in->verts;
    if (in->n_segs <= 0) return 0;
    m->segs = malloc(sizeof(pseg) * in->n_segs);
    m->v_prev = malloc(sizeof(int) * in->n_verts);
    m->v_next = malloc(sizeof(int) * in->n_verts);
    for (int v = 0; v < in->n_verts; ++v) m->v_prev[v] = m->v_next[v] = -1;
    float lox = INFINITY, loy = INFINITY, hix = -INFINITY, hiy = -INFINITY;
    for (int i = 0; i < in->n_segs; ++i) {
        int i0 = in->segs[2 * i], i1 = in->segs[2 * i + 1];
        if (i0 < 0 || i1 < 0 || i0 >= in->n_verts || i1 >= in->n_verts) return -1;
        seg_prepare(&m->segs[i], in->verts, i0, i1);
        /* lowest segment index wins when a vertex has several in/out segments */
        if (m->v_next[i0] < 0) m->v_next[i0] = i;
        if (m->v_prev[i1] < 0) m->v_prev[i1] = i;
        for (int k = 0; k < 2; ++k) {
            int v = k ? i1 : i0;
            lox = fminf(lox, in->verts[2 * v]); hix = fmaxf(hix, in->verts[2 * v]);
            loy = fminf(loy, in->verts[2 * v + 1]); hiy = fmaxf(hiy, in->verts[2 * v + 1]);
        }
    }
    /* conservative absolute pad so that box pruning can never cut a segment whose
     * computed distance ties or beats the current best (DESIGN.md "pruning slack") */
    float ext = fmaxf(fmaxf(fabsf(lox), fabsf(hix)), fmaxf(fabsf(loy), fabsf(hiy)));
    float pad = ext * 0x1p-20f + 1e-30f;
    cent *c = malloc(sizeof(cent) * in->n_segs);
    for (int i = 0; i < in->n_segs; ++i) {
        c[i].idx = i;
        c[i].cx = m->segs[i].ax + 0.5f * m->segs[i].ex;
        c[i].cy = m->segs[i].ay + 0.5f * m->segs[i].ey;
    }
    m->nodes = malloc(sizeof(bnode) * (2 * (size_t)in->n_segs + 1));
    m->n_nodes = 0;
    bvh_build_rec(m, c, 0, in->n_segs, pad);
    m->order = malloc(sizeof(int) * in->n_segs);
    for (int i = 0; i < in->n_segs; ++i) m->order[i] = c[i].idx;
    free(c);
    return 0;
}

/* ---- closest point on one segment (definition of lbvh nearest +
 *      distance_calculator at integrator.cu:138) ---------------------------- */
static inline void seg_closest(const pseg *s, float qx, float qy, float *d2, float *t_raw)
{
    /* squared distance in the segment's own frame: u along the axis, v across it */
    float wx = qx - s->cx, wy = qy - s->cy;
    float u = wo_dot2(wx, wy, s->ux, s->uy);
    float v = wo_cross2(s->ux, s->uy, wx, wy);
    float du = fmaxf(fabsf(u) - s->hl, 0.0f);
    float dv = fabsf(v);
    *d2 = wo_dot2(du, dv, du, dv);
    /* unclamped projection ratio along p0 -> p1 (lbvh::computeProjectionRatio) */
    *t_raw = wo_dot2(qx - s->ax, qy - s->ay, s->ex, s->ey) * s->inv_len2;
}


/* candidate (d2, idx) beats (bd2, bidx): smaller distance, ties -> lower original index */
static inline int cp_better(float d2, int idx, float bd2, int bidx)
{
    return (d2 < bd2) || (d2 == bd2 && idx < bidx);
}

static cp_result closest_brute(const pmesh *m, float qx, float qy)
{
    cp_result r = { -1, INFINITY };
    int bidx = WO_FAR_IDX;
    for (int i = 0; i < m->n_segs; ++i) {
        float d2, tr;
        seg_closest(&m->segs[i], qx, qy, &d2, &tr);
        if (cp_better(d2, i, r.d2, bidx)) { r.d2 = d2; bidx = i; }
    }
    r.idx = (bidx == WO_FAR_IDX) ? -1 : bidx;
    return r;
}

static inline float box_d2(const bnode *n, float qx, float qy)
{
    float dx = fmaxf(fmaxf(n->lox - qx, qx - n->hix), 0.0f);
    float dy = fmaxf(fmaxf(n->loy - qy, qy - n->hiy), 0.0f);
    return wo_dot2(dx, dy, dx, dy);
}

cp_result closest_bvh(const pmesh *m, float qx, float qy)
{
    cp_result r = { -1, INFINITY };
    int bidx = WO_FAR_IDX;
    int stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const bnode *n = &m->nodes[stack[--sp]];
        /* relative slack: the box distance and the exact segment distance are rounded independently, and far outside the
         * scene one ulp of either exceeds the padding of the boxes (a fuzz test met closest points that differed from
         * closest_brute at 50 scene sizes); a box skipped wrongly changes the answer, one opened needlessly does not -- but a
         * slack of 10^-4 opens every box for a walker thousands of scene sizes away */
        if (box_d2(n, qx, qy) > r.d2 * 1.0000153f) continue;       /* 1 + 2^-16: the rounding is a few 10^-7 */
        if (n->left < 0) {
            for (int k = n->first; k < n->first + n->count; ++k) {
                int i = m->order[k];
                float d2, tr;
                seg_closest(&m->segs[i], qx, qy, &d2, &tr);
                if (cp_better(d2, i, r.d2, bidx)) { r.d2 = d2; bidx = i; }
            }
        } else {
            float dl = box_d2(&m->nodes[n->left], qx, qy);
            float dr = box_d2(&m->nodes[n->right], qx, qy);
            if (dl <= dr) { stack[sp++] = n->right; stack[sp++] = n->left; }
            else { stack[sp++] = n->left; stack[sp++] = n->right; }
        }
    }
    r.idx = (bidx == WO_FAR_IDX) ? -1 : bidx;
    return r;
}

/* lbvh::checkPointSide (integrator.cu:148): sign of cross(p1-p0, q-p0), left = +1 */
/* lbvh::computeProjectionRatio (integrator.cu:149): unclamped parameter along p0->p1 */
/* ---- closest silhouette vertex (definition of nearest_silhouette(q,false),
 *      integrator.cu:189; test follows FCPW's isSilhouetteVertex) ----------- */
float closest_silhouette(const pmesh *m, float qx, float qy, float rmax)
{
    float best2 = rmax * rmax; /* inf*inf = inf */
    int found = 0;
    for (int v = 0; v < m->n_verts; ++v) {
        int sp = m->v_prev[v], sn = m->v_next[v];
        if (sp < 0 && sn < 0) continue;
        float vx = qx - m->verts[2 * v], vy = qy - m->verts[2 * v + 1];
        float d2 = wo_dot2(vx, vy, vx, vy);
        if (d2 > best2) continue;
        int is_sil = (sp < 0 || sn < 0);
        if (!is_sil) {
            const pseg *s0 = &m->segs[sp], *s1 = &m->segs[sn];
            float d = sqrtf(d2);
            if (d <= WO_SIL_PRECISION) {
                float det = wo_cross2(s0->nx, s0->ny, s1->nx, s1->ny);
                is_sil = (-det > WO_SIL_PRECISION); /* flipNormalOrientation = false */
            } else {
                float ux = vx / d, uy = vy / d;
                float dot0 = wo_dot2(ux, uy, s0->nx, s0->ny);
                float dot1 = wo_dot2(ux, uy, s1->nx, s1->ny);
                if (fabsf(dot0) <= WO_SIL_PRECISION || fabsf(dot1) <= WO_SIL_PRECISION) is_sil = 0;
                else is_sil = (dot0 * dot1 < 0.0f);
            }
        }
        if (is_sil && (d2 < best2 || !found)) { best2 = d2; found = 1; }
    }
    return found ? sqrtf(best2) : INFINITY;
}

/* ---- ray / segment (definition of ray_intersect + intersect_test,
 *      integrator.cu:385-390,500) ------------------------------------------- */
/* returns 1 and *t when the ray o + t*d, t in [0, tmax], crosses the segment */
static inline int seg_ray(const pseg *s, float ox, float oy, float dx, float dy, float tmax, float *t)
{
    float ux = s->ax - ox, uy = s->ay - oy;
    float dv = wo_cross2(dx, dy, s->ex, s->ey);
    if (dv == 0.0f) return 0;
    float ud = wo_cross2(ux, uy, dx, dy);   /* s * dv */
    float uv = wo_cross2(ux, uy, s->ex, s->ey); /* t * dv */
    float adv = fabsf(dv);
    float sgn = (dv < 0.0f) ? -1.0f : 1.0f;
    float ud_s = ud * sgn, uv_s = uv * sgn;
    if (ud_s < 0.0f || ud_s > adv) return 0;          /* s outside [0,1] */
    if (uv_s < 0.0f || uv_s > tmax * adv) return 0;    /* t outside [0,tmax] */
    *t = uv / dv;
    return 1;
}

int ray_closest(const pmesh *m, float ox, float oy, float dx, float dy, float tmax,
                       float *t_out, int *idx_out)
{
    int hit = 0;
    float bt = INFINITY;
    int bi = -1;
    for (int i = 0; i < m->n_segs; ++i) {
        float t;
        if (seg_ray(&m->segs[i], ox, oy, dx, dy, tmax, &t)) {
            if (!hit || t < bt) { bt = t; bi = i; hit = 1; }
        }
    }
    *t_out = bt; *idx_out = bi;
    return hit;
}

int ray_any(const pmesh *m, float ox, float oy, float dx, float dy, float tmax)
{
    for (int i = 0; i < m->n_segs; ++i) {
        float t;
        if (seg_ray(&m->segs[i], ox, oy, dx, dy, tmax, &t)) return 1;
    }
    return 0;
}

/* ---- primitive sampling in a ball (definition of sample_object_in_sphere,
 *      integrator.cu:349-354): segments touching the ball, probability
 *      proportional to their length, chosen by inverse CDF in index order;
 *      returned pdf is the density w.r.t. arc length: P(i) / len_i. ---------- */
int sample_in_sphere(const pmesh *m, float qx, float qy, float R, float u, float *pdf)
{
    float R2 = R * R;
    float total = 0.0f;
    for (int i = 0; i < m->n_segs; ++i) {
        float d2, tr;
        seg_closest(&m->segs[i], qx, qy, &d2, &tr);
        if (d2 <= R2 && m->segs[i].len > 0.0f) total += m->segs[i].len;
    }
    *pdf = 0.0f;
    if (!(total > 0.0f)) return -1;
    float target = u * total;
    float cum = 0.0f;
    int last = -1;
    for (int i = 0; i < m->n_segs; ++i) {
        float d2, tr;
        seg_closest(&m->segs[i], qx, qy, &d2, &tr);
        if (d2 <= R2 && m->segs[i].len > 0.0f) {
            cum += m->segs[i].len;
            last = i;
            if (target < cum) break;
        }
    }
    *pdf = (m->segs[last].len / total) / m->segs[last].len;
    return last;
}

/* ------------------------------------------------------------------------ */
/* surface colour: integrator/common.h:242-260 + functors.h:60-64            */
/* ------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------ */
/* source term                                                               */
/* ------------------------------------------------------------------------ */
static void source_tap(const wo_source *src, int i, int j, float v[3])
{
    if (i < 0 || j < 0 || i >= src->nx || j >= src->ny) { v[0] = v[1] = v[2] = 0.0f; return; }
    const float *p = src->rgb + 3 * ((size_t)j * src->nx + i);
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
}

void wo_source_eval(const wo_source *src, float x, float y, float out[3])
{
    const float gx = fmaf(x, src->index_scale[0], src->index_offset[0]);
    const float gy = fmaf(y, src->index_scale[1], src->index_offset[1]);
    const float fx = floorf(gx), fy = floorf(gy);
    const float u = gx - fx, v = gy - fy;
    /* keep the integer conversion in range for far-away points */
    const int i = (int)fmaxf(fminf(fx, 1e9f), -1e9f), j = (int)fmaxf(fminf(fy, 1e9f), -1e9f);
    float v00[3], v10[3], v01[3], v11[3];
    source_tap(src, i, j, v00); source_tap(src, i + 1, j, v10);
    source_tap(src, i, j + 1, v01); source_tap(src, i + 1, j + 1, v11);
    for (int c = 0; c < 3; ++c) {
        const float a = v00[c] + (v10[c] - v00[c]) * u;
        const float b = v01[c] + (v11[c] - v01[c]) * u;
        out[c] = (a + (b - a) * v) * src->intensity;
    }
}

int wo_sample_source(const wo_source *src, const pmesh *nm, float eps, float px, float py, float R_B, int on_n,
                     float nx, float ny, float thp, wo_pcg *rng, float out[3])
{
    /* direction: uniform on the circle, or on the half circle around the Neumann normal */
    float dirx, diry, dir_pdf, alpha = 1.0f;
    const float u0 = wo_pcg_next_float(rng);
    if (on_n) {
        float lc, ls;
        wo_sincos_2pi(u0 * 0.5f, &lc, &ls);
        float qx = -ny, qy = nx;
        float ql = sqrtf(wo_dot2(qx, qy, qx, qy));
        float tx = -(qx / ql), ty = -(qy / ql);
        dirx = tx * lc + nx * ls;
        diry = ty * lc + ny * ls;
        dir_pdf = (float)(1.0 / WO_PI_D);
        alpha = 0.5f;
    } else {
        wo_sincos_2pi(u0, &dirx, &diry);
        dir_pdf = 1.0f / WO_2PI;
    }
    /* how far the straight line stays inside the star-shaped region (integrator.cu:279-292) */
    float dist = R_B;
    if (nm->n_segs > 0) {
        float t; int hi;
        if (ray_closest(nm, px + eps * dirx, py + eps * diry, dirx, diry, dist, &t, &hi)) dist = fminf(t, dist);
    }
    /* HarmonicGreenBall<2>::sample (util/green.h:44-73): rejection, at most 1000 trials */
    const float norm = R_B * R_B / 4.0f, bound = 1.5f / R_B;
    float r = 0.0f;
    for (int iter = 0; iter < 1000; ++iter) {
        const float u = wo_pcg_next_float(rng);
        r = wo_pcg_next_float(rng) * R_B;
        const float pdf = (wo_logf(R_B / r) / WO_2PI) / norm;          /* r == 0 -> +inf: accepted */
        const float pdf_radius = pdf / (1.0f / WO_2PI);
        if (u < pdf_radius / bound) break;
    }
    r = fmaxf(1e-4f, r);                                                /* ELAINA_GREEN_FUNC_R_CLAMP */
    if (r > R_B) r = R_B / 2.0f;
    if (!(r <= dist)) return 0;
    float f[3];
    wo_source_eval(src, px + r * dirx, py + r * diry, f);
    const float c1 = (1.0f / WO_2PI) / r, c2 = dir_pdf / r;             /* conditionalSampleSpherePDF<2> */
    for (int c = 0; c < 3; ++c) out[c] = thp * f[c] * norm * c1 / c2 / alpha;
    return 1;
}

int wo_render_source(const wo_scene *sc, const wo_settings *st, float *out_rgb)
{
    if (!sc || !st || !out_rgb) return -1;
    for (int p = 0; p < st->width * st->height; ++p) {
        float x, y;
        wo_eval_point(sc, p % st->width, p / st->width, st->width, st->height, &x, &y);
        if (sc->source.nx > 0) wo_source_eval(&sc->source, x, y, out_rgb + 3 * (size_t)p);
        else out_rgb[3 * (size_t)p] = out_rgb[3 * (size_t)p + 1] = out_rgb[3 * (size_t)p + 2] = 0.0f;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* one pixel: the whole spp x depth loop (integrator.cu:529-623 restated     */
/* per pixel; legal because a pixel's walk only touches its own sampler and  */
/* solution -- workqueue.h:25-29, integrator.cu:255,342,466)                 */
/* ------------------------------------------------------------------------ */
typedef struct {
    const wo_scene *sc;
    const wo_settings *st;
    const pmesh *dm, *nm;
} solve_ctx;

typedef struct {
    uint64_t steps, started, absorbed, truncated, nhits;
} pix_stats;

static void solve_pixel(const solve_ctx *cx, int pixel_id, float sol_out[3], uint32_t *steps_out,
                        uint64_t *depth_hist, pix_stats *ps)
{
    const wo_scene *sc = cx->sc;
    const wo_settings *st = cx->st;
    const int has_d = cx->dm->n_segs > 0, has_n = cx->nm->n_segs > 0;
    const float eps = st->eps_shell;
    wo_pcg rng;
    wo_pcg_seed_pixel(&rng, pixel_id, st->width);       /* integrator.cu:71-77 */
    float sol[3] = { 0.0f, 0.0f, 0.0f };
    uint32_t steps = 0;
    const int masked = sc->mask && sc->mask[pixel_id] == 0; /* integrator.cu:92-95 */

    for (int sample = 0; sample < st->spp && !masked; ++sample) {
        /* generateEvaluationPoints + pushEvaluationPoint (integrator.cu:96-98, workqueue.h:99-110) */
        float px, py;
        wo_eval_point(sc, pixel_id % st->width, pixel_id / st->width, st->width, st->height, &px, &py);
        float thp[3] = { 1.0f, 1.0f, 1.0f };
        int on_n = 0;
        float nnx = 0.0f, nny = 0.0f;
        ps->started++;
        int depth;
        for (depth = 0; depth < st->max_depth; ++depth) {
            steps++;
            if (depth_hist) __atomic_fetch_add(&depth_hist[depth], 1, __ATOMIC_RELAXED);
            /* ---- separateEvaluationPoint (integrator.cu:128-211) ---- */
            float R_D = INFINITY;
            if (has_d) {
                cp_result cp = closest_bvh(cx->dm, px, py);
                const pseg *s = &cx->dm->segs[cp.idx];
                int side = seg_side(s, px, py);
                float uv = seg_proj_ratio(s, px, py);
                R_D = sqrtf(cp.d2);
                int in_shell = (R_D < eps) && (uv > 0.0f && uv < 1.0f);
                if (in_shell) {
                    /* ---- handleBoundary (integrator.cu:224-231) ---- */
                    float col[3];
                    surface_color(cx->dm->colors, s->i0, s->i1, side, uv, col);
                    for (int c = 0; c < 3; ++c) {
                        col[c] *= sc->dirichlet_intensity;
                        col[c] *= thp[c];
                        sol[c] = col[c] + sol[c];       /* workqueue.h:25-29 */
                    }
                    ps->absorbed++;
                    break;
                }
            }
            float R_N = INFINITY;
            if (has_n) {
                /* only min(R_D, R_N) is used below, so the search may be bounded by R_D */
                R_N = closest_silhouette(cx->nm, px, py, R_D);
            }
            float R_B = fmaxf(WO_R_B_FLOOR, fminf(R_D, R_N));
            R_B *= WO_R_B_SHRINK;
            if (isinf(R_B)) break;                      /* integrator.cu:197-200 */

            /* ---- sampleSource (integrator.cu:235-316), only when the problem has a source ---- */
            if (sc->source.nx > 0) {
                float col[3];
                if (wo_sample_source(&sc->source, cx->nm, eps, px, py, R_B, on_n, nnx, nny, thp[0], &rng, col))
                    for (int c = 0; c < 3; ++c) sol[c] = col[c] + sol[c];
            }

            /* ---- sampleNeumann (integrator.cu:336-444) ---- */
            if (has_n) {
                float u0 = wo_pcg_next_float(&rng);
                float u1 = wo_pcg_next_float(&rng);
                float pdf;
                int oi = sample_in_sphere(cx->nm, px, py, R_B, u0, &pdf);
                if (oi != -1 && pdf > 0) {
                    const pseg *so = &cx->nm->segs[oi];
                    float spx = fmaf(u1, so->ex, so->ax), spy = fmaf(u1, so->ey, so->ay);
                    float rx = spx - px, ry = spy - py;
                    float r = sqrtf(wo_dot2(rx, ry, rx, ry));
                    if (r < R_B && r > 0) {
                        int blocked = 0;
                        {
                            float ox = px, oy = py;
                            if (on_n) { ox += eps * nnx; oy += eps * nny; }
                            float dx = spx - ox, dy = spy - oy;
                            float cd = sqrtf(wo_dot2(dx, dy, dx, dy));
                            if (cd > 0) { dx /= cd; dy /= cd; }
                            blocked = ray_any(cx->nm, ox, oy, dx, dy, cd - eps);
                        }
                        if (!blocked) {
                            int side = seg_side(so, px, py);
                            float uv = seg_proj_ratio(so, spx, spy);
                            if (on_n) {
                                float dn = wo_dot2(so->nx, so->ny, nnx, nny);
                                side = (0.0f < dn) - (dn < 0.0f);
                            }
                            if (side != 0) {
                                float col[3];
                                surface_color(cx->nm->colors, so->i0, so->i1, side, uv, col);
                                float alpha = on_n ? 0.5f : 1.0f;
                                float G = wo_logf(R_B / r) / WO_2PI;      /* util/green.h:24-27 */
                                for (int c = 0; c < 3; ++c) {
                                    col[c] *= sc->neumann_intensity;
                                    col[c] *= thp[c] * G / alpha / pdf;
                                    sol[c] = -col[c] + sol[c];
                                }
                            }
                        }
                    }
                }
            }

            /* ---- oneStepWalk (integrator.cu:465-525) ---- */
            float dirx, diry, pdf, alpha = 1.0f;
            float cxp = px, cyp = py;
            if (on_n) {
                /* uniformSampleHemisphere<2> (sampling.h:80-85): phi = pi*u == 2*pi*(u/2) */
                float u = wo_pcg_next_float(&rng);
                float lc, ls;
#ifdef WOST_ORACLE_LIBM
                float phi = (float)(WO_PI_D * (double)u);
                lc = cosf(phi); ls = sinf(phi);
#else
                wo_sincos_2pi(u * 0.5f, &lc, &ls);
#endif
                /* frameFromNormal (transformation.h:52-55): T = -normalize(-n.y, n.x) */
                float qx = -nny, qy = nnx;
                float ql = sqrtf(wo_dot2(qx, qy, qx, qy));
                float tx = -(qx / ql), ty = -(qy / ql);
                dirx = tx * lc + nnx * ls;             /* Frame<2>::toWorld (transformation.h:35-37) */
                diry = ty * lc + nny * ls;
                pdf = (float)(1.0 / WO_PI_D);           /* sampling.h:91-94 */
                alpha = 0.5f;
                cxp += eps * nnx; cyp += eps * nny;     /* integrator.cu:479 */
            } else {
                float u = wo_pcg_next_float(&rng);
                wo_sincos_2pi(u, &dirx, &diry);         /* sampling.h:29-33 */
                pdf = 1.0f / WO_2PI;                    /* sampling.h:39-42 */
            }
            float nx_ = px + R_B * dirx, ny_ = py + R_B * diry; /* integrator.cu:489 */
            int hit = 0;
            float hnx = 0.0f, hny = 0.0f;
            if (has_n) {
                float t; int hi;
                hit = ray_closest(cx->nm, cxp, cyp, dirx, diry, R_B, &t, &hi);
                if (hit) {
                    hnx = cx->nm->segs[hi].nx; hny = cx->nm->segs[hi].ny;
                    if (wo_dot2(hnx, hny, dirx, diry) > 0) { hnx = -hnx; hny = -hny; }
                    nx_ = cxp + t * dirx; ny_ = cyp + t * diry;
                    ps->nhits++;
                }
            }
            for (int c = 0; c < 3; ++c) thp[c] = thp[c] / pdf / alpha / WO_2PI; /* integrator.cu:521 */
            px = nx_; py = ny_;
            on_n = hit; nnx = hnx; nny = hny;
        }
        if (depth == st->max_depth) ps->truncated++;
    }
    /* resolve: integrator.cu:616-620 */
    for (int c = 0; c < 3; ++c) sol_out[c] = sol[c] / (float)st->spp;
    if (steps_out) *steps_out = steps;
    ps->steps += steps;
}

/* ------------------------------------------------------------------------ */
/* threading                                                                 */
/* ------------------------------------------------------------------------ */
typedef struct {
    solve_ctx cx;
    int begin, end;
    int next;               /* atomic chunk cursor */
    float *field;
    uint32_t *steps;
    uint64_t *depth_hist;
    pix_stats total;
    pthread_mutex_t mu;
    int mode;               /* 0 = solve, 1 = sdf */
    float *sdf;
} job;

#define WO_CHUNK 256

static void *worker(void *arg)
{
    job *j = arg;
    pix_stats loc; memset(&loc, 0, sizeof(loc));
    for (;;) {
        int b = __atomic_fetch_add(&j->next, WO_CHUNK, __ATOMIC_RELAXED);
        if (b >= j->end) break;
        int e = b + WO_CHUNK; if (e > j->end) e = j->end;
        for (int p = b; p < e; ++p) {
            if (j->mode == 0) {
                solve_pixel(&j->cx, p, &j->field[3 * (size_t)(p - j->begin)],
                            j->steps ? &j->steps[p - j->begin] : NULL, j->depth_hist, &loc);
            } else {
                const wo_settings *st = j->cx.st;
                float x, y;
                wo_eval_point(j->cx.sc, p % st->width, p / st->width, st->width, st->height, &x, &y);
                float d = INFINITY;
                if (j->cx.dm->n_segs > 0) d = sqrtf(closest_bvh(j->cx.dm, x, y).d2);
                j->sdf[p - j->begin] = d;
            }
        }
    }
    pthread_mutex_lock(&j->mu);
    j->total.steps += loc.steps; j->total.started += loc.started; j->total.absorbed += loc.absorbed;
    j->total.truncated += loc.truncated; j->total.nhits += loc.nhits;
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int run_job(job *j, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    pthread_mutex_init(&j->mu, NULL);
    j->next = j->begin;
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, worker, j);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&j->mu);
    return 0;
}

int wo_solve(const wo_scene *sc, const wo_settings *st, int pixel_begin, int pixel_end,
             int n_threads, float *field_rgb, uint32_t *steps_per_pixel,
             uint64_t *depth_hist, wo_stats *stats)
{
    if (!sc || !st || !field_rgb) return -1;
    if (pixel_begin < 0 || pixel_end > st->width * st->height || pixel_begin > pixel_end) return -1;
    pmesh dm, nm;
    if (pmesh_prepare(&dm, &sc->dirichlet)) return -2;
    if (pmesh_prepare(&nm, &sc->neumann)) { pmesh_free(&dm); return -2; }
    if (depth_hist) memset(depth_hist, 0, sizeof(uint64_t) * st->max_depth);
    job j; memset(&j, 0, sizeof(j));
    j.cx.sc = sc; j.cx.st = st; j.cx.dm = &dm; j.cx.nm = &nm;
    j.begin = pixel_begin; j.end = pixel_end;
    j.field = field_rgb; j.steps = steps_per_pixel; j.depth_hist = depth_hist;
    double t0 = now_s();
    run_job(&j, n_threads);
    double t1 = now_s();
    if (stats) {
        stats->walk_steps = j.total.steps; stats->walks_started = j.total.started;
        stats->walks_absorbed = j.total.absorbed; stats->walks_truncated = j.total.truncated;
        stats->neumann_hits = j.total.nhits; stats->seconds = t1 - t0;
    }
    pmesh_free(&dm); pmesh_free(&nm);
    return 0;
}

int wo_render_dirichlet_sdf(const wo_scene *sc, const wo_settings *st, int n_threads, float *out)
{
    if (!sc || !st || !out) return -1;
    pmesh dm, nm;
    if (pmesh_prepare(&dm, &sc->dirichlet)) return -2;
    memset(&nm, 0, sizeof(nm));
    job j; memset(&j, 0, sizeof(j));
    j.cx.sc = sc; j.cx.st = st; j.cx.dm = &dm; j.cx.nm = &nm;
    j.begin = 0; j.end = st->width * st->height; j.mode = 1; j.sdf = out;
    run_job(&j, n_threads);
    pmesh_free(&dm);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* batch query entry points (used by the tests to pin the lbvh boundary)     */
/* ------------------------------------------------------------------------ */
int wo_closest_point_batch(const wo_mesh *mesh, const float *pts, int n, int mode,
                           int *out_idx, float *out_dist, float *out_uv, int *out_side)
{
    pmesh m;
    if (pmesh_prepare(&m, mesh)) return -2;
    if (m.n_segs <= 0) { pmesh_free(&m); return -1; }
    for (int k = 0; k < n; ++k) {
        float qx = pts[2 * k], qy = pts[2 * k + 1];
        cp_result r = mode ? closest_bvh(&m, qx, qy) : closest_brute(&m, qx, qy);
        if (out_idx) out_idx[k] = r.idx;
        if (out_dist) out_dist[k] = sqrtf(r.d2);
        if (out_uv) out_uv[k] = seg_proj_ratio(&m.segs[r.idx], qx, qy);
        if (out_side) out_side[k] = seg_side(&m.segs[r.idx], qx, qy);
    }
    pmesh_free(&m);
    return 0;
}

int wo_closest_silhouette_batch(const wo_mesh *mesh, const float *pts, const float *rmax,
                                int n, float *out_dist)
{
    pmesh m;
    if (pmesh_prepare(&m, mesh)) return -2;
    for (int k = 0; k < n; ++k)
        out_dist[k] = closest_silhouette(&m, pts[2 * k], pts[2 * k + 1], rmax ? rmax[k] : INFINITY);
    pmesh_free(&m);
    return 0;
}

int wo_ray_intersect_batch(const wo_mesh *mesh, const float *origins, const float *dirs,
                           const float *tmax, int n, int *out_hit, float *out_t, int *out_idx)
{
    pmesh m;
    if (pmesh_prepare(&m, mesh)) return -2;
    for (int k = 0; k < n; ++k) {
        float t; int idx;
        int hit = ray_closest(&m, origins[2 * k], origins[2 * k + 1], dirs[2 * k], dirs[2 * k + 1],
                              tmax[k], &t, &idx);
        out_hit[k] = hit; out_t[k] = t; out_idx[k] = idx;
    }
    pmesh_free(&m);
    return 0;
}

const char *wo_version(void)
{
#ifdef WOST_ORACLE_LIBM
    return "wost-oracle 1 (libm trig)";
#else
    return "wost-oracle 1 (deterministic math)";
#endif
}
