/*
 * wost_internal.h -- types and helpers shared by the translation units of the CPU oracle
 * (wost_oracle.c: uniform solver and geometric queries; wost_guided.c: guided solver).
 * TEST INFRASTRUCTURE ONLY, see wost_oracle.h.
 */
#ifndef WOST_INTERNAL_H
#define WOST_INTERNAL_H

#include <math.h>

#include "wost_oracle.h"

/* ------------------------------------------------------------------------ */
/* constants: core/math/include/krrmath/constants.h:9-13                     */
/* M_PI comes from <math.h> as a double in the reference translation units   */
/* (constants.h only defines it when absent), so expressions such as         */
/* `1.0f / M_PI` are evaluated in double and then rounded to float.          */
/* ------------------------------------------------------------------------ */
#define WO_PI_D 3.14159265358979323846
#define WO_2PI 6.28318530717958647693f
#define WO_R_B_FLOOR 1e-4f           /* integrator.cu:193 */
#define WO_R_B_SHRINK 0.99f          /* integrator.cu:195 */
#define WO_SIL_PRECISION 1e-3f       /* silhouette test tolerance */
#define WO_FAR_IDX 0x7fffffff

/* ------------------------------------------------------------------------ */
/* prepared mesh                                                             */
/* ------------------------------------------------------------------------ */
typedef struct {
    float ax, ay, ex, ey, inv_len2, len, nx, ny;
    float cx, cy, ux, uy, hl;   /* centre, unit axis, half length (distance form) */
    int i0, i1;
} pseg;

typedef struct { float lox, loy, hix, hiy; int left, right, first, count; } bnode;

typedef struct {
    int n_segs, n_verts;
    pseg *segs;
    const float *colors;
    /* silhouette candidates: per vertex incident segments (prev: vertex is i1, next: vertex is i0) */
    int *v_prev, *v_next;
    const float *verts;
    /* BVH over segments */
    bnode *nodes;
    int n_nodes;
    int *order;   /* leaf order -> original segment index */
} pmesh;

typedef struct { int idx; float d2; } cp_result;

static inline float wo_dot2(float ax, float ay, float bx, float by)
{
    return fmaf(ax, bx, ay * by);
}
static inline float wo_cross2(float ax, float ay, float bx, float by)
{
    return fmaf(ax, by, -(ay * bx));
}

/* lbvh::checkPointSide (integrator.cu:148): left of the directed segment is positive */
static inline int seg_side(const pseg *s, float qx, float qy)
{
    float cr = wo_cross2(s->ex, s->ey, qx - s->ax, qy - s->ay);
    return (0.0f < cr) - (cr < 0.0f);
}
/* lbvh::computeProjectionRatio (integrator.cu:149): unclamped parameter along p0->p1 */
static inline float seg_proj_ratio(const pseg *s, float qx, float qy)
{
    return wo_dot2(qx - s->ax, qy - s->ay, s->ex, s->ey) * s->inv_len2;
}

/* computeSurfaceColor<2> + geometric_interpolate (integrator/common.h:242-260) */
static inline void surface_color(const float *colors, int i0, int i1, int side, float uv, float out[3])
{
    for (int c = 0; c < 3; ++c) {
        float a = 0.0f, b = 0.0f;
        if (colors) {
            int off = (side >= 0) ? 0 : 3;
            a = colors[6 * i0 + off + c];
            b = colors[6 * i1 + off + c];
        }
        out[c] = a * (1 - uv) + b * uv;
    }
}

/* bilinear sample of the source grid at a world point, times the intensity */
void wo_source_eval(const wo_source *src, float x, float y, float out[3]);
/* sampleSource of one out-of-shell point (integrator/uniform/integrator.cu:255-314): returns 1 and
 * the contribution to ADD when the sampled point lies inside the star-shaped region */
int wo_sample_source(const wo_source *src, const pmesh *nm, float eps, float px, float py, float R_B, int on_n,
                     float nx, float ny, float thp, wo_pcg *rng, float out[3]);

void pmesh_free(pmesh *m);
int pmesh_prepare(pmesh *m, const wo_mesh *in);
cp_result closest_bvh(const pmesh *m, float qx, float qy);
float closest_silhouette(const pmesh *m, float qx, float qy, float rmax);
int ray_closest(const pmesh *m, float ox, float oy, float dx, float dy, float tmax, float *t_out, int *idx_out);
int ray_any(const pmesh *m, float ox, float oy, float dx, float dy, float tmax);
int sample_in_sphere(const pmesh *m, float qx, float qy, float R, float u, float *pdf);

/* von Mises mixture (wost_vmm.c) */
#define WV_NCOMP 8
typedef struct { float lambda, kappa, mux, muy, ox, oy; } wv_lobe;
typedef struct { wv_lobe sg[WV_NCOMP]; float weight[WV_NCOMP]; float total; } wv_vmm;
void wo_vmm_build(wv_vmm *m, const float *data);
float wo_vmm_pdf(const wv_vmm *m, float wx, float wy);
void wo_vmm_sample(const wv_vmm *m, wo_pcg *rng, float *ox, float *oy);

#endif
