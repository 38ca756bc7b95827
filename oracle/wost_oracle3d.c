/*
 * wost_oracle3d.c -- CPU oracle of the 3-D uniform Walk-on-Stars path.
 *
 * TEST INFRASTRUCTURE ONLY (see wost_oracle.h).  Restates, line by line, the DIM == 3 branches of
 *   integrator/uniform/integrator.cu:128-211 (separateEvaluationPoint: triangle, barycentric uv, :150-168),
 *   :224-231 (handleBoundary), :336-444 (sampleNeumann, u[3]), :465-525 (oneStepWalk),
 *   integrator/common.h:242-260 + core/math/include/krrmath/functors.h:66-76 (barycentric_interpolate),
 *   core/evaluation_grid.h:43-70 (EvaluationGrid<3>), util/sampling.h:20-27 (uniformSampleSphere<3>),
 *   :57-66 (uniformSampleHemisphere<3>), :39-52,91-104,112-115 (pdfs, sphereMeasurement<3>),
 *   util/transformation.h:11-22,62-67 (Frame<3>, frameFromNormal(Vector3f)),
 *   util/math_utils.h:141-151 (getPerpendicular(Vector3f)), util/green.h:77-119 (HarmonicGreenBall<3>).
 * The source term in 3-D (sampleSource with a nanovdb volume) is not restated.
 *
 * PARITY STATUS: unpinned.  Every geometric query of the path is snch-lbvh (absent); the reference
 * ships no 3-D scene, test or golden vector.  As in 2-D the oracle implements the mathematical
 * definition of each query (exact closest point on a triangle mesh, closest silhouette edge, first
 * ray hit) by brute force and is checked against analytic harmonic solutions.
 *
 * Arithmetic contract, fp32, -ffp-contract=off, fmaf only where written:
 *   dot3(a,b)  = fmaf(a.x, b.x, fmaf(a.y, b.y, a.z*b.z))
 *   cross3(a,b) = ( fmaf(a.y,b.z,-(a.z*b.y)), fmaf(a.z,b.x,-(a.x*b.z)), fmaf(a.x,b.y,-(a.y*b.x)) )
 *   closest point on a triangle: the region walk of Ericson, Real-Time Collision Detection 5.1.5,
 *   written out below; ties between triangles go to the lowest index.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "wost_detmath.h"
#include "wost_internal.h"
#include "wost_oracle.h"

#define WO_4PI 12.5663706143591729539f

typedef struct { float x, y, z; } v3;

static inline v3 v3_sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline float dot3(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
static inline v3 cross3(v3 a, v3 b)
{
    v3 r = { fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)) };
    return r;
}
/* p + t * d, one fmaf per component */
static inline v3 v3_madd(v3 p, float t, v3 d) { v3 r = { fmaf(t, d.x, p.x), fmaf(t, d.y, p.y), fmaf(t, d.z, p.z) }; return r; }
static inline v3 v3_normalize(v3 a)
{
    const float l = sqrtf(dot3(a, a));
    v3 r = { a.x / l, a.y / l, a.z / l };
    return r;
}

/* ---- prepared triangle mesh ------------------------------------------------------------- */
typedef struct {
    v3 p0, e0, e1;       /* p0, p1 - p0, p2 - p0 */
    v3 p1, p2;
    v3 nraw, n;          /* cross3(e0, e1) and the unit normal (0 when degenerate) */
    float area;
    int i0, i1, i2;
    v3 bc;               /* bounding sphere, pruning only */
    float br;
} ptri;

typedef struct { int a, b, t0, t1; } pedge;   /* vertices in the winding order of t0; t1 = -1: boundary edge */

typedef struct {
    int n_tris, n_verts, n_edges;
    ptri *tris;
    pedge *edges;
    const float *verts, *colors;
} pmesh3;

static v3 vert3(const float *v, int i) { v3 r = { v[3 * i], v[3 * i + 1], v[3 * i + 2] }; return r; }

static void pmesh3_free(pmesh3 *m) { free(m->tris); free(m->edges); memset(m, 0, sizeof(*m)); }

typedef struct { int a, b, t, k; } edge_key;
static int edge_key_cmp(const void *x, const void *y)
{
    const edge_key *p = x, *q = y;
    if (p->a != q->a) return p->a < q->a ? -1 : 1;
    if (p->b != q->b) return p->b < q->b ? -1 : 1;
    if (p->t != q->t) return p->t < q->t ? -1 : 1;
    return 0;
}

static int pmesh3_prepare(pmesh3 *m, const wo3_mesh *in)
{
    memset(m, 0, sizeof(*m));
    m->n_tris = in->n_tris; m->n_verts = in->n_verts; m->verts = in->verts; m->colors = in->colors;
    if (in->n_tris <= 0) return 0;
    m->tris = calloc((size_t)in->n_tris, sizeof(ptri));
    for (int t = 0; t < in->n_tris; ++t) {
        ptri *T = &m->tris[t];
        T->i0 = in->tris[3 * t]; T->i1 = in->tris[3 * t + 1]; T->i2 = in->tris[3 * t + 2];
        if (T->i0 < 0 || T->i1 < 0 || T->i2 < 0 || T->i0 >= in->n_verts || T->i1 >= in->n_verts || T->i2 >= in->n_verts) return -1;
        T->p0 = vert3(in->verts, T->i0); T->p1 = vert3(in->verts, T->i1); T->p2 = vert3(in->verts, T->i2);
        T->e0 = v3_sub(T->p1, T->p0); T->e1 = v3_sub(T->p2, T->p0);
        T->nraw = cross3(T->e0, T->e1);
        const float l = sqrtf(dot3(T->nraw, T->nraw));
        T->area = 0.5f * l;
        if (l > 0.0f) { T->n.x = T->nraw.x / l; T->n.y = T->nraw.y / l; T->n.z = T->nraw.z / l; }
        /* bounding sphere around the centroid (double, padded): pruning only */
        double cx = ((double)T->p0.x + T->p1.x + T->p2.x) / 3.0, cy = ((double)T->p0.y + T->p1.y + T->p2.y) / 3.0,
               cz = ((double)T->p0.z + T->p1.z + T->p2.z) / 3.0, r = 0.0;
        const v3 P[3] = { T->p0, T->p1, T->p2 };
        for (int k = 0; k < 3; ++k) {
            const double dx = P[k].x - cx, dy = P[k].y - cy, dz = P[k].z - cz, d = sqrt(dx * dx + dy * dy + dz * dz);
            if (d > r) r = d;
        }
        T->bc.x = (float)cx; T->bc.y = (float)cy; T->bc.z = (float)cz;
        T->br = (float)(r * 1.001 + 1e-6 * (fabs(cx) + fabs(cy) + fabs(cz) + 1.0));
    }
    /* edges: (min vertex, max vertex) -> the first two incident triangles in index order; the edge
     * keeps the direction it has in its first triangle */
    edge_key *keys = malloc(sizeof(edge_key) * 3 * (size_t)in->n_tris);
    for (int t = 0; t < in->n_tris; ++t) {
        const int v[3] = { m->tris[t].i0, m->tris[t].i1, m->tris[t].i2 };
        for (int k = 0; k < 3; ++k) {
            const int a = v[k], b = v[(k + 1) % 3];
            edge_key e = { a < b ? a : b, a < b ? b : a, t, k };
            keys[3 * t + k] = e;
        }
    }
    qsort(keys, 3 * (size_t)in->n_tris, sizeof(edge_key), edge_key_cmp);
    m->edges = malloc(sizeof(pedge) * 3 * (size_t)in->n_tris);
    int ne = 0;
    for (int i = 0; i < 3 * in->n_tris;) {
        int j = i;
        while (j < 3 * in->n_tris && keys[j].a == keys[i].a && keys[j].b == keys[i].b) ++j;
        const int t0 = keys[i].t, k0 = keys[i].k;
        const int v[3] = { m->tris[t0].i0, m->tris[t0].i1, m->tris[t0].i2 };
        pedge e = { v[k0], v[(k0 + 1) % 3], t0, (j - i >= 2) ? keys[i + 1].t : -1 };
        if (e.a != e.b) m->edges[ne++] = e;
        i = j;
    }
    free(keys);
    m->n_edges = ne;
    return 0;
}

/* closest point of the triangle to q (Ericson 5.1.5), squared distance */
static float tri_d2(const ptri *T, v3 q)
{
    const v3 ab = T->e0, ac = T->e1, ap = v3_sub(q, T->p0);
    const float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    v3 c;
    if (d1 <= 0.0f && d2 <= 0.0f) c = T->p0;
    else {
        const v3 bp = v3_sub(q, T->p1);
        const float d3 = dot3(ab, bp), d4 = dot3(ac, bp);
        if (d3 >= 0.0f && d4 <= d3) c = T->p1;
        else {
            const float vc = fmaf(d1, d4, -(d3 * d2));
            if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) c = v3_madd(T->p0, d1 / (d1 - d3), ab);
            else {
                const v3 cp = v3_sub(q, T->p2);
                const float d5 = dot3(ab, cp), d6 = dot3(ac, cp);
                if (d6 >= 0.0f && d5 <= d6) c = T->p2;
                else {
                    const float vb = fmaf(d5, d2, -(d1 * d6));
                    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) c = v3_madd(T->p0, d2 / (d2 - d6), ac);
                    else {
                        const float va = fmaf(d3, d6, -(d5 * d4));
                        if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f)
                            c = v3_madd(T->p1, (d4 - d3) / ((d4 - d3) + (d5 - d6)), v3_sub(T->p2, T->p1));
                        else {
                            const float denom = 1.0f / (va + vb + vc);
                            c = v3_madd(v3_madd(T->p0, vb * denom, ab), vc * denom, ac);
                        }
                    }
                }
            }
        }
    }
    const v3 w = v3_sub(q, c);
    return dot3(w, w);
}

typedef struct { int idx; float d2; } cp3;

static cp3 closest_tri(const pmesh3 *m, v3 q)
{
    cp3 best = { -1, INFINITY };
    for (int t = 0; t < m->n_tris; ++t) {
        const ptri *T = &m->tris[t];
        /* bounding-sphere rejection in double with slack: cannot drop a triangle that ties or wins */
        if (best.idx >= 0) {
            const double dx = (double)q.x - T->bc.x, dy = (double)q.y - T->bc.y, dz = (double)q.z - T->bc.z;
            const double lb = sqrt(dx * dx + dy * dy + dz * dz) - T->br;
            if (lb > 0.0 && lb * lb * (1.0 - 1e-5) > (double)best.d2) continue;
        }
        const float d = tri_d2(T, q);
        if (d < best.d2) { best.d2 = d; best.idx = t; }     /* ascending t: lowest index wins ties */
    }
    return best;
}

/* lbvh::checkPointSide(p0, p1, p2, q): side of the plane the (unnormalised) normal points to */
static int tri_side(const ptri *T, v3 q)
{
    const float s = dot3(T->nraw, v3_sub(q, T->p0));
    return (0.0f < s) - (s < 0.0f);
}
/* lbvh::computeProjectionRatio(p0, p1, p2, q): barycentric (u, v) of the projection of q onto the
 * plane, point = (1-u-v) p0 + u p1 + v p2 (functors.h:66-76), unclamped */
static void tri_uv(const ptri *T, v3 q, float *u, float *v)
{
    const v3 ap = v3_sub(q, T->p0);
    const float d00 = dot3(T->e0, T->e0), d01 = dot3(T->e0, T->e1), d11 = dot3(T->e1, T->e1);
    const float d20 = dot3(ap, T->e0), d21 = dot3(ap, T->e1);
    const float denom = fmaf(d00, d11, -(d01 * d01));
    *u = fmaf(d11, d20, -(d01 * d21)) / denom;
    *v = fmaf(d00, d21, -(d01 * d20)) / denom;
}

/* computeSurfaceColor<3> + barycentric_interpolate: (a w + b u) + c v, w = 1 - u - v */
static void surface_color3(const float *colors, const ptri *T, int side, float u, float v, float out[3])
{
    const float w = 1 - u - v;
    for (int c = 0; c < 3; ++c) {
        float a = 0.0f, b = 0.0f, cc = 0.0f;
        if (colors) {
            const int off = (side >= 0) ? 0 : 3;
            a = colors[6 * T->i0 + off + c]; b = colors[6 * T->i1 + off + c]; cc = colors[6 * T->i2 + off + c];
        }
        out[c] = (a * w + b * u) + cc * v;
    }
}

/* closest silhouette EDGE (FCPW isSilhouetteEdge, flipNormalOrientation = false), within rmax */
static float closest_silhouette3(const pmesh3 *m, v3 q, float rmax)
{
    float best2 = rmax * rmax;
    int found = 0;
    for (int i = 0; i < m->n_edges; ++i) {
        const pedge *E = &m->edges[i];
        const v3 pa = vert3(m->verts, E->a), pb = vert3(m->verts, E->b), e = v3_sub(pb, pa);
        const float ee = dot3(e, e);
        float t = ee > 0.0f ? dot3(v3_sub(q, pa), e) / ee : 0.0f;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const v3 pt = v3_madd(pa, t, e), view = v3_sub(q, pt);
        const float d2 = dot3(view, view);
        if (d2 > best2) continue;
        int is_sil = E->t1 < 0;
        if (!is_sil) {
            const v3 n0 = m->tris[E->t0].n, n1 = m->tris[E->t1].n;
            const float d = sqrtf(d2);
            if (d <= WO_SIL_PRECISION) {
                const v3 ed = v3_normalize(e);
                const float det = dot3(ed, cross3(n0, n1));
                is_sil = (-det > WO_SIL_PRECISION);
            } else {
                const v3 vd = { view.x / d, view.y / d, view.z / d };
                const float dot0 = dot3(vd, n0), dot1 = dot3(vd, n1);
                if (fabsf(dot0) <= WO_SIL_PRECISION || fabsf(dot1) <= WO_SIL_PRECISION) is_sil = 0;
                else is_sil = (dot0 * dot1 < 0.0f);
            }
        }
        if (is_sil && (d2 < best2 || !found)) { best2 = d2; found = 1; }
    }
    return found ? sqrtf(best2) : INFINITY;
}

/* ray / triangle (Moeller-Trumbore with the sign folded out, like the 2-D segment test) */
static int tri_ray(const ptri *T, v3 o, v3 d, float tmax, float *t)
{
    const v3 pvec = cross3(d, T->e1);
    const float det = dot3(T->e0, pvec);
    if (det == 0.0f) return 0;
    const v3 tvec = v3_sub(o, T->p0);
    const float sgn = det < 0.0f ? -1.0f : 1.0f, adet = fabsf(det);
    const float u = dot3(tvec, pvec) * sgn;
    if (u < 0.0f || u > adet) return 0;
    const v3 qvec = cross3(tvec, T->e0);
    const float v = dot3(d, qvec) * sgn;
    if (v < 0.0f || u + v > adet) return 0;
    const float tt = dot3(T->e1, qvec);
    const float ts = tt * sgn;
    if (ts < 0.0f || ts > tmax * adet) return 0;
    *t = tt / det;
    return 1;
}
static int ray_closest3(const pmesh3 *m, v3 o, v3 d, float tmax, float *t_out, int *idx_out)
{
    int hit = 0, bi = -1;
    float bt = INFINITY;
    for (int i = 0; i < m->n_tris; ++i) {
        float t;
        if (tri_ray(&m->tris[i], o, d, tmax, &t) && (!hit || t < bt)) { bt = t; bi = i; hit = 1; }
    }
    *t_out = bt; *idx_out = bi;
    return hit;
}
static int ray_any3(const pmesh3 *m, v3 o, v3 d, float tmax)
{
    for (int i = 0; i < m->n_tris; ++i) { float t; if (tri_ray(&m->tris[i], o, d, tmax, &t)) return 1; }
    return 0;
}

/* sample_object_in_sphere for triangles: those touching the ball, probability ~ area, inverse CDF
 * in index order, pdf = P(i) / area(i) (density with respect to area) */
static int sample_in_sphere3(const pmesh3 *m, v3 q, float R, float u, float *pdf)
{
    const float R2 = R * R;
    float total = 0.0f;
    for (int i = 0; i < m->n_tris; ++i)
        if (m->tris[i].area > 0.0f && tri_d2(&m->tris[i], q) <= R2) total += m->tris[i].area;
    *pdf = 0.0f;
    if (!(total > 0.0f)) return -1;
    const float target = u * total;
    float cum = 0.0f;
    int last = -1;
    for (int i = 0; i < m->n_tris; ++i)
        if (m->tris[i].area > 0.0f && tri_d2(&m->tris[i], q) <= R2) {
            cum += m->tris[i].area; last = i;
            if (target < cum) break;
        }
    *pdf = (m->tris[last].area / total) / m->tris[last].area;
    return last;
}

/* getPerpendicular(Vector3f) + frameFromNormal(Vector3f) + Frame<3>::toWorld */
static v3 frame_to_world(v3 n, float lx, float ly, float lz)
{
    const float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    const unsigned uyx = (ax - ay) < 0 ? 1 : 0, uzx = (ax - az) < 0 ? 1 : 0, uzy = (ay - az) < 0 ? 1 : 0;
    const unsigned xm = uyx & uzx, ym = (1 ^ xm) & uzy, zm = 1 ^ (xm | ym);
    const v3 axis = { (float)xm, (float)ym, (float)zm };
    const v3 t = v3_normalize(cross3(n, axis)), b = v3_normalize(cross3(n, t));
    v3 r = { (t.x * lx + b.x * ly) + n.x * lz, (t.y * lx + b.y * ly) + n.y * lz, (t.z * lx + b.z * ly) + n.z * lz };
    return r;
}

/* EvaluationGrid<3>::getEvaluationPoint (core/evaluation_grid.h:57-61) */
static v3 eval_point3(const wo3_scene *sc, int px, int py, int width, int height)
{
    const float ndcx = 2.0f * (float)px / (float)width + -1.0f, ndcy = 2.0f * (float)py / (float)height + -1.0f;
    v3 r;
    r.x = sc->probe_scale * (ndcx * sc->probe_right[0] + ndcy * sc->probe_up[0]) + sc->probe_pos[0];
    r.y = sc->probe_scale * (ndcx * sc->probe_right[1] + ndcy * sc->probe_up[1]) + sc->probe_pos[1];
    r.z = sc->probe_scale * (ndcx * sc->probe_right[2] + ndcy * sc->probe_up[2]) + sc->probe_pos[2];
    return r;
}

/* ---- source term (integrator/uniform/integrator.cu:235-316, DIM == 3) ---- */
static void source3_tap(const wo3_source *src, int i, int j, int k, float v[3])
{
    if (i < 0 || j < 0 || k < 0 || i >= src->nx || j >= src->ny || k >= src->nz) { v[0] = v[1] = v[2] = 0.0f; return; }
    const float *p = src->rgb + 3 * (((size_t)k * src->ny + j) * src->nx + i);
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
}

void wo3_source_eval(const wo3_source *src, float x, float y, float z, float out[3])
{
    const float gx = fmaf(x, src->index_scale[0], src->index_offset[0]);
    const float gy = fmaf(y, src->index_scale[1], src->index_offset[1]);
    const float gz = fmaf(z, src->index_scale[2], src->index_offset[2]);
    const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
    const float u = gx - fx, v = gy - fy, w = gz - fz;
    const int i = (int)fmaxf(fminf(fx, 1e9f), -1e9f), j = (int)fmaxf(fminf(fy, 1e9f), -1e9f), k = (int)fmaxf(fminf(fz, 1e9f), -1e9f);
    float c[2][2][2][3];
    for (int dk = 0; dk < 2; ++dk)
        for (int dj = 0; dj < 2; ++dj)
            for (int di = 0; di < 2; ++di) source3_tap(src, i + di, j + dj, k + dk, c[dk][dj][di]);
    for (int ch = 0; ch < 3; ++ch) {
        /* x, then y, then z */
        const float a00 = c[0][0][0][ch] + (c[0][0][1][ch] - c[0][0][0][ch]) * u, a01 = c[0][1][0][ch] + (c[0][1][1][ch] - c[0][1][0][ch]) * u;
        const float a10 = c[1][0][0][ch] + (c[1][0][1][ch] - c[1][0][0][ch]) * u, a11 = c[1][1][0][ch] + (c[1][1][1][ch] - c[1][1][0][ch]) * u;
        const float b0 = a00 + (a01 - a00) * v, b1 = a10 + (a11 - a10) * v;
        out[ch] = (b0 + (b1 - b0) * w) * src->intensity;
    }
}

/* cube root of x in [0, 1] through the deterministic log / exp (the arithmetic contract's std::cbrt) */
static float cbrt01(float x) { return x > 0.0f ? wo_expf(wo_logf(x) * (1.0f / 3.0f)) : 0.0f; }

typedef struct { uint64_t steps, started, absorbed, truncated, nhits; } pix3_stats;

static void solve_pixel3(const wo3_scene *sc, const wo_settings *st, const pmesh3 *dm, const pmesh3 *nm, int pixel_id,
                         float sol_out[3], pix3_stats *ps)
{
    const int has_d = dm->n_tris > 0, has_n = nm->n_tris > 0;
    const float eps = st->eps_shell;
    wo_pcg rng;
    wo_pcg_seed_pixel(&rng, pixel_id, st->width);
    float sol[3] = { 0.0f, 0.0f, 0.0f };
    const int masked = sc->mask && sc->mask[pixel_id] == 0;
    for (int sample = 0; sample < st->spp && !masked; ++sample) {
        v3 p = eval_point3(sc, pixel_id % st->width, pixel_id / st->width, st->width, st->height);
        float thp = 1.0f;
        int on_n = 0;
        v3 nn = { 0.0f, 0.0f, 0.0f };
        ps->started++;
        int depth;
        for (depth = 0; depth < st->max_depth; ++depth) {
            ps->steps++;
            /* ---- separateEvaluationPoint (integrator.cu:128-211, DIM == 3) ---- */
            float R_D = INFINITY;
            if (has_d) {
                const cp3 cp = closest_tri(dm, p);
                const ptri *T = &dm->tris[cp.idx];
                const int side = tri_side(T, p);
                float u, v;
                tri_uv(T, p, &u, &v);
                R_D = sqrtf(cp.d2);
                if (R_D < eps && u > 0.0f && v > 0.0f && u + v < 1.0f) {
                    float col[3];
                    surface_color3(dm->colors, T, side, u, v, col);
                    for (int c = 0; c < 3; ++c) {
                        col[c] *= sc->dirichlet_intensity;
                        col[c] *= thp;
                        sol[c] = col[c] + sol[c];
                    }
                    ps->absorbed++;
                    break;
                }
            }
            float R_N = INFINITY;
            if (has_n) R_N = closest_silhouette3(nm, p, R_D);
            float R_B = fmaxf(WO_R_B_FLOOR, fminf(R_D, R_N));
            R_B *= WO_R_B_SHRINK;
            if (isinf(R_B)) break;
            /* ---- sampleSource (integrator.cu:235-316), only when the problem has a source ---- */
            if (sc->source.nx > 0) {
                /* direction: uniform on the sphere, or on the hemisphere around the Neumann normal (the draws of oneStepWalk) */
                v3 sdir;
                float dir_pdf, salpha = 1.0f;
                {
                    const float u1 = wo_pcg_next_float(&rng), u2 = wo_pcg_next_float(&rng);
                    float c, s;
                    wo_sincos_2pi(u2, &c, &s);
                    if (on_n) {
                        const float z = u1, r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
                        sdir = frame_to_world(nn, r * c, r * s, z);
                        dir_pdf = 1.0f / WO_2PI;
                        salpha = 0.5f;
                    } else {
                        const float z = 1 - 2 * u1, r = sqrtf(1 - z * z);
                        sdir = (v3){ r * c, r * s, z };
                        dir_pdf = 1.0f / WO_4PI;
                    }
                }
                /* how far the straight line stays inside the star-shaped region (:279-292) */
                float dist = R_B;
                if (has_n) {
                    float t; int hi;
                    const v3 o = { p.x + eps * sdir.x, p.y + eps * sdir.y, p.z + eps * sdir.z };
                    if (ray_closest3(nm, o, sdir, dist, &t, &hi)) dist = fminf(t, dist);
                }
                /* HarmonicGreenBall<3>::sample (util/green.h:101-116): closed form, two draws */
                const float g1 = wo_pcg_next_float(&rng), g2 = wo_pcg_next_float(&rng);
                float gc, gs;
                wo_sincos_2pi(g2, &gc, &gs);
                float r = (1.0f + sqrtf(1.0f - cbrt01(g1 * g1)) * gc) * R_B / 2.0f;
                r = fmaxf(1e-4f, r);                                            /* ELAINA_GREEN_FUNC_R_CLAMP */
                if (r > R_B) r = R_B / 2.0f;
                if (r <= dist) {
                    float f[3];
                    wo3_source_eval(&sc->source, p.x + r * sdir.x, p.y + r * sdir.y, p.z + r * sdir.z, f);
                    const float norm = R_B * R_B / 6.0f;
                    const float c1 = (1.0f / WO_4PI) / (r * r), c2 = dir_pdf / (r * r);     /* conditionalSampleSpherePDF<3> */
                    for (int c = 0; c < 3; ++c) {
                        const float col = thp * f[c] * norm * c1 / c2 / salpha;
                        sol[c] = col + sol[c];
                    }
                }
            }
            /* ---- sampleNeumann (integrator.cu:336-444): three draws in 3-D ---- */
            if (has_n) {
                const float u0 = wo_pcg_next_float(&rng), u1 = wo_pcg_next_float(&rng), u2 = wo_pcg_next_float(&rng);
                float pdf;
                const int oi = sample_in_sphere3(nm, p, R_B, u0, &pdf);
                if (oi != -1 && pdf > 0) {
                    const ptri *S = &nm->tris[oi];
                    /* sample_on_object: uniform point of the triangle from (u1, u2) */
                    const float su = sqrtf(u1), b1 = u2 * su, b0 = 1.0f - su;
                    const float b2 = 1.0f - b0 - b1;
                    v3 sp = { (S->p0.x * b0 + S->p1.x * b1) + S->p2.x * b2, (S->p0.y * b0 + S->p1.y * b1) + S->p2.y * b2,
                              (S->p0.z * b0 + S->p1.z * b1) + S->p2.z * b2 };
                    const v3 rv = v3_sub(sp, p);
                    const float r = sqrtf(dot3(rv, rv));
                    if (r < R_B && r > 0) {
                        v3 o = p;
                        if (on_n) o = (v3){ p.x + eps * nn.x, p.y + eps * nn.y, p.z + eps * nn.z };
                        v3 rd = v3_sub(sp, o);
                        const float cd = sqrtf(dot3(rd, rd));
                        if (cd > 0) { rd.x /= cd; rd.y /= cd; rd.z /= cd; }
                        if (!ray_any3(nm, o, rd, cd - eps)) {
                            int side = tri_side(S, p);
                            float uu, vv;
                            tri_uv(S, sp, &uu, &vv);
                            if (on_n) {
                                const float dn = dot3(S->n, nn);
                                side = (0.0f < dn) - (dn < 0.0f);
                            }
                            if (side != 0) {
                                float col[3];
                                surface_color3(nm->colors, S, side, uu, vv, col);
                                const float alpha = on_n ? 0.5f : 1.0f;
                                const float G = (1.0f / r - 1.0f / R_B) / WO_4PI;      /* HarmonicGreenBall<3>::eval */
                                for (int c = 0; c < 3; ++c) {
                                    col[c] *= sc->neumann_intensity;
                                    col[c] *= thp * G / alpha / pdf;
                                    sol[c] = -col[c] + sol[c];
                                }
                            }
                        }
                    }
                }
            }
            /* ---- oneStepWalk (integrator.cu:465-525) ---- */
            v3 dir, cur = p;
            float pdf, alpha = 1.0f;
            {
                const float u1 = wo_pcg_next_float(&rng), u2 = wo_pcg_next_float(&rng);
                float c, s;
                wo_sincos_2pi(u2, &c, &s);
                if (on_n) {
                    const float z = u1, r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
                    dir = frame_to_world(nn, r * c, r * s, z);
                    pdf = 1.0f / WO_2PI;
                    alpha = 0.5f;
                    cur = (v3){ p.x + eps * nn.x, p.y + eps * nn.y, p.z + eps * nn.z };
                } else {
                    const float z = 1 - 2 * u1, r = sqrtf(1 - z * z);
                    dir = (v3){ r * c, r * s, z };
                    pdf = 1.0f / WO_4PI;
                }
            }
            v3 nxt = { p.x + R_B * dir.x, p.y + R_B * dir.y, p.z + R_B * dir.z };
            int hit = 0;
            v3 hn = { 0.0f, 0.0f, 0.0f };
            if (has_n) {
                float t; int hi;
                hit = ray_closest3(nm, cur, dir, R_B, &t, &hi);
                if (hit) {
                    hn = nm->tris[hi].n;
                    if (dot3(hn, dir) > 0) { hn.x = -hn.x; hn.y = -hn.y; hn.z = -hn.z; }
                    nxt = (v3){ cur.x + t * dir.x, cur.y + t * dir.y, cur.z + t * dir.z };
                    ps->nhits++;
                }
            }
            thp = thp / pdf / alpha / WO_4PI;
            p = nxt; on_n = hit; nn = hn;
        }
        if (depth == st->max_depth) ps->truncated++;
    }
    for (int c = 0; c < 3; ++c) sol_out[c] = sol[c] / (float)st->spp;
}

int wo3_solve(const wo3_scene *sc, const wo_settings *st, int pixel_begin, int pixel_end, int n_threads, float *field_rgb,
              wo_stats *stats)
{
    if (!sc || !st || !field_rgb || pixel_begin < 0 || pixel_end > st->width * st->height || pixel_begin > pixel_end) return -1;
    pmesh3 dm, nm;
    if (pmesh3_prepare(&dm, &sc->dirichlet) != 0 || pmesh3_prepare(&nm, &sc->neumann) != 0) return -1;
    uint64_t steps = 0, started = 0, absorbed = 0, truncated = 0, nhits = 0;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads) reduction(+ : steps, started, absorbed, truncated, nhits)
    for (int pid = pixel_begin; pid < pixel_end; ++pid) {
        pix3_stats ps = { 0, 0, 0, 0, 0 };
        solve_pixel3(sc, st, &dm, &nm, pid, field_rgb + 3 * (size_t)(pid - pixel_begin), &ps);
        steps += ps.steps; started += ps.started; absorbed += ps.absorbed; truncated += ps.truncated; nhits += ps.nhits;
    }
    if (stats) {
        stats->walk_steps = steps; stats->walks_started = started; stats->walks_absorbed = absorbed;
        stats->walks_truncated = truncated; stats->neumann_hits = nhits; stats->seconds = 0.0;
    }
    pmesh3_free(&dm); pmesh3_free(&nm);
    return 0;
}

int wo3_closest_point_batch(const wo3_mesh *mesh, const float *pts, int n, int *out_idx, float *out_dist, float *out_uv,
                            int *out_side)
{
    pmesh3 m;
    if (pmesh3_prepare(&m, mesh) != 0 || m.n_tris == 0) return -1;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const v3 q = { pts[3 * i], pts[3 * i + 1], pts[3 * i + 2] };
        const cp3 c = closest_tri(&m, q);
        if (out_idx) out_idx[i] = c.idx;
        if (out_dist) out_dist[i] = sqrtf(c.d2);
        if (out_uv) tri_uv(&m.tris[c.idx], q, &out_uv[2 * i], &out_uv[2 * i + 1]);
        if (out_side) out_side[i] = tri_side(&m.tris[c.idx], q);
    }
    pmesh3_free(&m);
    return 0;
}

int wo3_render_sdf(const wo3_scene *sc, const wo_settings *st, int which, float *out_dist)
{
    pmesh3 m;
    if (!sc || !st || !out_dist || (which != 0 && which != 1)) return -1;
    if (pmesh3_prepare(&m, which == 0 ? &sc->dirichlet : &sc->neumann) != 0) return -1;
    const int n = st->width * st->height;
    for (int i = 0; i < n; ++i) {
        const v3 q = eval_point3(sc, i % st->width, i / st->width, st->width, st->height);
        float d = INFINITY;
        if (m.n_tris > 0) d = which == 0 ? sqrtf(closest_tri(&m, q).d2) : closest_silhouette3(&m, q, INFINITY);
        out_dist[i] = d;
    }
    pmesh3_free(&m);
    return 0;
}

int wo3_render_source(const wo3_scene *sc, const wo_settings *st, float *out_rgb)
{
    if (!sc || !st || !out_rgb) return -1;
    const int n = st->width * st->height;
    for (int i = 0; i < n; ++i) {
        const v3 q = eval_point3(sc, i % st->width, i / st->width, st->width, st->height);
        if (sc->source.nx > 0) wo3_source_eval(&sc->source, q.x, q.y, q.z, out_rgb + 3 * (size_t)i);
        else out_rgb[3 * (size_t)i] = out_rgb[3 * (size_t)i + 1] = out_rgb[3 * (size_t)i + 2] = 0.0f;
    }
    return 0;
}

int wo3_closest_silhouette_batch(const wo3_mesh *mesh, const float *pts, const float *rmax, int n, float *out_dist)
{
    pmesh3 m;
    if (pmesh3_prepare(&m, mesh) != 0) return -1;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const v3 q = { pts[3 * i], pts[3 * i + 1], pts[3 * i + 2] };
        out_dist[i] = m.n_tris ? closest_silhouette3(&m, q, rmax ? rmax[i] : INFINITY) : INFINITY;
    }
    pmesh3_free(&m);
    return 0;
}

int wo3_ray_intersect_batch(const wo3_mesh *mesh, const float *origins, const float *dirs, const float *tmax, int n,
                            int *out_hit, float *out_t, int *out_idx)
{
    pmesh3 m;
    if (pmesh3_prepare(&m, mesh) != 0) return -1;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const v3 o = { origins[3 * i], origins[3 * i + 1], origins[3 * i + 2] }, d = { dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2] };
        float t; int idx;
        out_hit[i] = ray_closest3(&m, o, d, tmax[i], &t, &idx);
        out_t[i] = t; out_idx[i] = idx;
    }
    pmesh3_free(&m);
    return 0;
}

/* HarmonicGreenBall<3> (util/green.h:77-119): eval, norm, pdfRadius -- exposed for unit tests */
void wo3_green_ball(float R, float r, float *eval, float *norm, float *pdf_radius)
{
    *eval = (1.0f / r - 1.0f / R) / WO_4PI;
    *norm = R * R / 6.0f;
    *pdf_radius = 6.0f * r * (R - r) / (R * R * R);
}

/* ---- von Mises-Fisher distribution on the sphere (util/vmf.h:21-67, the Jakob [2012] form the reference uses; the lobes
 * of GuidedIntegrator<3>'s mixture).  Deterministic math like the rest of the oracle; log1p(x) is log(1 + x). -------- */
#define WO3_M_EPSILON 1e-5f   /* M_EPSILON, core/math/include/krrmath/constants.h */

/* VMF::eval(cosTheta) (vmf.h:27-32) */
float wo3_vmf_eval(float kappa, float cos_theta)
{
    if (kappa < WO3_M_EPSILON) return 1.0f / WO_4PI;
    return wo_expf(kappa * fminf(0.0f, cos_theta - 1.0f)) * kappa / (WO_2PI * (1.0f - wo_expf(-2.0f * kappa)));
}

/* VMF::sample(sampler, mu) (vmf.h:45-70): two draws; the lobe about +z turned into the frame of mu */
void wo3_vmf_sample(float kappa, const float mu[3], wo_pcg *rng, float out[3])
{
    const float u0 = wo_pcg_next_float(rng), u1 = wo_pcg_next_float(rng);
    float c, s;
    wo_sincos_2pi(u1, &c, &s);
    v3 local;
    if (kappa < WO3_M_EPSILON) {
        /* uniformSampleSphere<3> (util/sampling.h), as in the walk */
        const float z = 1 - 2 * u0, r = sqrtf(1 - z * z);
        local.x = r * c; local.y = r * s; local.z = z;
    } else {
        const float cos_theta = 1.0f + wo_logf(1.0f + (-u0 + wo_expf(-2.0f * kappa) * u0)) / kappa;
        const float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));      /* safe_sqrt */
        local.x = c * sin_theta; local.y = s * sin_theta; local.z = cos_theta;
    }
    const v3 m = { mu[0], mu[1], mu[2] };
    const v3 w = frame_to_world(m, local.x, local.y, local.z);
    out[0] = w.x; out[1] = w.y; out[2] = w.z;
}

int wo3_vmf_eval_batch(const float *kappa, const float *cos_theta, int n, float *pdf)
{
    for (int i = 0; i < n; ++i) pdf[i] = wo3_vmf_eval(kappa[i], cos_theta[i]);
    return 0;
}

/* per point a PCG32 stream setSeed(seed[i], 1) and per_point consecutive directions */
int wo3_vmf_sample_batch(const float *kappa, const float *mu, const uint64_t *seed, int n, int per_point, float *dirs)
{
    for (int i = 0; i < n; ++i) {
        wo_pcg rng;
        wo_pcg_set_seed(&rng, seed[i], 1);
        for (int k = 0; k < per_point; ++k) wo3_vmf_sample(kappa[i], mu + 3 * (size_t)i, &rng, dirs + 3 * ((size_t)i * per_point + k));
    }
    return 0;
}

/* ---- VMM<3,8>: the mixture of eight vMF lobes the 3-D guided integrator samples from (integrator/guided/distribution.h:
 * 279-436, train.h:60-105,492-553 with common3d: 5 numbers per lobe -- lambda, kappa, mean vector -- and the selection
 * logit: 41 network outputs).  Same conventions as the 2-D mixture (oracle/wost_vmm.c). ----------------------------- */
#define WV3_NCOMP 8
typedef struct { float lambda, kappa, mu[3], mo[3]; } wv3_lobe;
typedef struct { wv3_lobe sg[WV3_NCOMP]; float weight[WV3_NCOMP], total; } wv3_vmm;

static float wv3_clampf(float v, float lo, float hi) { return fmaxf(fminf(v, hi), lo); }

static void wv3_build(wv3_vmm *m, const float *data)
{
    m->total = 0.0f;
    for (int i = 0; i < WV3_NCOMP; ++i) {
        const float *d = data + 5 * i;
        wv3_lobe *g = &m->sg[i];
        g->lambda = wo_expf(wv3_clampf(d[0], -10.0f, 15.0f));      /* train.h:60-72, Exponential */
        g->kappa = wo_expf(wv3_clampf(d[1], -10.0f, 15.0f));
        /* Eigen normalized(): v / sqrt(z) when z = squaredNorm > 0, else v unchanged (see wost_vmm.c) */
        const float z = (d[2] * d[2] + d[3] * d[3]) + d[4] * d[4], n = sqrtf(z);
        for (int c = 0; c < 3; ++c) { g->mo[c] = d[2 + c]; g->mu[c] = z > 0.0f ? d[2 + c] / n : d[2 + c]; }
        m->total += g->lambda;
    }
    for (int i = 0; i < WV3_NCOMP; ++i) m->weight[i] = m->sg[i].lambda / m->total;
}

static float wv3_lobe_pdf(const wv3_lobe *g, const float w[3])
{
    return wo3_vmf_eval(g->kappa, (w[0] * g->mu[0] + w[1] * g->mu[1]) + w[2] * g->mu[2]);
}

static float wv3_pdf(const wv3_vmm *m, const float w[3])
{
    float val = 0.0f;
    for (int i = 0; i < WV3_NCOMP; ++i) val += m->weight[i] * wv3_lobe_pdf(&m->sg[i], w);
    return val;
}

/* VMM<3,N>::sample (:333-345): one draw picks the lobe, two more the direction */
static void wv3_sample(const wv3_vmm *m, wo_pcg *rng, float out[3])
{
    float u = wo_pcg_next_float(rng);
    for (int i = 0; i < WV3_NCOMP; ++i) {
        if (u < m->weight[i]) { wo3_vmf_sample(m->sg[i].kappa, m->sg[i].mu, rng, out); return; }
        u -= m->weight[i];
    }
    wo3_vmf_sample(m->sg[0].kappa, m->sg[0].mu, rng, out);
}

int wo3_vmm_pdf_sample(const float *raw, const float *wi, const uint64_t *seed, int n, float *pdf, float *dir)
{
    for (int i = 0; i < n; ++i) {
        wv3_vmm m;
        wv3_build(&m, raw + 40 * (size_t)i);
        if (pdf) pdf[i] = wv3_pdf(&m, wi + 3 * (size_t)i);
        if (dir) {
            wo_pcg rng;
            wo_pcg_set_seed(&rng, seed[i], 1);
            wv3_sample(&m, &rng, dir + 3 * (size_t)i);
        }
    }
    return 0;
}

/* compute_dL_doutput_divergence with GuidedOutput = common3d (train.h:492-553) around
 * VMM<3,N>::gradients_probability (distribution.h:348-421).  raw / dl_draw: 41 floats per sample. */
int wo3_vmm_loss_gradients(const float *raw, const float *dir, const float *li, const float *dir_pdf,
                           const unsigned char *on_neumann, const float *normal, int n, float loss_scale,
                           float *dl_draw, float *likelihood)
{
    const float eps = 1e-5f;                                    /* M_EPSILON */
    const float scale = loss_scale / (float)n;
    for (int t = 0; t < n; ++t) {
        const float *data = raw + 41 * (size_t)t;
        float *grad = dl_draw + 41 * (size_t)t;
        wv3_vmm m;
        wv3_build(&m, data);
        const float w[3] = { dir[3 * t], dir[3 * t + 1], dir[3 * t + 2] };
        const int on_n = on_neumann ? on_neumann[t] : 0;
        float r[3] = { 0.0f, 0.0f, 0.0f };
        if (on_n) {   /* reflect(wi, n) = wi - 2 (wi . n) n */
            const float *nn = normal + 3 * (size_t)t;
            const float d = (w[0] * nn[0] + w[1] * nn[1]) + w[2] * nn[2];
            for (int c = 0; c < 3; ++c) r[c] = w[c] - 2 * d * nn[c];
        }
        float pk[WV3_NCOMP], pkr[WV3_NCOMP];
        for (int k = 0; k < WV3_NCOMP; ++k) {
            pk[k] = wv3_lobe_pdf(&m.sg[k], w);
            pkr[k] = on_n ? wv3_lobe_pdf(&m.sg[k], r) : 0.0f;
        }
        float probability = 0.0f;
        for (int sg = 0; sg < WV3_NCOMP; ++sg) {
            const wv3_lobe *g = &m.sg[sg];
            const float lambda = g->lambda, kappa = g->kappa;
            const float ox = g->mo[0], oy = g->mo[1], oz = g->mo[2];
            const float vmf = pk[sg];
            probability += m.weight[sg] * vmf;
            float vmfr = 0.0f;
            if (on_n) { vmfr = pkr[sg]; probability += m.weight[sg] * vmfr; }
            float dF_dlambda = (vmf + vmfr) * (m.total - lambda) / (m.total * m.total);
            for (int k = 0; k < WV3_NCOMP; ++k) {
                if (k == sg) continue;
                dF_dlambda -= m.weight[k] / m.total * pk[k];
                if (on_n) dF_dlambda -= m.weight[k] / m.total * pkr[k];
            }
            /* 1/kappa - coth(kappa), below 1 by the reference's fitted parabola (:389-397) */
            float ik;
            if (kappa < 1) ik = 0.000962f + -0.344883f * kappa + 0.030147f * (kappa * kappa);
            else ik = 1 / kappa - (1 + wo_expf(-2 * kappa)) / (1 - wo_expf(-2 * kappa));
            float dF_dkappa = m.weight[sg] * vmf * ((w[0] * g->mu[0] + w[1] * g->mu[1] + w[2] * g->mu[2]) + ik);
            if (on_n) dF_dkappa += m.weight[sg] * vmfr * ((r[0] * g->mu[0] + r[1] * g->mu[1] + r[2] * g->mu[2]) + ik);
            const float n2 = (ox * ox + oy * oy) + oz * oz;
            float denom = n2 * sqrtf(n2);                       /* pow(., 1.5f) */
            if (denom < eps) denom = eps;
            const float x = w[0], y = w[1], z = w[2], xr = r[0], yr = r[1], zr = r[2];
            float dF_dx = m.weight[sg] * vmf * kappa * (-ox * oy * y - ox * oz * z + (oy * oy) * x + (oz * oz) * x) / denom;
            if (on_n) dF_dx += m.weight[sg] * vmfr * kappa * (-ox * oy * yr - ox * oz * zr + (oy * oy) * xr + (oz * oz) * xr) / denom;
            float dF_dy = m.weight[sg] * vmf * kappa * (-ox * oy * x - oy * oz * z + (ox * ox) * y + (oz * oz) * y) / denom;
            if (on_n) dF_dy += m.weight[sg] * vmfr * kappa * (-ox * oy * xr - oy * oz * zr + (ox * ox) * yr + (oz * oz) * yr) / denom;
            float dF_dz = m.weight[sg] * vmf * kappa * (-ox * oz * x - oy * oz * y + (ox * ox) * z + (oy * oy) * z) / denom;
            if (on_n) dF_dz += m.weight[sg] * vmfr * kappa * (-ox * oz * xr - oy * oz * yr + (ox * ox) * zr + (oy * oy) * zr) / denom;
            grad[5 * sg + 0] = dF_dlambda; grad[5 * sg + 1] = dF_dkappa;
            grad[5 * sg + 2] = dF_dx; grad[5 * sg + 3] = dF_dy; grad[5 * sg + 4] = dF_dz;
        }
        const float Li = li[t];
        const float dirPdf = dir_pdf[t] + eps;
        const float guidePdf = probability + eps;
        const float prefix = -Li / dirPdf / guidePdf * scale;
        if (likelihood) likelihood[t] = -Li / dirPdf * wo_logf(guidePdf);
        for (int sg = 0; sg < WV3_NCOMP; ++sg) {
            grad[5 * sg + 0] = prefix * grad[5 * sg + 0] * wo_expf(wv3_clampf(data[5 * sg + 0], -10.0f, 15.0f));
            grad[5 * sg + 1] = prefix * grad[5 * sg + 1] * wo_expf(wv3_clampf(data[5 * sg + 1], -10.0f, 15.0f));
            for (int c = 2; c < 5; ++c) grad[5 * sg + c] = prefix * grad[5 * sg + c];
        }
        /* selection probability (:541-552): uniformSampleSpherePDF<3> = 1/4pi, hemisphere 1/2pi; Logistic activation */
        const float e = 0.2f;
        const float uni = on_n ? 1.0f / WO_2PI : 1.0f / WO_4PI;
        const float sgm = 1.0f / (1.0f + wo_expf(-data[40]));
        grad[40] = scale * (-e) * Li * (guidePdf - uni) / (dirPdf * dirPdf) * (sgm * (1 - sgm));
    }
    return 0;
}


/* ==== GuidedIntegrator<3> (SURVEY 8a rows a21-a27 with DIM == 3) ==========================================================
 * The guided solve of oracle/wost_guided.c on triangle meshes: integrator/guided/integrator.cu with DIM == 3 (:153-249 separate,
 * :252-274 handleBoundary + records, :367-494 sampleNeumann with three draws, :497-526 routing, :529-563 inference, :618-668
 * training, :671-779 / :782-880 uniform / guided sampling with VMM<3,8> and the reflection about the Neumann normal, :883-965
 * oneStepWalk, :968-1094 the sample loop), guided/parameters.h:26-33 (3 inputs, 8 x 5 + 1 = 41 outputs padded to 48), train.h:149-155
 * (normalizeSpatialCoord with the diagonal of the 3-D box), :423-471 (training data), :492-553 (loss gradients, wo3_vmm_loss_gradients
 * above).  The same free choices as in 2-D make it deterministic: training set in (pixel, record) order, the fp32 network of
 * wost_net.c with three inputs.  The source term (sampleSource, guided/integrator.cu:277-364, templated on DIM) follows the uniform
 * 3-D step's restatement, its contribution recorded like a Neumann one (recordSourceContribution, guided.h:59-68). */
typedef struct {
    float sol[3];
    v3 p, d, n;
    float pdf, thp;
    int on_n;
} g3_record;

#define G3_MAX_TRAIN_DEPTH 4

typedef struct {
    wo_pcg rng;
    float sol[3];
    int state;            /* 0 = none, 1 = evaluation point queued, 2 = out of shell (has R_B) */
    v3 x, nn;
    float thp, R_B;
    int on_n;
    g3_record rec[G3_MAX_TRAIN_DEPTH + 1];
    unsigned cur_depth;
} g3_pixel;

typedef struct { float min[3], max[3]; } g3_aabb;

static int aabb3_contains(const g3_aabb *b, v3 q)
{
    return b->min[0] <= q.x && q.x <= b->max[0] && b->min[1] <= q.y && q.y <= b->max[1] && b->min[2] <= q.z && q.z <= b->max[2];
}

/* train.h:149-155: inflate by 0.5 % of the diagonal length, then 0.5 + (p - centre) / extent */
static void normalize_coord3(const g3_aabb *b, v3 q, float out[3])
{
    const float e[3] = { b->max[0] - b->min[0], b->max[1] - b->min[1], b->max[2] - b->min[2] };
    const float infl = sqrtf((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]) * 0.005f;      /* Eigen norm(): squared terms added in order */
    const float c[3] = { q.x, q.y, q.z };
    for (int a = 0; a < 3; ++a) {
        const float lo = b->min[a] - infl, hi = b->max[a] + infl;
        out[a] = 0.5f + (c[a] - (lo + hi) / 2.0f) / (hi - lo);
    }
}

static void record_solution3(g3_pixel *p, const float c[3], int inclusive)
{
    unsigned depth = p->cur_depth < G3_MAX_TRAIN_DEPTH ? p->cur_depth : G3_MAX_TRAIN_DEPTH;
    unsigned end = inclusive ? depth + 1 : depth;      /* guided.h:48-57 against :59-68 */
    for (unsigned i = 0; i < end; ++i)
        for (int ch = 0; ch < 3; ++ch) p->rec[i].sol[ch] = p->rec[i].sol[ch] + c[ch];
}

static void increment_depth3(g3_pixel *p, v3 dir, float pdf)
{
    unsigned d = p->cur_depth;
    if (d >= G3_MAX_TRAIN_DEPTH) return;
    g3_record *r = &p->rec[d];
    r->sol[0] = r->sol[1] = r->sol[2] = 0.0f;
    r->p = p->x; r->d = dir; r->pdf = pdf; r->thp = p->thp; r->on_n = p->on_n; r->n = p->nn;
    p->cur_depth = d + 1;
}

/* the shared tail of the three sampling kernels: intersect, advance, record */
static void advance_walker3(const pmesh3 *nm, g3_pixel *p, float eps, v3 dir, float pdf, float alpha, int record, uint64_t *nhits)
{
    v3 cur = p->x;
    if (p->on_n) cur = (v3){ p->x.x + eps * p->nn.x, p->x.y + eps * p->nn.y, p->x.z + eps * p->nn.z };
    v3 nxt = { p->x.x + p->R_B * dir.x, p->x.y + p->R_B * dir.y, p->x.z + p->R_B * dir.z };
    int hit = 0;
    v3 hn = { 0.0f, 0.0f, 0.0f };
    if (nm->n_tris > 0) {
        float t; int hi;
        hit = ray_closest3(nm, cur, dir, p->R_B, &t, &hi);
        if (hit) {
            hn = nm->tris[hi].n;
            if (dot3(hn, dir) > 0) { hn.x = -hn.x; hn.y = -hn.y; hn.z = -hn.z; }
            nxt = (v3){ cur.x + t * dir.x, cur.y + t * dir.y, cur.z + t * dir.z };
            if (nhits) __atomic_fetch_add(nhits, 1, __ATOMIC_RELAXED);
        }
    }
    if (record) increment_depth3(p, dir, pdf);      /* records the state BEFORE the step */
    p->thp = p->thp / pdf / alpha / WO_4PI;
    p->x = nxt; p->on_n = hit; p->nn = hn;
    p->state = 1;
}

/* uniformSampleSphere<3> / uniformSampleHemisphere<3> in the frame of the Neumann normal: the two draws of oneStepWalk */
static void uniform_direction3(g3_pixel *p, v3 *dir, float *pdf, float *alpha)
{
    const float u1 = wo_pcg_next_float(&p->rng), u2 = wo_pcg_next_float(&p->rng);
    float c, s;
    wo_sincos_2pi(u2, &c, &s);
    if (p->on_n) {
        const float z = u1, r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
        *dir = frame_to_world(p->nn, r * c, r * s, z);
        *pdf = 1.0f / WO_2PI;
        *alpha = 0.5f;
    } else {
        const float z = 1 - 2 * u1, r = sqrtf(1 - z * z);
        *dir = (v3){ r * c, r * s, z };
        *pdf = 1.0f / WO_4PI;
        *alpha = 1.0f;
    }
}

int wo3_solve_guided(const wo3_scene *sc, const wo3_guided_settings *gs, const wo_net_config *nc, float *params, int n_threads,
                     float *field_rgb, wo_guided_stats *stats, int dump_spp, wo3_train_dump *dump)
{
    if (!sc || !gs || !nc || !params || !field_rgb) return -1;
    if (nc->n_output != 41 || nc->n_output_padded < 41) return -1;
    const int W = gs->width, H = gs->height, N = W * H;
    pmesh3 dm, nm;
    if (pmesh3_prepare(&dm, &sc->dirichlet) != 0) return -2;
    if (pmesh3_prepare(&nm, &sc->neumann) != 0) { pmesh3_free(&dm); return -2; }
    const int has_d = dm.n_tris > 0, has_n = nm.n_tris > 0;
    const float eps = gs->eps_shell;
    g3_aabb box;
    for (int a = 0; a < 3; ++a) { box.min[a] = gs->aabb_min[a]; box.max[a] = gs->aabb_max[a]; }
    const uint64_t n_params = wo_net3_n_params(nc);
    const int NO = nc->n_output_padded;
    g3_pixel *px = calloc((size_t)N, sizeof(g3_pixel));
    float *inf_params = malloc(sizeof(float) * n_params);
    float *m1 = calloc(n_params, sizeof(float)), *m2 = calloc(n_params, sizeof(float)), *ema = calloc(n_params, sizeof(float));
    uint32_t *param_steps = calloc(n_params, sizeof(uint32_t));
    float *grad = malloc(sizeof(float) * n_params);
    float *net_in = malloc(sizeof(float) * 3 * (size_t)N);
    float *net_out = malloc(sizeof(float) * (size_t)NO * N);
    int *slot_of = malloc(sizeof(int) * (size_t)N);
    const size_t max_samples = (size_t)N * G3_MAX_TRAIN_DEPTH;
    float *t_x = malloc(sizeof(float) * 3 * max_samples), *t_dir = malloc(sizeof(float) * 3 * max_samples);
    float *t_li = malloc(sizeof(float) * max_samples), *t_pdf = malloc(sizeof(float) * max_samples);
    float *t_nrm = malloc(sizeof(float) * 3 * max_samples), *t_sol = malloc(sizeof(float) * 3 * max_samples);
    unsigned char *t_onn = malloc(max_samples);
    float *t_out = NULL, *t_dl = NULL;
    memcpy(inf_params, params, sizeof(float) * n_params);
    int opt_step = 0;
    uint64_t steps = 0, nhits = 0, guided_steps = 0, train_samples = 0, absorbed = 0, truncated = 0, started = 0;
    if (n_threads < 1) n_threads = 1;
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
            g3_pixel *q = &px[p];
            q->cur_depth = 0;
            q->state = 0;
            if (sc->mask && sc->mask[p] == 0) continue;
            q->x = eval_point3(sc, p % W, p / W, W, H);
            q->thp = 1.0f; q->on_n = 0; q->nn = (v3){ 0.0f, 0.0f, 0.0f };
            q->state = 1;
            started++;
        }
        for (int depth = 0; depth < gs->max_depth; ++depth) {
            const int guiding = depth < max_guided_depth;
            /* ---- separate + handleBoundary + sampleNeumann ---- */
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads) reduction(+ : steps, absorbed)
            for (int p = 0; p < N; ++p) {
                g3_pixel *q = &px[p];
                if (q->state != 1) continue;
                steps++;
                const int train_px = training && ((unsigned)(p - gs->train_pixel_offset) % (unsigned)gs->train_pixel_stride == 0);
                float R_D = INFINITY;
                if (has_d) {
                    const cp3 cp = closest_tri(&dm, q->x);
                    const ptri *T = &dm.tris[cp.idx];
                    const int side = tri_side(T, q->x);
                    float u, v;
                    tri_uv(T, q->x, &u, &v);
                    R_D = sqrtf(cp.d2);
                    if (R_D < eps && u > 0.0f && v > 0.0f && u + v < 1.0f) {
                        float col[3];
                        surface_color3(dm.colors, T, side, u, v, col);
                        for (int c = 0; c < 3; ++c) {
                            col[c] *= sc->dirichlet_intensity;
                            col[c] *= q->thp;
                            q->sol[c] = col[c] + q->sol[c];
                        }
                        if (train_px) record_solution3(q, col, 0);
                        q->state = 0;
                        absorbed++;
                        continue;
                    }
                }
                float R_N = INFINITY;
                if (has_n) R_N = closest_silhouette3(&nm, q->x, R_D);
                const float R_B = fmaxf(WO_R_B_FLOOR, fminf(R_D, R_N));      /* no 0.99 in the guided integrator (:238-239) */
                if (isinf(R_B)) { q->state = 0; continue; }
                q->R_B = R_B;
                q->state = 2;
                if (sc->source.nx > 0) {                        /* sampleSource (guided/integrator.cu:277-364 with DIM == 3) */
                    v3 sdir;
                    float dir_pdf, salpha = 1.0f;
                    {
                        const float u1 = wo_pcg_next_float(&q->rng), u2 = wo_pcg_next_float(&q->rng);
                        float c, s;
                        wo_sincos_2pi(u2, &c, &s);
                        if (q->on_n) {
                            const float z = u1, r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
                            sdir = frame_to_world(q->nn, r * c, r * s, z);
                            dir_pdf = 1.0f / WO_2PI;
                            salpha = 0.5f;
                        } else {
                            const float z = 1 - 2 * u1, r = sqrtf(1 - z * z);
                            sdir = (v3){ r * c, r * s, z };
                            dir_pdf = 1.0f / WO_4PI;
                        }
                    }
                    float dist = R_B;
                    if (has_n) {
                        float t; int hi;
                        const v3 o = { q->x.x + eps * sdir.x, q->x.y + eps * sdir.y, q->x.z + eps * sdir.z };
                        if (ray_closest3(&nm, o, sdir, dist, &t, &hi)) dist = fminf(t, dist);
                    }
                    const float g1 = wo_pcg_next_float(&q->rng), g2 = wo_pcg_next_float(&q->rng);
                    float gc, gs;
                    wo_sincos_2pi(g2, &gc, &gs);
                    float r = (1.0f + sqrtf(1.0f - cbrt01(g1 * g1)) * gc) * R_B / 2.0f;
                    r = fmaxf(1e-4f, r);
                    if (r > R_B) r = R_B / 2.0f;
                    if (r <= dist) {
                        float f[3], col[3];
                        wo3_source_eval(&sc->source, q->x.x + r * sdir.x, q->x.y + r * sdir.y, q->x.z + r * sdir.z, f);
                        const float norm = R_B * R_B / 6.0f;
                        const float c1 = (1.0f / WO_4PI) / (r * r), c2 = dir_pdf / (r * r);
                        for (int c = 0; c < 3; ++c) {
                            col[c] = q->thp * f[c] * norm * c1 / c2 / salpha;
                            q->sol[c] = col[c] + q->sol[c];
                        }
                        if (train_px) record_solution3(q, col, 1);
                    }
                }
                if (has_n) {                                    /* sampleNeumann: three draws in 3-D */
                    const float u0 = wo_pcg_next_float(&q->rng), u1 = wo_pcg_next_float(&q->rng), u2 = wo_pcg_next_float(&q->rng);
                    float pdf;
                    const int oi = sample_in_sphere3(&nm, q->x, R_B, u0, &pdf);
                    if (oi != -1 && pdf > 0) {
                        const ptri *S = &nm.tris[oi];
                        const float su = sqrtf(u1), b1 = u2 * su, b0 = 1.0f - su;
                        const float b2 = 1.0f - b0 - b1;
                        const v3 sp = { (S->p0.x * b0 + S->p1.x * b1) + S->p2.x * b2, (S->p0.y * b0 + S->p1.y * b1) + S->p2.y * b2,
                                        (S->p0.z * b0 + S->p1.z * b1) + S->p2.z * b2 };
                        const v3 rv = v3_sub(sp, q->x);
                        const float r = sqrtf(dot3(rv, rv));
                        if (r < R_B && r > 0) {
                            v3 o = q->x;
                            if (q->on_n) o = (v3){ q->x.x + eps * q->nn.x, q->x.y + eps * q->nn.y, q->x.z + eps * q->nn.z };
                            v3 rd = v3_sub(sp, o);
                            const float cd = sqrtf(dot3(rd, rd));
                            if (cd > 0) { rd.x /= cd; rd.y /= cd; rd.z /= cd; }
                            if (!ray_any3(&nm, o, rd, cd - eps)) {
                                int side = tri_side(S, q->x);
                                float uu, vv;
                                tri_uv(S, sp, &uu, &vv);
                                if (q->on_n) {
                                    const float dn = dot3(S->n, q->nn);
                                    side = (0.0f < dn) - (dn < 0.0f);
                                }
                                if (side != 0) {
                                    float col[3];
                                    surface_color3(nm.colors, S, side, uu, vv, col);
                                    const float alpha = q->on_n ? 0.5f : 1.0f;
                                    const float G = (1.0f / r - 1.0f / R_B) / WO_4PI;
                                    for (int c = 0; c < 3; ++c) {
                                        col[c] *= sc->neumann_intensity;
                                        col[c] *= q->thp * G / alpha / pdf;
                                        col[c] = -col[c];
                                        q->sol[c] = col[c] + q->sol[c];
                                    }
                                    if (train_px) record_solution3(q, col, 1);
                                }
                            }
                        }
                    }
                }
            }
            /* ---- inferenceStep ---- */
            int n_out = 0;
            for (int p = 0; p < N; ++p) {
                slot_of[p] = -1;
                if (px[p].state == 2) {
                    slot_of[p] = n_out;
                    normalize_coord3(&box, px[p].x, &net_in[3 * (size_t)n_out]);
                    n_out++;
                }
            }
            if (n_out == 0) break;
            if (guiding) {
                const int chunk = 256;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
                for (int b = 0; b < n_out; b += chunk) {
                    const int cnt = n_out - b < chunk ? n_out - b : chunk;
                    wo_net3_forward(nc, inf_params, net_in + 3 * (size_t)b, cnt, net_out + (size_t)NO * b, NULL);
                }
            }
            /* ---- handleOutShellPoint + guided / uniform sampling, or oneStepWalk ---- */
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads) reduction(+ : guided_steps)
            for (int p = 0; p < N; ++p) {
                g3_pixel *q = &px[p];
                if (q->state != 2) continue;
                const int train_px = training && ((unsigned)(p - gs->train_pixel_offset) % (unsigned)gs->train_pixel_stride == 0);
                const int record = train_px && depth < gs->max_train_depth;
                v3 dir;
                float pdf, alpha;
                if (!guiding) {
                    uniform_direction3(q, &dir, &pdf, &alpha);
                    advance_walker3(&nm, q, eps, dir, pdf, alpha, record, &nhits);
                    continue;
                }
                const float *raw = net_out + (size_t)NO * slot_of[p];
                const float sel = 1 / (1.f + wo_expf(-raw[40]));
                const int inside = aabb3_contains(&box, q->x);
                int to_guided = (uniform_fraction == 0) || (wo_pcg_next_float(&q->rng) < sel);
                to_guided = to_guided && inside;
                if (to_guided) {
                    if (!(uniform_fraction < 1.0f)) { q->state = 0; continue; }
                    wv3_vmm m;
                    wv3_build(&m, raw);
                    float w[3];
                    wv3_sample(&m, &q->rng, w);
                    float guided_pdf = wv3_pdf(&m, w);
                    float uniform_pdf = 1.0f / WO_4PI;
                    alpha = 1.0f;
                    if (q->on_n) {
                        uniform_pdf = 1.0f / WO_2PI;
                        alpha = 0.5f;
                        /* reflect(v, n) = v - 2 (v . n) n; Eigen dot(): products added in order */
                        const float dn = (w[0] * q->nn.x + w[1] * q->nn.y) + w[2] * q->nn.z;
                        const float r[3] = { w[0] - 2 * dn * q->nn.x, w[1] - 2 * dn * q->nn.y, w[2] - 2 * dn * q->nn.z };
                        if ((q->nn.x * w[0] + q->nn.y * w[1]) + q->nn.z * w[2] <= 0) { w[0] = r[0]; w[1] = r[1]; w[2] = r[2]; }
                        guided_pdf += wv3_pdf(&m, r);
                    }
                    dir = (v3){ w[0], w[1], w[2] };
                    pdf = sel * guided_pdf + (1.0f - sel) * uniform_pdf;
                    guided_steps++;
                } else {
                    uniform_direction3(q, &dir, &pdf, &alpha);
                    if (inside) {
                        wv3_vmm m;
                        wv3_build(&m, raw);
                        const float w[3] = { dir.x, dir.y, dir.z };
                        float guided_pdf = wv3_pdf(&m, w);
                        if (q->on_n) {
                            const float dn = (w[0] * q->nn.x + w[1] * q->nn.y) + w[2] * q->nn.z;
                            const float r[3] = { w[0] - 2 * dn * q->nn.x, w[1] - 2 * dn * q->nn.y, w[2] - 2 * dn * q->nn.z };
                            guided_pdf += wv3_pdf(&m, r);
                        }
                        pdf = sel * guided_pdf + (1.0f - sel) * pdf;
                    }
                }
                advance_walker3(&nm, q, eps, dir, pdf, alpha, record, &nhits);
            }
            if (depth == gs->max_depth - 1)
                for (int p = 0; p < N; ++p) truncated += px[p].state == 1;
        }
        /* ---- trainStep ---- */
        if (training) {
            size_t n = 0;
            for (unsigned p = (unsigned)gs->train_pixel_offset; p < (unsigned)N; p += (unsigned)gs->train_pixel_stride) {
                const g3_pixel *q = &px[p];
                for (unsigned k = 0; k < q->cur_depth; ++k) {
                    const g3_record *r = &q->rec[k];
                    if (!aabb3_contains(&box, r->p)) continue;
                    float s3[3];
                    for (int ch = 0; ch < 3; ++ch) {
                        float v = 0.0f;
                        if (fabsf(r->thp) > 1e-5f) v = r->sol[ch] / r->thp;
                        s3[ch] = fabsf(v);
                    }
                    float in3[3];
                    normalize_coord3(&box, r->p, in3);
                    if (isnan(in3[0]) || isnan(in3[1]) || isnan(in3[2]) || isnan(r->d.x) || isnan(r->d.y) || isnan(r->d.z) || isnan(r->pdf) ||
                        r->pdf == 0 || isnan(s3[0]) || isnan(s3[1]) || isnan(s3[2]))
                        continue;
                    memcpy(t_x + 3 * n, in3, sizeof(in3));
                    t_dir[3 * n] = r->d.x; t_dir[3 * n + 1] = r->d.y; t_dir[3 * n + 2] = r->d.z;
                    t_sol[3 * n] = s3[0]; t_sol[3 * n + 1] = s3[1]; t_sol[3 * n + 2] = s3[2];
                    t_li[n] = (s3[0] + s3[1] + s3[2]) / 3.0f;
                    t_pdf[n] = r->pdf;
                    t_onn[n] = (unsigned char)r->on_n;
                    t_nrm[3 * n] = r->n.x; t_nrm[3 * n + 1] = r->n.y; t_nrm[3 * n + 2] = r->n.z;
                    n++;
                }
            }
            train_samples += n;
            if (dump && sample == dump_spp) {
                dump->n = (int)n;
                const size_t m = n < (size_t)dump->capacity ? n : (size_t)dump->capacity;
                if (dump->xyz) memcpy(dump->xyz, t_x, sizeof(float) * 3 * m);
                if (dump->dir) memcpy(dump->dir, t_dir, sizeof(float) * 3 * m);
                if (dump->solution) memcpy(dump->solution, t_sol, sizeof(float) * 3 * m);
                if (dump->dir_pdf) memcpy(dump->dir_pdf, t_pdf, sizeof(float) * m);
                if (dump->on_neumann) memcpy(dump->on_neumann, t_onn, m);
                if (dump->normal) memcpy(dump->normal, t_nrm, sizeof(float) * 3 * m);
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
                    const int cnt = local - b < (size_t)chunk ? (int)(local - b) : chunk;
                    wo_net3_forward(nc, params, t_x + 3 * (o + b), cnt, t_out + (size_t)NO * b, NULL);
                }
                float *raw41 = malloc(sizeof(float) * 41 * local), *dl41 = malloc(sizeof(float) * 41 * local);
                for (size_t i = 0; i < local; ++i) memcpy(raw41 + 41 * i, t_out + (size_t)NO * i, sizeof(float) * 41);
                wo3_vmm_loss_gradients(raw41, t_dir + 3 * o, t_li + o, t_pdf + o, t_onn + o, t_nrm + 3 * o, (int)local, gs->loss_scale, dl41, NULL);
                memset(t_dl, 0, sizeof(float) * (size_t)NO * local);
                for (size_t i = 0; i < local; ++i) memcpy(t_dl + (size_t)NO * i, dl41 + 41 * i, sizeof(float) * 41);
                free(raw41); free(dl41);
                wo_net3_backward(nc, params, t_x + 3 * o, t_dl, (int)local, grad);
                opt_step++;
                wo_net3_optimizer_step(nc, params, m1, m2, ema, inf_params, grad, opt_step, gs->loss_scale, param_steps);
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
    free(t_x); free(t_dir); free(t_li); free(t_pdf); free(t_nrm); free(t_sol); free(t_onn); free(t_out); free(t_dl);
    pmesh3_free(&dm); pmesh3_free(&nm);
    return 0;
}
