// wost_walk.h -- the pieces of one walk step that the uniform round kernel (wost_hip.hip) and
// the guided wavefront kernels (wost_guided.hip) share: Neumann boundary sampling, the uniform
// star-region direction, and the boundary intersection that advances the walker.  Device code
// for gfx950, all __forceinline__: each kernel gets its own specialised copy.
#pragma once

#include "wost_device.h"

namespace wost {

// sampleNeumann (reference integrator/uniform/integrator.cu:336-444, guided/integrator.cu:
// 384-493): two draws from the pixel's stream whether or not the boundary emits (:343-347),
// object sampling in the ball, visibility, Green's-function weight.  Returns true when the
// sample contributes; (cr, cg, cb) is then the term to ADD to the solution (already negated).
template <bool EMISSIVE, bool TREE, class STK>
__device__ __forceinline__ bool neumann_sample(const DevMesh &nm, float neumann_intensity, float eps, float px, float py,
                                               float R_B, bool on_n, float nx, float ny, float thp, Pcg &rng,
                                               const STK &stk, float &cr, float &cg, float &cb)
{
    if (!EMISSIVE) {
        pcg_skip2(rng);
        return false;
    }
    const float u0 = pcg_next_float(rng);
    const float u1 = pcg_next_float(rng);
    float pdf;
    const int oi = (TREE && nm.obox_levels > 0) ? sample_in_sphere_tree(nm, px, py, R_B, u0, pdf) : sample_in_sphere_flat(nm, px, py, R_B, u0, pdf);
    if (!(oi != -1 && pdf > 0)) return false;
    const DevFlatSeg so = nm.flat[oi];
    const float spx = __builtin_fmaf(u1, so.ex, so.ax), spy = __builtin_fmaf(u1, so.ey, so.ay);
    const float rx = spx - px, ry = spy - py;
    const float r = sqrtf(dot2(rx, ry, rx, ry));
    if (!(r < R_B && r > 0)) return false;
    float ox = px, oy = py;
    if (on_n) { ox += eps * nx; oy += eps * ny; }
    float dx = spx - ox, dy = spy - oy;
    const float cd = sqrtf(dot2(dx, dy, dx, dy));
    if (cd > 0) { dx /= cd; dy /= cd; }
    if (ray_any<TREE>(nm, ox, oy, dx, dy, cd - eps, stk)) return false;
    const float crs = cross2(so.ex, so.ey, px - so.ax, py - so.ay);
    int side = (0.0f < crs) - (crs < 0.0f);
    const float uv = dot2(spx - so.ax, spy - so.ay, so.ex, so.ey) * so.inv_len2;
    if (on_n) {
        const float dn = dot2(so.nx, so.ny, nx, ny);
        side = (0.0f < dn) - (dn < 0.0f);
    }
    if (side == 0) return false;
    surface_color(nm.flatCol + 12 * (size_t)oi, side, uv, cr, cg, cb);
    const float alpha = on_n ? 0.5f : 1.0f;
    const float G = det_logf(R_B / r) / WOST_2PI;
    const float w = thp * G / alpha / pdf;
    cr *= neumann_intensity; cg *= neumann_intensity; cb *= neumann_intensity;
    cr *= w; cg *= w; cb *= w;
    cr = -cr; cg = -cg; cb = -cb;
    return true;
}

// bilinear sample of the source grid at a world point, times the intensity
__device__ __forceinline__ void source_eval(const DevSource &src, float x, float y, float &r, float &g, float &b)
{
    const float gx = __builtin_fmaf(x, src.sx, src.ox), gy = __builtin_fmaf(y, src.sy, src.oy);
    const float fx = floorf(gx), fy = floorf(gy);
    const float u = gx - fx, v = gy - fy;
    const int i = (int)fmaxf(fminf(fx, 1e9f), -1e9f), j = (int)fmaxf(fminf(fy, 1e9f), -1e9f);
    float t[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ii = i + (k & 1), jj = j + (k >> 1);
        const bool in = ii >= 0 && jj >= 0 && ii < src.nx && jj < src.ny;
        const float *p = src.rgb + 3 * ((size_t)(in ? jj : 0) * src.nx + (in ? ii : 0));
        t[k][0] = in ? p[0] : 0.0f; t[k][1] = in ? p[1] : 0.0f; t[k][2] = in ? p[2] : 0.0f;
    }
    float out[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = t[0][c] + (t[1][c] - t[0][c]) * u;
        const float bb = t[2][c] + (t[3][c] - t[2][c]) * u;
        out[c] = (a + (bb - a) * v) * src.intensity;
    }
    r = out[0]; g = out[1]; b = out[2];
}

__device__ __forceinline__ void uniform_direction(bool on_n, float nx, float ny, Pcg &rng, float &dirx, float &diry,
                                                  float &pdf, float &alpha);
template <bool TREE, class STK>
__device__ __forceinline__ bool ray_closest(const DevMesh &m, float ox, float oy, float dx, float dy, float tmax, float &t_out,
                                            int &idx_out, const STK &stk);

// sampleSource (reference integrator/uniform/integrator.cu:255-314, guided/integrator.cu:297-363):
// one direction draw, the straight line clipped by the Neumann boundary, a radius from the
// rejection sampler of HarmonicGreenBall<2> (util/green.h:44-73: two draws per trial, at most
// 1000 trials), the source value there.  Returns true and the term to ADD to the solution when
// the sampled point lies inside the star-shaped region.
template <bool TREE, class STK>
__device__ __forceinline__ bool source_sample(const DevSource &src, const DevMesh &nm, float eps, float px, float py, float R_B,
                                              bool on_n, float nx, float ny, float thp, Pcg &rng, const STK &stk, float &cr,
                                              float &cg, float &cb)
{
    float dirx, diry, dir_pdf, alpha;
    uniform_direction(on_n, nx, ny, rng, dirx, diry, dir_pdf, alpha);
    float dist = R_B;
    if (nm.n_segs > 0) {
        float t;
        int hi;
        if (ray_closest<TREE>(nm, px + eps * dirx, py + eps * diry, dirx, diry, dist, t, hi, stk)) dist = fminf(t, dist);
    }
    const float norm = R_B * R_B / 4.0f, bound = 1.5f / R_B;
    float r = 0.0f;
    for (int iter = 0; iter < 1000; ++iter) {
        const float u = pcg_next_float(rng);
        r = pcg_next_float(rng) * R_B;
        const float pdf = (det_logf(R_B / r) / WOST_2PI) / norm;        // r == 0 -> +inf: accepted
        const float pdf_radius = pdf / (1.0f / WOST_2PI);
        if (u < pdf_radius / bound) break;
    }
    r = fmaxf(1e-4f, r);                                                 // ELAINA_GREEN_FUNC_R_CLAMP
    if (r > R_B) r = R_B / 2.0f;
    if (!(r <= dist)) return false;
    float fr, fg, fb;
    source_eval(src, px + r * dirx, py + r * diry, fr, fg, fb);
    const float c1 = (1.0f / WOST_2PI) / r, c2 = dir_pdf / r;            // conditionalSampleSpherePDF<2>
    cr = thp * fr * norm * c1 / c2 / alpha;
    cg = thp * fg * norm * c1 / c2 / alpha;
    cb = thp * fb * norm * c1 / c2 / alpha;
    return true;
}

// uniformSampleSphere<2> / uniformSampleHemisphere<2> in the frame of the Neumann normal
// (util/sampling.h:29-33,80-85, util/transformation.h:30-55): one draw.
__device__ __forceinline__ void uniform_direction(bool on_n, float nx, float ny, Pcg &rng, float &dirx, float &diry,
                                                  float &pdf, float &alpha)
{
    const float u = pcg_next_float(rng);
    // ONE evaluation of the sine / cosine kernel for both kinds of lane (phi = pi u is the angle 2 pi (u / 2), and u / 2 is exact):
    // written as two calls in the two branches, a wave that holds walkers of both kinds ran the kernel twice
    float lc, ls;
    sincos_2pi(on_n ? u * 0.5f : u, lc, ls);
    if (on_n) {
        const float qx = -ny, qy = nx;              // frameFromNormal: T = -normalize(-n.y, n.x)
        const float ql = sqrtf(dot2(qx, qy, qx, qy));
        const float tx = -(qx / ql), ty = -(qy / ql);
        dirx = tx * lc + nx * ls;
        diry = ty * lc + ny * ls;
        pdf = (float)(1.0 / 3.14159265358979323846);
        alpha = 0.5f;
    } else {
        dirx = lc;
        diry = ls;
        pdf = 1.0f / WOST_2PI;
        alpha = 1.0f;
    }
}

// The tail of oneStepWalk (integrator/uniform/integrator.cu:479-513): move R_B along the
// direction, or to the first Neumann hit of the ray started at x (+ eps n on the boundary).
template <bool TREE, class STK>
__device__ __forceinline__ bool walk_advance(const DevMesh &nm, float eps, float px, float py, float R_B, bool on_n,
                                             float nx, float ny, float dirx, float diry, const STK &stk, float &nxt_x,
                                             float &nxt_y, float &hnx, float &hny)
{
    float cxp = px, cyp = py;
    if (on_n) {
        cxp += eps * nx;
        cyp += eps * ny;
    }
    nxt_x = px + R_B * dirx;
    nxt_y = py + R_B * diry;
    hnx = 0.0f;
    hny = 0.0f;
    bool hit = false;
    if (nm.n_segs > 0) {
        float t;
        int hi;
        hit = ray_closest<TREE>(nm, cxp, cyp, dirx, diry, R_B, t, hi, stk);
        if (hit) {
            hnx = nm.flat[hi].nx;
            hny = nm.flat[hi].ny;
            if (dot2(hnx, hny, dirx, diry) > 0) { hnx = -hnx; hny = -hny; }
            nxt_x = cxp + t * dirx;
            nxt_y = cyp + t * diry;
        }
    }
    return hit;
}

}  // namespace wost
