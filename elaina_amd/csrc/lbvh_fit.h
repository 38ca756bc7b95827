// lbvh_fit.h -- the arithmetic of the 2-D tree's child records that the host builder (lbvh_build.cpp) and the device builder
// (wost_build2.hip) must agree on bit for bit.  Everything here is a chain of correctly rounded double / float operations
// (+ - * / sqrt, conversions, comparisons; compiled with -ffp-contract=off on both sides), on operands that do not depend on the
// order in which the primitives of a node are met: exact minima / maxima, and integer sums.  No libm call whose last bit could
// differ between glibc and the device library (the round-4 host builder used atan2 / cos / sin / hypot here).
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIP__)      // the HIP language (a .hip unit): lbvh_build.cpp goes through hipcc as plain C++
#define WOST_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define WOST_HD inline
#endif

namespace wost {

// the sixteen fixed candidate directions of an oriented box: (float)cos / (float)sin of pi a / 16, as glibc rounds them (constants,
// so that neither side computes them)
struct FitDir { float c, s; };
constexpr int kFitDirs = 16;
WOST_HD FitDir fit_dir(int a)
{
    constexpr FitDir t[kFitDirs] = {
        {0x1.000000p+0f, 0x0.0p+0f},          {0x1.f6297cp-1f, 0x1.8f8b84p-3f},  {0x1.d906bcp-1f, 0x1.87de2ap-2f},  {0x1.a9b662p-1f, 0x1.1c73b4p-1f},
        {0x1.6a09e6p-1f, 0x1.6a09e6p-1f},     {0x1.1c73b4p-1f, 0x1.a9b662p-1f},  {0x1.87de2ap-2f, 0x1.d906bcp-1f},  {0x1.8f8b84p-3f, 0x1.f6297cp-1f},
        {0x1.1a6264p-54f, 0x1.000000p+0f},    {-0x1.8f8b84p-3f, 0x1.f6297cp-1f}, {-0x1.87de2ap-2f, 0x1.d906bcp-1f}, {-0x1.1c73b4p-1f, 0x1.a9b662p-1f},
        {-0x1.6a09e6p-1f, 0x1.6a09e6p-1f},    {-0x1.a9b662p-1f, 0x1.1c73b4p-1f}, {-0x1.d906bcp-1f, 0x1.87de2ap-2f}, {-0x1.f6297cp-1f, 0x1.8f8b84p-3f}};
    return t[a];
}

// Moments of the end points of a node's segments on a 2^20 grid over the mesh's bounding box: integer sums, the same in any
// order and additive over the children of a node.  (2^40 per squared coordinate: 2^22 end points fit 62 bits.)
struct FitSums {
    long long n, sx, sy, sxx, sxy, syy;
};
WOST_HD double fit_grid_scale(float lox, float loy, float hix, float hiy)
{
    const double ex = (double)hix - (double)lox, ey = (double)hiy - (double)loy, e = ex > ey ? ex : ey;
    return e > 0.0 ? 1048576.0 / e : 0.0;
}
WOST_HD void fit_add_point(FitSums &s, float x, float y, float lox, float loy, double scale)
{
    const long long qx = (long long)(((double)x - (double)lox) * scale), qy = (long long)(((double)y - (double)loy) * scale);
    s.n += 1; s.sx += qx; s.sy += qy; s.sxx += qx * qx; s.sxy += qx * qy; s.syy += qy * qy;
}
WOST_HD void fit_add_sums(FitSums &s, const FitSums &o)
{
    s.n += o.n; s.sx += o.sx; s.sy += o.sy; s.sxx += o.sxx; s.sxy += o.sxy; s.syy += o.syy;
}

// the principal axis of those moments as the fp32 unit vector a box is stored with: theta = atan2(2 sxy, sxx - syy) / 2 without
// the arc tangent -- cos 2 theta = a / r, then the half-angle formulas (theta in (-pi/2, pi/2], its cosine is not negative)
WOST_HD void fit_pca_axis(const FitSums &s, float &ux, float &uy)
{
    const double n = (double)s.n;
    const double vxx = n * (double)s.sxx - (double)s.sx * (double)s.sx, vyy = n * (double)s.syy - (double)s.sy * (double)s.sy;
    const double a = vxx - vyy, b = 2.0 * (n * (double)s.sxy - (double)s.sx * (double)s.sy);
    const double r = sqrt(a * a + b * b);
    if (!(r > 0.0) || !(r < 1.0e300)) {
        ux = 1.0f; uy = 0.0f;
        return;
    }
    const double c2 = a / r;
    const double c = sqrt(0.5 * (1.0 + c2)), sn = sqrt(0.5 * (1.0 - c2));
    ux = (float)c;
    uy = (float)(b < 0.0 ? -sn : sn);
}

// extents of a point set along a stored axis and across it
struct FitExtent {
    double umin, umax, vmin, vmax;
};
WOST_HD FitExtent fit_extent_empty() { return FitExtent{1e300, -1e300, 1e300, -1e300}; }
WOST_HD void fit_extent_add(FitExtent &e, float uxf, float uyf, float x, float y)
{
    const double ux = uxf, uy = uyf, px = x, py = y;
    const double u = px * ux + py * uy, v = -px * uy + py * ux;
    e.umin = u < e.umin ? u : e.umin; e.umax = u > e.umax ? u : e.umax;
    e.vmin = v < e.vmin ? v : e.vmin; e.vmax = v > e.vmax ? v : e.vmax;
}
WOST_HD double fit_score(const FitExtent &e) { return (e.umax - e.umin) + (e.vmax - e.vmin); }
// the box record {cx cy ux uy hl hw} of the chosen axis: extents measured in the frame of the fp32 axis it is stored with,
// inflated by a relative 10^-6 plus `pad`
WOST_HD void fit_box(const FitExtent &e, float uxf, float uyf, double pad, float out[6])
{
    const double ux = uxf, uy = uyf, n2 = ux * ux + uy * uy;
    const double uc = 0.5 * (e.umin + e.umax), vc = 0.5 * (e.vmin + e.vmax);
    out[0] = (float)((uc * ux - vc * uy) / n2);
    out[1] = (float)((uc * uy + vc * ux) / n2);
    out[2] = uxf; out[3] = uyf;
    out[4] = (float)(0.5 * (e.umax - e.umin) * (1.0 + 1e-6) + pad);
    out[5] = (float)(0.5 * (e.vmax - e.vmin) * (1.0 + 1e-6) + pad);
}

// ---- SNCH normal cones ------------------------------------------------------------------------------------------------------
// The cone of a child: axis = the sum of the unit normals it must cover (2^-36 fixed point: an integer sum), half angle = the
// widest of them from that axis plus 10^-4 (cosine and sine by the addition theorem), radius = the farthest end point from the
// child's box centre.  cos(half) = -1 marks "cannot prune": an open polyline end below the child, no normal, normals that
// cancel, or a cone wider than a right angle less 10^-3.
constexpr double kNormalFix2 = 68719476736.0;      // 2^36
constexpr double kConePadCos2 = 0.999999995000000004166666665277778, kConePadSin2 = 9.99999998333333341666666646825397e-5;
struct ConeSums {
    long long sx, sy, cnt;
    int open;
};
WOST_HD void cone_add_normal(ConeSums &s, float nx, float ny)
{
    // llrint of a double that is far below 2^52: round to nearest even = add and subtract 2^52 * 1.5 (exact on both sides)
    const double big = 6755399441055744.0;
    const double ax = (double)nx * kNormalFix2, ay = (double)ny * kNormalFix2;
    s.sx += (long long)((ax + big) - big);
    s.sy += (long long)((ay + big) - big);
    s.cnt += 1;
}
WOST_HD void cone_add_sums(ConeSums &s, const ConeSums &o) { s.sx += o.sx; s.sy += o.sy; s.cnt += o.cnt; s.open |= o.open; }
// the unit axis, or false when the child cannot be pruned by its normals
WOST_HD bool cone_axis(const ConeSums &s, double &ax, double &ay)
{
    if (s.open || s.cnt <= 0) return false;
    ax = (double)s.sx / kNormalFix2;
    ay = (double)s.sy / kNormalFix2;
    const double al = sqrt(ax * ax + ay * ay);
    if (!(al > 1e-9 * (double)s.cnt)) return false;
    ax /= al;
    ay /= al;
    return true;
}
WOST_HD double cone_cos_to(double ax, double ay, float nx, float ny)
{
    const double x = nx, y = ny;
    return (ax * x + ay * y) / sqrt(x * x + y * y);
}
// out: ax ay cos(half) sin(half) radius; false: cannot prune (the caller writes the "cannot prune" record with the radius)
WOST_HD bool cone_finish(double ax, double ay, double cmin, float out[4])
{
    const double cc = cmin > 1.0 ? 1.0 : (cmin < -1.0 ? -1.0 : cmin);
    const double s2 = 1.0 - cc * cc, ss = sqrt(s2 > 0.0 ? s2 : 0.0);
    const double ch = cc * kConePadCos2 - ss * kConePadSin2, sh = ss * kConePadCos2 + cc * kConePadSin2;
    if (ch <= 1.0e-3) return false;
    out[0] = (float)ax; out[1] = (float)ay; out[2] = (float)ch; out[3] = (float)sh;
    return true;
}
WOST_HD float cone_radius(double rad, float ext)
{
    // padded by more than the silhouette test's absolute precision (10^-3): a query within that distance of a vertex takes the
    // test's near branch, which the cone argument does not cover -- it must count as inside
    return (float)(rad * (1.0 + 1e-6) + (double)ext * 0x1p-18 + 2.0e-3);
}

}  // namespace wost
