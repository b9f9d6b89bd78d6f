// terrain_nonfinite.h -- per-pixel evaluation of the Numba engine's rule for +-Inf pixels (see terrain_nonfinite.hip).
// Host/device: the kernel of terrain_nonfinite.hip and the CPU harness of tests/hostsim instantiate the same function.
#pragma once
#include <string.h>

#include "terrain_math.h"

namespace xd {

struct NfParams {
    double w[5][25];  // zx, zy, zxx, zyy, zxy: table / divider in double, flipped to correlation order, zeros kept
    double sin_alt, cos_alt, az, zf;
    int M, fit, directional, degrees, hs_clip;
    uint32_t mask;
};
template <typename TOUT> struct NfPlanes { TOUT* p[10]; };

inline void nf_fill_params(NfParams& P, int surface_fit, int curv_directional, double resolution, double hs_alt, double hs_az,
                           double hs_z, int degrees, uint32_t surface_mask, int hs_clip = 1) {
    memset(&P, 0, sizeof P);
    fill_ref_weights(surface_fit, resolution, P.w);
    const double deg = 0.017453292519943295;  // np.deg2rad's factor
    P.sin_alt = sin(hs_alt * deg);
    P.cos_alt = cos(hs_alt * deg);
    P.az = (360.0 - hs_az) * deg;
    P.zf = hs_z;
    P.M = surface_fit == 2 ? 5 : 3;
    P.fit = surface_fit;
    P.directional = curv_directional;
    P.degrees = degrees;
    P.hs_clip = hs_clip;
    P.mask = surface_mask;
}

XD_HD double nf_pymod_2pi(double a) {  // Python's float `%` with a positive modulus (npy_divmod)
    const double b = 6.283185307179586;
    double m = fmod(a, b);
    if (m != 0.0) {
        if (m < 0.0) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

// One output pixel (r, c) of a row block with halo rows: nothing happens unless its window holds an infinite value and no NaN.
template <typename TIN, typename TOUT>
XD_HD void nf_pixel(const TIN* dem, int64_t r, int64_t c, int64_t H, int64_t W, int64_t stride, int64_t halo_top,
                    int64_t halo_bottom, const NfParams& P, const NfPlanes<TOUT>& out) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const int M = P.M, h = M / 2;
    const int64_t rows_total = halo_top + H + halo_bottom;
    const double qnan = __builtin_nan("");
    double z[25];
    bool any_inf = false, any_nan = false;
    for (int a = 0; a < M; ++a) {
        const int64_t rr = r + halo_top - h + a;
        for (int b = 0; b < M; ++b) {
            const int64_t cc = c - h + b;
            double v = qnan;  // outside the buffer = the NaN padding of surfit.py:1275-1282
            if (rr >= 0 && rr < rows_total && cc >= 0 && cc < W) v = (double)dem[rr * stride + cc];
            z[a * M + b] = v;
            any_inf |= (bool)isinf(v);
            any_nan |= (bool)isnan(v);
        }
    }
    if (!any_inf || any_nan) return;  // finite windows and NaN windows: both engines agree, the fused kernel's value stands
    double cf[5] = {0, 0, 0, 0, 0};
    const int ncoef = (P.mask & A_ANY_CURV) && P.fit != 0 ? 5 : 2;
    for (int k = 0; k < M * M; ++k)
        for (int d = 0; d < ncoef; ++d) {
            const double t = z[k] * P.w[d][k];
            cf[d] = cf[d] + t;
        }
    const double zx = cf[0], zy = cf[1], zxx = cf[2], zyy = cf[3], zxy = cf[4];
    const double g2 = zx * zx + zy * zy;
    const double opg = (1.0 + zx * zx) + zy * zy;
    const int64_t o = r * W + c;
    auto put = [&](int plane, double v, int post) {  // post: 0 none, 1 rad2deg (if degrees), 2 clip to [0, 255]
        TOUT t = (TOUT)v;
        if (post == 1 && P.degrees) t = t * DegScale<TOUT>::v();
        if (post == 2 && P.hs_clip) t = isnan((double)t) ? t : (t < (TOUT)0 ? (TOUT)0 : (t > (TOUT)255 ? (TOUT)255 : t));
        out.p[plane][o] = t;
    };
    double slope = 0.0, aspect = 0.0;
    if (P.mask & (A_SLOPE | A_HILLSHADE)) slope = atan(sqrt(g2));
    if (P.mask & (A_ASPECT | A_HILLSHADE)) aspect = nf_pymod_2pi(-atan2(-zx, zy));
    if (P.mask & A_SLOPE) put(P_SLOPE, slope, 1);
    if (P.mask & A_ASPECT) put(P_ASPECT, aspect, 1);
    if (P.mask & A_HILLSHADE) {
        const double smap = (P.zf != 1.0) ? atan(tan(slope) * P.zf) : slope;
        put(P_HILLSHADE, 1.5 + 254.0 * (P.sin_alt * cos(smap) + P.cos_alt * sin(smap) * sin(P.az - aspect)), 2);
    }
    if (P.fit == 0) return;
    if (P.mask & A_CURVATURE) put(P_CURVATURE, -2.0 * (zxx + zyy) * 100.0, 0);
    if (!(P.mask & (A_ANY_CURV & ~A_CURVATURE))) return;
    const bool flat = g2 == 0.0;
    const double num_prof = -(zxx * (zx * zx) + 2.0 * zxy * zx * zy + zyy * (zy * zy));
    const double num_tan = -(zxx * (zy * zy) - 2.0 * zxy * zx * zy + zyy * (zx * zx));
    const double num_flow = zx * zy * (zxx - zyy) - zxy * (zx * zx - zy * zy);
    const double g6 = g2 * g2 * g2, o3 = opg * opg * opg;
    if (P.mask & A_PROFILE) put(P_PROFILE, (flat ? 0.0 : num_prof / (P.directional ? g2 : g2 * sqrt(o3))) * 100.0, 0);
    if (P.mask & A_TANGENTIAL) put(P_TANGENTIAL, (flat ? 0.0 : num_tan / (P.directional ? g2 : g2 * sqrt(opg))) * 100.0, 0);
    if (P.mask & A_PLANFORM) put(P_PLANFORM, ((g2 < 10e-15) ? 0.0 : num_tan / sqrt(g6)) * 100.0, 0);
    if (P.mask & A_FLOWLINE) {
        const double v = P.directional ? (flat ? 0.0 : num_flow / sqrt(g6)) : ((g2 < 10e-15) ? 0.0 : num_flow / (sqrt(g6) * sqrt(opg)));
        put(P_FLOWLINE, v * 100.0, 0);
    }
    if (P.mask & (A_MAXC | A_MINC)) {
        double vmax, vmin;
        if (P.directional) {
            const double half_tr = (zxx + zyy) / 2.0;
            const double dd = (zxx - zyy) / 2.0;
            const double rad = sqrt(dd * dd + zxy * zxy);
            vmax = flat ? 0.0 : -(half_tr - rad);
            vmin = flat ? 0.0 : -(half_tr + rad);
        } else {
            const double q = (1.0 + zy * zy) * zxx - 2.0 * zxy * zx * zy + (1.0 + zx * zx) * zyy;
            const double q_u = (1.0 + zy * zy) * zxx - 2.0 * zy * zx * zxy + (1.0 + zx * zx) * zyy;
            const double d = 2.0 * sqrt(o3);
            const double mean = flat ? 0.0 : -q / d;
            const double e = q_u / d;
            const double unsph = flat ? 0.0 : sqrt(e * e - (zxx * zyy - zxy * zxy) / (opg * opg));
            vmax = flat ? 0.0 : mean + unsph;
            vmin = flat ? 0.0 : mean - unsph;
        }
        if (P.mask & A_MAXC) put(P_MAXC, vmax * 100.0, 0);
        if (P.mask & A_MINC) put(P_MINC, vmin * 100.0, 0);
    }
}

}  // namespace xd
