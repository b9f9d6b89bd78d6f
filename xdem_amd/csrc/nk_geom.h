// nk_geom.h -- geometry of a Nuth-Kaab plan's buffers and the bilinear sample of the shifted DEM (shared by the step kernels of
// nuthkaab.hip / nk_onepass.h and by the full-grid translation resample xdemhip_shift_bilinear).
#pragma once
#include <math.h>
#include <stdint.h>

#include "common.h"
#include "select.h"

namespace xd {

// Geometry of a plan's buffers.  Every per-pixel array (ref, tba, valid, slope_tan, aspect, dh) covers raster rows
// [roff, roff + nbuf) -- the whole raster, or a rank's row block plus its halo rows (multi-GPU) -- and is indexed by the
// LOCAL linear index q = (row - roff) * W + col.
struct NkGeom {
    int64_t H, W;      // raster shape (global)
    int64_t roff;      // raster row of buffer row 0
    double dr, dc;     // tap position = (row + dr, col + dc)
    int rule;          // NaN rule of the bilinear taps (context option "nk_nan_rule")
};

// ---- bilinear sample of tba at a shifted position ------------------------------------------------------------------
// geoutils' _interp_points (un-vendored, absent here) is restated as: bilinear, float64 weights, result rounded to the DEM
// dtype.  How nodata spreads is NOT pinned by anything in this image, so it is switchable (context option "nk_nan_rule"):
//   0 "4tap"      NaN if any of the four taps is non-finite or outside the raster, zero weights included (what
//                 scipy.ndimage.map_coordinates(order=1) does to NaN: 0 * NaN = NaN)                         [default]
//   1 "weighted"  taps with zero weight are ignored: at integer shifts the last row / column keep their values
//   2 "dilate3x3" NaN if any pixel of the 3 x 3 neighbourhood of the NEAREST pixel is non-finite or outside
//   3 "dilate_cross" the same with the 4-connected cross instead of the square (SciPy's default binary-dilation structure)
struct BiTap {
    int64_t q00;       // local index of the top-left tap; the others are q00 + dc1, q00 + drw, q00 + drw + dc1
    int64_t drw;       // W, or 0 where the lower row is ignored (rule 1, zero row weight)
    int dc1;           // 1, or 0 where the right column is ignored
    int64_t qn;        // nearest pixel (rule 2), -1 if its 3 x 3 neighbourhood leaves the raster
    double fr, fc;
    bool in;
};
// The tap position separates into a row part and a column part (kernels that walk down a column compute the latter once).
struct BiAxis { int64_t k0; int d1; double f; double pos; bool in; };
__device__ __forceinline__ BiAxis bi_axis(int64_t idx, double shift, int64_t extent, int rule) {
    BiAxis a;
    a.pos = t_add((double)idx, shift);
    const double k0f = floor(a.pos);
    a.f = t_sub(a.pos, k0f);
    a.k0 = (int64_t)k0f;
    // a node exactly on the upper edge needs no tap beyond it (any linear interpolator returns the node value there)
    a.d1 = (a.f == 0.0 && (rule == 1 || a.k0 + 1 >= extent)) ? 0 : 1;
    a.in = a.k0 >= 0 && a.k0 + a.d1 < extent;
    return a;
}
__device__ __forceinline__ BiTap bi_combine(const NkGeom& g, const BiAxis& r, const BiAxis& c) {
    BiTap t;
    t.fr = r.f;
    t.fc = c.f;
    t.in = r.in && c.in;
    t.q00 = t.in ? (r.k0 - g.roff) * g.W + c.k0 : 0;
    t.drw = t.in ? (int64_t)r.d1 * g.W : 0;
    t.dc1 = t.in ? c.d1 : 0;
    t.qn = -1;
    if (g.rule >= 2) {
        const int64_t rn = (int64_t)floor(r.pos + 0.5), cn = (int64_t)floor(c.pos + 0.5);
        if (rn >= 1 && cn >= 1 && rn + 1 < g.H && cn + 1 < g.W) t.qn = (rn - g.roff) * g.W + cn;
    }
    return t;
}
__device__ __forceinline__ BiTap bi_locate(const NkGeom& g, int64_t i, int64_t j) {
    return bi_combine(g, bi_axis(i, g.dr, g.H, g.rule), bi_axis(j, g.dc, g.W, g.rule));
}
template <typename T> struct BiVals { T a00, a01, a10, a11; };
template <typename T> __device__ __forceinline__ BiVals<T> bi_load(const T* __restrict__ img, const BiTap& t) {
    const T* q = img + t.q00;
    BiVals<T> v;
    v.a00 = q[0]; v.a01 = q[t.dc1]; v.a10 = q[t.drw]; v.a11 = q[t.drw + t.dc1];
    return v;
}
template <typename T>
__device__ __forceinline__ bool bi_value(const NkGeom& g, const T* __restrict__ img, const BiTap& t, T a00, T a01, T a10, T a11, T& out) {
    bool ok = t.in && t_finite(a00) && t_finite(a01) && t_finite(a10) && t_finite(a11);
    if (g.rule >= 2) {
        ok = ok && t.qn >= 0;
        if (ok)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                    if (g.rule == 2 || dy == 0 || dx == 0) ok = ok && t_finite(img[t.qn + dy * g.W + dx]);  // rule 3: the cross only
    }
    const double v00 = a00, v01 = a01, v10 = a10, v11 = a11;
    const double top = t_add(v00, t_mul(t.fc, t_sub(v01, v00)));
    const double bot = t_add(v10, t_mul(t.fc, t_sub(v11, v10)));
    out = (T)t_add(top, t_mul(t.fr, t_sub(bot, top)));
    return ok;
}
// row / column of a local linear index (W <= 2^31, q < 2^52: one float64 multiply and a correction step)
__device__ __forceinline__ void row_col(int64_t q, int64_t W, double invW, int64_t& li, int64_t& j) {
    li = (int64_t)((double)q * invW);
    j = q - li * W;
    if (j < 0) { --li; j += W; }
    else if (j >= W) { ++li; j -= W; }
}

}  // namespace xd
