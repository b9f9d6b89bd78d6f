// terrain_math.h -- per-column marching stencil + attribute math of the fused terrain kernel.
//
// Shared between the gfx950 kernel (terrain.hip) and the host-compiled numerics harness under
// tests/hostsim/ (g++), so that the window logic and the float64 formulas can be checked against the
// oracle on a machine without a GPU.  The harness is test infrastructure; the shipped library only
// contains the device instantiation.
//
// What is computed (reference recipe):
//   1. derivative estimates zx, zy, zxx, zyy, zxy = integer-weighted stencil sums over the 3x3 (Horn,
//      Zevenbergen-Thorne) or 5x5 (Florinsky) window, accumulated in float64 (exact for float32 input),
//      scaled by 1/divider, then ROUNDED TO THE INPUT DTYPE -- the reference's SciPy engine stores
//      scipy.ndimage.convolve's input-dtype result (xdem/spatialstats.py:2523-2525, 2575);
//      orientation is that of a true convolution: zx = (west - east), zy = (south - north);
//   2. every attribute from those five numbers in float64 (xdem/terrain/surfit.py:590-943), rounded once
//      to the output dtype; then rad->deg / clip in the output dtype (xdem/terrain/terrain.py:586-596);
//   3. NaN for every surface attribute iff the full window holds a non-finite value or leaves the raster
//      (surfit.py:1185-1192); TPI / TRI follow plain IEEE propagation like the reference's
//      generic_filter(cval=nan) callbacks (xdem/terrain/window.py:67-252).
//
// Each thread owns one raster column of a tile and marches down its rows keeping a rotating register
// window of per-row partial sums, so a pixel costs one new row of partials instead of a full 25-tap
// gather.  Transcendentals are avoided: with rw = 1/sqrt(1+g2) and rg = 1/sqrt(g2) the hillshade and all
// curvatures are algebraic, and slope / aspect reduce to one arcsine polynomial each on [0, sqrt(1/2)].
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define XD_HD __host__ __device__ __forceinline__
#else
#define XD_HD inline
#endif

namespace xd {

enum : uint32_t {
    A_SLOPE = 1u << 0, A_ASPECT = 1u << 1, A_HILLSHADE = 1u << 2, A_CURVATURE = 1u << 3, A_PROFILE = 1u << 4,
    A_TANGENTIAL = 1u << 5, A_PLANFORM = 1u << 6, A_FLOWLINE = 1u << 7, A_MAXC = 1u << 8, A_MINC = 1u << 9,
    A_TPI = 1u << 10, A_TRI = 1u << 11, A_ROUGH = 1u << 12,
    A_ANY_CURV = A_CURVATURE | A_PROFILE | A_TANGENTIAL | A_PLANFORM | A_FLOWLINE | A_MAXC | A_MINC,
    A_ANY_WIN = A_TPI | A_TRI | A_ROUGH
};
constexpr int N_ATTR = 13;
enum { P_SLOPE = 0, P_ASPECT, P_HILLSHADE, P_CURVATURE, P_PROFILE, P_TANGENTIAL, P_PLANFORM, P_FLOWLINE, P_MAXC,
       P_MINC, P_TPI, P_TRI, P_ROUGH };

struct TerrainParams {
    double s1;       // 1 / (c * res)        first derivatives  (c = 8 Horn, 2 ZT, 420 Florinsky)
    double sxx;      // 1 / (c * res^2)      zxx, zyy           (c = 1 ZT, 35 Florinsky)
    double sxy;      // 1 / (c * res^2)      zxy                (c = 4 ZT, 100 Florinsky)
    double hs_sin_alt;   // 254 * sin(altitude)                        (the hillshade's scale factor 254 is folded in)
    double hs_kx;        // 254 * -cos(altitude) * z_factor * cos(az'),   az' = deg2rad(360 - azimuth)
    double hs_ky;        // 254 *  cos(altitude) * z_factor * sin(az')
    double hs_zf2;       // z_factor^2
    uint32_t mask;
    int curv_directional;
    int tri_wilson;
    int degrees;
};

// ---- small float64 primitives without divisions ------------------------------------------------------
// Hardware seeds: v_rsq_f64 (device).  The host stand-in deliberately truncates the seed to ~26 bits so
// that the harness exercises the Newton step the way the GPU does.
XD_HD double rsq_seed(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(x);
#else
    double y = 1.0 / sqrt(x);
    uint64_t b;
    __builtin_memcpy(&b, &y, 8);
    b &= ~((uint64_t(1) << 26) - 1);
    __builtin_memcpy(&y, &b, 8);
    return y;
#endif
}

// 1/sqrt(x) for finite x > 0 (one Newton step on the hardware seed: ~2^-50).
XD_HD double rsqrt_pos(double x) {
    double y = rsq_seed(x);
    double e = fma(-x * y, y, 1.0);
    return fma(0.5 * y, e, y);
}

// sqrt(x) for x >= 0 or NaN, IEEE results at the ends without selects: the seed is taken at x + 1e-300 (exact no-op for
// x > 1e-284) so x = 0 gives 0 * 1e150 = 0, a negative x keeps a NaN seed, and x = +inf (seed 0 -> NaN) is restored by the
// final maxNum with x * 1e-160 (<= sqrt(x) for every finite x, +inf for +inf, NaN for NaN).
XD_HD double sqrt_nr(double x) {
    const double y = rsq_seed(x + 1e-300);
    double g = x * y;
    const double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    return fmax(g, x * 1e-160);
}
// same for a radicand that may be slightly negative (-> NaN like the reference's `** 0.5`) and is never +inf
XD_HD double sqrt_nr_signed(double x) {
    const double y = rsq_seed(x + 1e-300);
    double g = x * y;
    const double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    return fma(g, r, g);
}

// p * s + c with the loop-invariant constant c kept in a scalar register pair: one v_fma_f64.  (Left to itself
// hipcc parks such constants in VGPRs and emits v_mov_b64 + v_fmac_f64 per Horner step.)
XD_HD double fma_c(double p, double s, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(s), "s"(c));
    return r;
#else
    return fma(p, s, c);
#endif
}

// a * b + c as ONE VOP3 v_fma_f64 with every operand in a VGPR pair.  (hipcc prefers the VOP2 v_fmac_f64 and, when `c` stays
// live, pays a v_mov_b64 copy in front of it.)
XD_HD double fma_v(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return fma(a, b, c);
#endif
}
// a * k + c with the loop-invariant multiplier k in a scalar register pair
XD_HD double fma_ks(double a, double k, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
    return r;
#else
    return fma(a, k, c);
#endif
}
// 2 * a + c (the 2.0 is an inline constant of the instruction)
XD_HD double fma_2(double a, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, 2.0, %2" : "=v"(r) : "v"(a), "v"(c));
    return r;
#else
    return fma(2.0, a, c);
#endif
}

// asin(x) for 0 <= x <= 0.7075: x * P(x^2), degree-12 near-minimax fit of asin(sqrt(s))/sqrt(s) on
// [0, 0.5005] (max relative error 1.4e-12, fitted with mpmath at Chebyshev nodes).
XD_HD double asin_small(double x) {
    const double s = x * x;
    double p = 0.24356914533777926376;
    p = fma_c(p, s, -0.52512032962871855258);
    p = fma_c(p, s, 0.56591394039083322551);
    p = fma_c(p, s, -0.33592284970511680326);
    p = fma_c(p, s, 0.14996103080076613321);
    p = fma_c(p, s, -0.023070071561117294953);
    p = fma_c(p, s, 0.024014351127563767045);
    p = fma_c(p, s, 0.021579572865057994895);
    p = fma_c(p, s, 0.030441879262631139926);
    p = fma_c(p, s, 0.044640181228123456917);
    p = fma_c(p, s, 0.075000061658013184589);
    p = fma_c(p, s, 0.16666666611178785838);
    p = fma_c(p, s, 1.0000000000008227697);
    return x * p;
}

template <typename T> struct DegScale;
template <> struct DegScale<float> { static XD_HD float v() { return 57.295776f; } };  // 180.0f / float(pi) in float arithmetic, as np.rad2deg
template <> struct DegScale<double> { static XD_HD double v() { return 57.29577951308232; } };

// Output sink: one pointer per attribute plane (null when not requested).  Pixels are addressed by a 32-bit BYTE
// offset from the (wave-uniform) plane pointer, which maps to the scalar-base + VGPR-offset store form.
template <typename TOUT> struct Planes { TOUT* p[N_ATTR]; };
template <typename TOUT> XD_HD void put(TOUT* plane, uint32_t byte_off, TOUT v) {
    *reinterpret_cast<TOUT*>(reinterpret_cast<char*>(plane) + byte_off) = v;
}

// Compile-time specialisation knobs.  CMASK != 0 fixes the attribute mask: every `if (mask & ...)` folds away
// and the independent attribute chains land in ONE basic block the scheduler can interleave; DIR / DEG /
// WILSON / ZF1 = -1 mean "read the runtime flag".  Spec<0,-1,-1,-1,-1> is the fully general kernel.
template <uint32_t CMASK_, int DIR_, int DEG_, int WILSON_, int ZF1_ = -1> struct Spec {
    static constexpr uint32_t CMASK = CMASK_;
    static constexpr int DIR = DIR_, DEG = DEG_, WILSON = WILSON_, ZF1 = ZF1_;  // ZF1: hillshade z_factor == 1
};
typedef Spec<0, -1, -1, -1, -1> SpecRuntime;
constexpr uint32_t MASK_FULL11 = 0xFFFu & ~A_CURVATURE;  // (bits 0-11)  // the 11-attribute headline set (no deprecated 'curvature')
constexpr uint32_t MASK_SAH_WIN = A_SLOPE | A_ASPECT | A_HILLSHADE | A_TPI | A_TRI;

// ---- attributes from the five derivative estimates ----------------------------------------------------
// Invalid windows arrive with zx (and zxx) already NaN ("poisoned" by the marcher), so NaN propagates through
// every formula below without per-attribute selects; comparisons with NaN are false, which keeps each
// special-case branch (flat, tiny, steep, quadrant) on its arithmetic path.
template <bool CURV, class SP, typename TOUT>
XD_HD void surface_pixel(double zx, double zy, double zxx, double zyy, double zxy, const TerrainParams& P,
                         const Planes<TOUT>& out, uint32_t o) {
    const uint32_t m = SP::CMASK ? SP::CMASK : P.mask;
    const bool deg = SP::DEG < 0 ? (P.degrees != 0) : (SP::DEG != 0);
    const double zx2 = zx * zx, zy2 = zy * zy;
    const double g2 = zx2 + zy2;
    const double opg = (1.0 + zx2) + zy2;
    const bool flat = (g2 == 0.0);
    const double rw = rsqrt_pos(opg);                 // cos(slope)
    const double rg = flat ? 0.0 : rsqrt_pos(g2);     // 1 / |grad|   (0 on flat ground: kills every x/g term)
    const double g = g2 * rg;                         // tan(slope)

    if (m & A_SLOPE) {
        // slope = atan(g): asin(g*rw) below 45 deg, pi/2 - asin(rw) above
        const bool steep = g2 > 1.0;
        double a = asin_small(steep ? rw : g * rw);
        a = steep ? (1.5707963267948966 - a) : a;
        TOUT v = (TOUT)a;
        if (deg) v = v * DegScale<TOUT>::v();
        put(out.p[P_SLOPE], o, (TOUT)(v));
    }
    if (m & A_ASPECT) {
        // aspect = atan2(zx, zy) mod 2pi, first-quadrant angle from the smaller normalised component
        const double ax = fabs(zx), ay = fabs(zy);
        const bool xbig = ax > ay;
        double a = asin_small(fmin(ax, ay) * rg);
        a = xbig ? (1.5707963267948966 - a) : a;
        a = (zy < 0.0) ? (3.141592653589793 - a) : a;
        a = (zx < 0.0) ? -a : a;
        a = (a < 0.0) ? (a + 6.283185307179586) : a;   // flat ground: rg = 0 -> a = 0 already
        TOUT v = (TOUT)a;
        if (deg) v = v * DegScale<TOUT>::v();
        put(out.p[P_ASPECT], o, (TOUT)(v));
    }
    if (m & A_HILLSHADE) {
        // 1.5 + 254 (sin(alt) cos(s') + cos(alt) sin(s') sin(az' - aspect)), s' = atan(zf * g), all algebraic
        double rwz = rw;  // z_factor 1 (the default): cos(s') = cos(slope)
        if (SP::ZF1 < 0 ? (P.hs_zf2 != 1.0) : (SP::ZF1 == 0)) rwz = rsqrt_pos(fma(P.hs_zf2, g2, 1.0));
        // 1.5 + 254 * shade; the factor 254 is folded into the three sun coefficients on the host (fill_params)
        TOUT v = (TOUT)fma_c(rwz, P.hs_sin_alt + fma(P.hs_ky, zy, P.hs_kx * zx), 1.5);
        v = v < (TOUT)0 ? (TOUT)0 : (v > (TOUT)255 ? (TOUT)255 : v);
        put(out.p[P_HILLSHADE], o, (TOUT)(v));
    }
    if (!CURV) return;
    if (m & A_CURVATURE) put(out.p[P_CURVATURE], o, (TOUT)((TOUT)(-2.0 * (zxx + zyy) * 100.0)));
    if (m & (A_ANY_CURV & ~A_CURVATURE)) {
        const bool dir = SP::DIR < 0 ? (P.curv_directional != 0) : (SP::DIR != 0);
        const double zxzy = zx * zy;
        const double cross = 2.0 * zxy * zxzy;
        const double n_prof = fma(zyy, zy2, fma(zxx, zx2, cross));       // zxx zx^2 + 2 zxy zx zy + zyy zy^2
        const double n_tan = fma(zyy, zx2, fma(zxx, zy2, -cross));        // zxx zy^2 - 2 zxy zx zy + zyy zx^2
        const double rg2 = rg * rg * 100.0;                               // (the x100 of every curvature folded in)
        const double rg_t = (g2 < 10e-15) ? 0.0 : rg;                     // planform / flowline zero below 1e-14
        if (m & A_PROFILE) {
            double v = -n_prof * rg2;
            if (!dir) v *= rw * rw * rw;
            put(out.p[P_PROFILE], o, (TOUT)(v));
        }
        const double t_dir = -n_tan * rg2;
        if (m & A_TANGENTIAL) put(out.p[P_TANGENTIAL], o, (TOUT)((dir ? t_dir : t_dir * rw)));
        if (m & A_PLANFORM) put(out.p[P_PLANFORM], o, (TOUT)((TOUT)(t_dir * rg_t)));
        if (m & A_FLOWLINE) {
            const double n_flow = fma(zxzy, zxx - zyy, -zxy * (zx2 - zy2));
            const double v = dir ? n_flow * rg2 * rg : n_flow * rg2 * rg_t * rw;
            put(out.p[P_FLOWLINE], o, (TOUT)(v));
        }
        if (m & (A_MAXC | A_MINC)) {
            double vmax, vmin;  // already x100
            if (dir) {
                const double half_tr = 50.0 * (zxx + zyy);
                const double hd = 50.0 * (zxx - zyy), sxy = 100.0 * zxy;
                const double rad = sqrt_nr_signed(fma(hd, hd, sxy * sxy));
                vmax = -(half_tr - rad);
                vmin = -(half_tr + rad);
            } else {
                // mean curvature H and unsphericity sqrt(H^2 - K); negative radicand -> NaN like the reference
                const double q = (zxx + zyy) + n_tan;
                const double rw2 = rw * rw * 100.0;
                const double mean = -0.5 * q * rw2 * rw;
                const double gauss = fma(zxx, zyy, -zxy * zxy) * rw2 * rw2;
                const double uns = sqrt_nr_signed(fma(mean, mean, -gauss));
                vmax = mean + uns;
                vmin = mean - uns;
            }
            if (m & A_MAXC) put(out.p[P_MAXC], o, (TOUT)(flat ? 0.0 : vmax));
            if (m & A_MINC) put(out.p[P_MINC], o, (TOUT)(flat ? 0.0 : vmin));
        }
    }
}

// TPI / TRI of a 3x3 window given as raw values (row-major n0..n8, n4 = centre).  Plain IEEE propagation.
template <class SP, typename TOUT>
XD_HD void window3_pixel(const double (&n)[9], double sum9, const TerrainParams& P, const Planes<TOUT>& out, uint32_t o) {
    const uint32_t m = SP::CMASK ? SP::CMASK : P.mask;
    const bool wilson = SP::WILSON < 0 ? (P.tri_wilson != 0) : (SP::WILSON != 0);
    const double c = n[4];
    if (m & A_TPI) put(out.p[P_TPI], o, (TOUT)((TOUT)fma_ks(sum9 - c, -0.125, c)));  // c - (sum9 - c) / 8, exact scaling
    if (m & A_ROUGH) {
        // Dartnell roughness: max - min of the window, NaN if any NaN (window.py:261-289); +-Inf propagate like NumPy
        double mx = n[0], mn = n[0];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            mx = (n[k] > mx) ? n[k] : mx;
            mn = (n[k] < mn) ? n[k] : mn;
        }
        const bool has_nan = (n[0] != n[0]) | (n[1] != n[1]) | (n[2] != n[2]) | (n[3] != n[3]) | (n[4] != n[4]) |
                             (n[5] != n[5]) | (n[6] != n[6]) | (n[7] != n[7]) | (n[8] != n[8]);
        put(out.p[P_ROUGH], o, has_nan ? (TOUT)NAN : (TOUT)(mx - mn));
    }
    if (m & A_TRI) {
        double acc = c - c;  // the centre's own term: 0, or NaN when the centre is +-Inf (IEEE, like the reference)
        if (wilson) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (k != 4) acc += fabs(n[k] - c);
            put(out.p[P_TRI], o, (TOUT)((TOUT)(acc * 0.125)));
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (k != 4) {
                    const double d = n[k] - c;
                    acc = fma(d, d, acc);
                }
            put(out.p[P_TRI], o, (TOUT)((TOUT)sqrt_nr(acc)));
        }
    }
}

template <typename TIN> XD_HD double round_in(double v) { return (double)(TIN)v; }

// ---- the column marcher -------------------------------------------------------------------------------
// `col` points at the tile element of this thread's column in the first tile row; tile row t holds raster
// row (first output row - HALO + t); element col[t * pitch + d] is the pixel d columns to the right.
// Emits n_out output rows; the plane pointers in `out` are already offset to the tile's first output pixel
// (wave-uniform, so stores use the scalar-base + 32-bit VGPR offset addressing form) and the BYTE offset of
// output row i is o0 + i * ostride (32-bit: a tile spans far less than 4 GiB per plane).
template <int FIT> struct Halo { static constexpr int v = (FIT == 2) ? 2 : 1; };

template <int FIT, bool CURV, bool WIN, class SP, typename TIN, typename TOUT>
XD_HD void march_column(const TIN* col, int pitch, int n_out, const TerrainParams& P, const Planes<TOUT>& out,
                        uint32_t o0, uint32_t ostride) {
    constexpr int HALO = Halo<FIT>::v;
    constexpr int NS = 2 * HALO + 1;  // rotating window slots
    const int nrows = n_out + 2 * HALO;
    const uint32_t m = SP::CMASK ? SP::CMASK : P.mask;

    // per-row partials (float64) -- Florinsky
    double A[NS], B[NS], R[NS], Wr[NS], Ua[NS], Ub[NS], D2[NS];  // D2 = A + 2 B (mixed derivative rows)
    // per-row partials -- 3x3 fits
    double Dr[NS], S[NS], Zc[NS];
    // 3-wide row sums and the float64 copies of the three centre columns for TPI / TRI (and the 3x3 detector)
    double R3[NS];
    double Nl[NS], Nc[NS], Nr[NS];

    for (int r0 = 0; r0 < nrows; r0 += NS) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int r = r0 + k;
            if (r < nrows) {
                const TIN* row = col + (int64_t)r * pitch;
                const TIN tl = row[-1], tc = row[0], tr = row[1];
                const double zl = (double)tl, zc = (double)tc, zr = (double)tr;
                if (FIT == 2) {
                    const double z0 = (double)row[-2], z4 = (double)row[2];
                    const double p = z0 + z4, q = zl + zr;
                    A[k] = zr - zl;
                    B[k] = z4 - z0;
                    if (CURV) D2[k] = fma_2(B[k], A[k]);
                    R[k] = (p + q) + zc;
                    if (CURV) Wr[k] = fma(2.0, p - zc, -q);
                    Ua[k] = fma(68.0, zc, fma(62.0, q, 44.0 * p));
                    Ub[k] = fma(17.0, zc, fma(5.0, q, -31.0 * p));
                    if (WIN) R3[k] = q + zc;
                } else {
                    Dr[k] = zr - zl;
                    S[k] = zl + zr;
                    Zc[k] = zc;
                    R3[k] = (zl + zr) + zc;
                }
                if (WIN) { Nl[k] = zl; Nc[k] = zc; Nr[k] = zr; }

                const int i = r - 2 * HALO;  // output row whose window is now complete
                if (i >= 0) {
                    uint32_t o = o0 + (uint32_t)i * ostride;  // byte offset
#if defined(__HIP_DEVICE_COMPILE__)
                    // keep `o` an opaque 32-bit VGPR: stops loop-strength-reduction from turning every plane
                    // into its own 64-bit running pointer (11 VGPR pairs + one 64-bit add per store)
                    asm volatile("" : "+v"(o));
#endif
                    // slot of window row (centre + d): the newest row (slot k) is centre + HALO
#define XD_SLOT(d) ((k + NS - HALO + (d)) % NS)
                    double zx, zy, zxx = 0.0, zyy = 0.0, zxy = 0.0;
                    if (FIT == 2) {
                        const int m2 = XD_SLOT(-2), m1 = XD_SLOT(-1), c0 = XD_SLOT(0), p1 = XD_SLOT(1), p2 = XD_SLOT(2);
                        const double sx = fma(17.0, B[c0], fma(68.0, A[c0],
                                          fma(5.0, B[m1] + B[p1], fma(62.0, A[m1] + A[p1],
                                          fma(-31.0, B[m2] + B[p2], 44.0 * (A[m2] + A[p2]))))));
                        zx = round_in<TIN>(-sx * P.s1);
                        zy = round_in<TIN>(((Ua[p1] - Ua[m1]) + (Ub[p2] - Ub[m2])) * P.s1);
                        double det;  // an all-25-pixel sum: non-finite <=> some pixel non-finite or outside the raster
                        if (CURV) {
                            det = ((Wr[m2] + Wr[m1]) + (Wr[c0] + Wr[p1])) + Wr[p2];
                            zxx = round_in<TIN>(det * P.sxx);
                            zyy = round_in<TIN>(fma(2.0, (R[m2] + R[p2]) - R[c0], -(R[m1] + R[p1])) * P.sxx);
                            zxy = round_in<TIN>(fma(2.0, D2[m2] - D2[p2], D2[m1] - D2[p1]) * P.sxy);
                        } else {
                            det = ((R[m2] + R[m1]) + (R[c0] + R[p1])) + R[p2];
                        }
                        const double poison = det - det;  // 0, or NaN for an invalid window
                        zx += poison;
                        if (CURV) zxx += poison;
                    } else {
                        const int m1 = XD_SLOT(-1), c0 = XD_SLOT(0), p1 = XD_SLOT(1);
                        const double det = (R3[m1] + R3[c0]) + R3[p1];
                        const double poison = det - det;
                        if (FIT == 0) {  // Horn: [1 2 1] smoothing across the derivative direction
                            zx = round_in<TIN>(-(fma(2.0, Dr[c0], Dr[m1] + Dr[p1])) * P.s1);
                            zy = round_in<TIN>((fma(2.0, Zc[p1] - Zc[m1], S[p1] - S[m1])) * P.s1);
                        } else {         // Zevenbergen-Thorne: central differences
                            zx = round_in<TIN>(-Dr[c0] * P.s1);
                            zy = round_in<TIN>((Zc[p1] - Zc[m1]) * P.s1);
                            if (CURV) {
                                zxx = round_in<TIN>(fma(-2.0, Zc[c0], S[c0]) * P.sxx) + poison;
                                zyy = round_in<TIN>(fma(-2.0, Zc[c0], Zc[m1] + Zc[p1]) * P.sxx);
                                zxy = round_in<TIN>((Dr[m1] - Dr[p1]) * P.sxy);
                            }
                        }
                        zx += poison;
                    }
                    if (m & ~A_ANY_WIN) surface_pixel<CURV, SP, TOUT>(zx, zy, zxx, zyy, zxy, P, out, o);
                    if (WIN) {
                        const int w1 = XD_SLOT(-1), w0 = XD_SLOT(0), w2 = XD_SLOT(1);
                        const double n[9] = {Nl[w1], Nc[w1], Nr[w1], Nl[w0], Nc[w0], Nr[w0], Nl[w2], Nc[w2], Nr[w2]};
                        window3_pixel<SP, TOUT>(n, (R3[w1] + R3[w0]) + R3[w2], P, out, o);
                    }
#undef XD_SLOT
                }
            }
        }
    }
}

}  // namespace xd
