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

#if defined(__HIP_DEVICE_COMPILE__)
#define XD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define XD_SCHED_FENCE() ((void)0)
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
    int hs_clip = 1;   // 1: hillshade clipped to [0, 255] (the caller's post-step, terrain.py:594-596, fused; default), 0: as the ENGINE returns it
    // The reference's own convolution weights (integer table / divider in double, flipped to correlation order, row-major
    // over the window; |w| <= DBL_EPSILON -> 0 = skipped, as scipy.ndimage does) for zx, zy, zxx, zyy, zxy: used by the
    // rare exact-cancellation path of march_column (see ref_order_sum).
    double wref[5][25];
    int slot[N_ATTR];  // attribute -> index among the requested planes (staged row stores of terrain.hip)
};

// ---- the reference's divided convolution kernels (host side) -------------------------------------------
// xdem/terrain/surfit.py:65-304 (integer tables, generated here from their structure) divided in double by the resolution
// dividers of surfit.py:281-302 exactly as surfit.py:373-377 does (`table.astype(float64) / (c * res**p)`), then flipped
// on both axes because scipy.ndimage.convolve correlates with the flipped kernel (xdem/spatialstats.py:2521-2525).
inline void fill_ref_weights(int fit, double res, double (&w)[5][25]) {
    const int M = (fit == 2) ? 5 : 3;
    double tab[5][25] = {};
    double div[5] = {1, 1, 1, 1, 1};
    if (fit == 0) {  // Horn
        const int s3[3] = {1, 2, 1}, d3[3] = {-1, 0, 1};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                tab[0][a * 3 + b] = s3[a] * d3[b];
                tab[1][a * 3 + b] = -d3[a] * s3[b];
            }
        div[0] = div[1] = 8 * res;
    } else if (fit == 1) {  // Zevenbergen & Thorne
        const int d3[3] = {-1, 0, 1}, c3[3] = {1, -2, 1};
        for (int k = 0; k < 3; ++k) {
            tab[0][1 * 3 + k] = d3[k];
            tab[1][k * 3 + 1] = -d3[k];
            tab[2][1 * 3 + k] = c3[k];
            tab[3][k * 3 + 1] = c3[k];
        }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) tab[4][a * 3 + b] = -d3[a] * d3[b];
        div[0] = div[1] = 2 * res;
        div[2] = div[3] = pow(res, 2.0);
        div[4] = 4 * pow(res, 2.0);
    } else {  // Florinsky
        const int u5[5] = {-2, -1, 0, 1, 2}, c5[5] = {2, -1, -2, -1, 2}, al[5] = {44, 62, 68, 62, 44},
                  be[5] = {-31, 5, 17, 5, -31}, a5[5] = {0, -1, 0, 1, 0}, b5[5] = {-1, 0, 0, 0, 1};
        for (int a = 0; a < 5; ++a)
            for (int b = 0; b < 5; ++b) {
                tab[0][a * 5 + b] = al[a] * a5[b] + be[a] * b5[b];
                tab[1][a * 5 + b] = -(al[b] * a5[a] + be[b] * b5[a]);
                tab[2][a * 5 + b] = c5[b];
                tab[3][a * 5 + b] = c5[a];
                tab[4][a * 5 + b] = -u5[a] * u5[b];
            }
        div[0] = div[1] = 420 * res;
        div[2] = div[3] = 35 * pow(res, 2.0);
        div[4] = 100 * pow(res, 2.0);
    }
    for (int d = 0; d < 5; ++d)
        for (int a = 0; a < M; ++a)
            for (int b = 0; b < M; ++b) {
                const double v = tab[d][(M - 1 - a) * M + (M - 1 - b)] / div[d];
                w[d][a * M + b] = (fabs(v) > 2.220446049250313e-16) ? v : 0.0;
            }
}

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

// ---- float32 primitives of the mixed-precision tail (float32 DEM -> float32 attributes) -----------------
// Seeds: v_rsq_f32 / v_sqrt_f32 (1 ulp, device).  The host stand-ins are correctly rounded, i.e. at least as accurate.
XD_HD float rsq32_seed(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return (float)(1.0 / sqrt((double)x));
#endif
}
// 1/sqrt(x) for x in [1e-13, 1e16]: seed + one Newton step (~1 unit of 2^-24 relative)
XD_HD float rsq32(float x) {
    const float y = rsq32_seed(x);
    const float e = fmaf(-x * y, y, 1.0f);
    return fmaf(0.5f * y, e, y);
}
XD_HD float sqrt32_hw(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
XD_HD uint32_t f32_bits(float x) { uint32_t b; __builtin_memcpy(&b, &x, 4); return b; }
XD_HD float bits_f32(uint32_t b) { float x; __builtin_memcpy(&x, &b, 4); return x; }
// c ? a : b on float32 values, decided by x > y in float64.  (Written out for the device: hipcc's VOP2 form of
// v_cndmask_b32 -- the one that reads VCC -- issues at ~20 cycles on gfx950 (tools/ubench.hip), the VOP3 form with the
// mask in a scalar register pair at 4.)
XD_HD float select_gt(double x, double y, float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    uint64_t mk;
    asm("v_cmp_gt_f64_e64 %1, %2, %3\n\tv_cndmask_b32_e64 %0, %5, %4, %1" : "=v"(r), "=&s"(mk) : "v"(x), "v"(y), "v"(a), "v"(b));
    return r;
#else
    return (x > y) ? a : b;
#endif
}
// x > y ? a : b, all float32: compare into a scalar mask + VOP3 select (see select_gt)
XD_HD float select_gt32(float x, float y, float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    uint64_t mk;
    asm("v_cmp_gt_f32_e64 %1, %2, %3\n\tv_cndmask_b32_e64 %0, %5, %4, %1" : "=v"(r), "=&s"(mk) : "v"(x), "v"(y), "v"(a), "v"(b));
    return r;
#else
    return (x > y) ? a : b;
#endif
}
// |x| > |y| ? a : b (the absolute values are operand modifiers of the compare)
XD_HD float select_absgt32(float x, float y, float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    uint64_t mk;
    asm("v_cmp_gt_f32_e64 %1, |%2|, |%3|\n\tv_cndmask_b32_e64 %0, %5, %4, %1" : "=v"(r), "=&s"(mk) : "v"(x), "v"(y), "v"(a), "v"(b));
    return r;
#else
    return (fabsf(x) > fabsf(y)) ? a : b;
#endif
}
// clamp to [lo, hi] that keeps NaN (v_med3_f32 alone would return the smaller bound for a NaN input)
XD_HD float clamp_keep_nan(float v, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(v, lo, hi) + (v - v);
#else
    return (v < lo ? lo : (v > hi ? hi : v)) + (v - v);
#endif
}
// asin(x) for 0 <= x <= 0.7075 as x + x*s*Q(s), s = x^2: degree-6 Q fitted for the relative error of asin
// (tools/fit_poly.py: 1.7e-8 with these float32 coefficients, a quarter of a float32 rounding); the final fma keeps the
// evaluation error at one rounding.
XD_HD float asin32(float x) {
    const float s = x * x;
    float p = 1.115436330e-01f;
    p = fmaf(p, s, -9.298548102e-02f);
    p = fmaf(p, s, 7.720199972e-02f);
    p = fmaf(p, s, 1.631883346e-02f);
    p = fmaf(p, s, 4.651309177e-02f);
    p = fmaf(p, s, 7.488407940e-02f);
    p = fmaf(p, s, 1.666690707e-01f);
    return fmaf(x * s, p, x);
}

// ---- two float32 evaluations per instruction (v_pk_mul_f32 / v_pk_fma_f32) ---------------------------------------------------
// The fused kernel is bound by instruction ISSUE (one wave instruction per ~4 cycles and SIMD whatever its type: profiles/
// README.md, r03), so float32 work that comes in pairs -- the two arcsine polynomials, the two reciprocal square roots, the
// eight squared differences of the TRI -- is packed: same IEEE operations per lane (the host versions below ARE the
// definition), half the instructions.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float xd_v2f __attribute__((ext_vector_type(2)));
#endif
// (asin32(x1), asin32(x2))
XD_HD void asin32_pair(float x1, float x2, float& r1, float& r2) {
#if defined(__HIP_DEVICE_COMPILE__)
    const xd_v2f x = {x1, x2};
    const xd_v2f s = x * x;
    xd_v2f p = {1.115436330e-01f, 1.115436330e-01f};
    p = __builtin_elementwise_fma(p, s, (xd_v2f){-9.298548102e-02f, -9.298548102e-02f});
    p = __builtin_elementwise_fma(p, s, (xd_v2f){7.720199972e-02f, 7.720199972e-02f});
    p = __builtin_elementwise_fma(p, s, (xd_v2f){1.631883346e-02f, 1.631883346e-02f});
    p = __builtin_elementwise_fma(p, s, (xd_v2f){4.651309177e-02f, 4.651309177e-02f});
    p = __builtin_elementwise_fma(p, s, (xd_v2f){7.488407940e-02f, 7.488407940e-02f});
    p = __builtin_elementwise_fma(p, s, (xd_v2f){1.666690707e-01f, 1.666690707e-01f});
    const xd_v2f r = __builtin_elementwise_fma(x * s, p, x);
    r1 = r.x; r2 = r.y;
#else
    r1 = asin32(x1); r2 = asin32(x2);
#endif
}
// (rsq32(x1), rsq32(x2)): two hardware seeds, one packed Newton step
XD_HD void rsq32_pair(float x1, float x2, float& r1, float& r2) {
#if defined(__HIP_DEVICE_COMPILE__)
    const xd_v2f x = {x1, x2};
    const xd_v2f y = {__builtin_amdgcn_rsqf(x1), __builtin_amdgcn_rsqf(x2)};
    const xd_v2f e = __builtin_elementwise_fma(-x * y, y, (xd_v2f){1.0f, 1.0f});
    const xd_v2f r = __builtin_elementwise_fma(y * (xd_v2f){0.5f, 0.5f}, e, y);
    r1 = r.x; r2 = r.y;
#else
    r1 = rsq32(x1); r2 = rsq32(x2);
#endif
}
// acc0 + sum over the eight neighbours k != 4 of (n[k] - c)^2, as two interleaved partial sums (even / odd neighbours in
// row-major order 0 1 2 3 5 6 7 8), added at the end
XD_HD float tri_sumsq8(const float (&n)[9], float c, float acc0) {
    // (round 3: a packed form of these eight terms needed eight v_mov to pair the window registers and packed float32 FMAs
    // cost two plain ones on gfx950 -- tools/ubench.hip -- so the plain form is the cheaper one; same two partial sums)
    float ax = acc0, ay = 0.0f;
    const int ev[4] = {0, 2, 5, 7}, od[4] = {1, 3, 6, 8};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float d0 = n[ev[k]] - c, d1 = n[od[k]] - c;
        ax = fmaf(d0, d0, ax);
        ay = fmaf(d1, d1, ay);
    }
    return ax + ay;
}

template <typename T> struct DegScale;
template <> struct DegScale<float> { static XD_HD float v() { return 57.295776f; } };  // 180.0f / float(pi) in float arithmetic, as np.rad2deg
template <> struct DegScale<double> { static XD_HD double v() { return 57.29577951308232; } };

// Output sinks.  Planes = one pointer per attribute plane (null when not requested).  DirectSink stores every value
// straight to its plane: pixels are addressed by a 32-bit BYTE offset from the (wave-uniform) plane pointer at the tile
// origin, which maps to the scalar-base + VGPR-offset store form; the offset of output row i is o0 + i * ostride (a tile
// spans far less than 4 GiB per plane).  (terrain.hip adds a sink that stages rows in LDS for 1 KiB row stores.)
template <typename TOUT> struct Planes { TOUT* p[N_ATTR]; };
// PTR_COPY: copy the plane pointer through a scalar register inside the store's asm text (see put<>); kernels whose plane
// pointers are never spilled to VGPR lanes (no v_readlane in their code: tests/test_cabi_and_host.py checks the ISA of the
// streaming kernels) can do without: 11 scalar instructions less per output row of an issue-bound kernel.
template <typename TOUT, bool PTR_COPY = true> struct DirectSink {
    typedef TOUT out_t;
    Planes<TOUT> org;
    uint32_t o0, ostride, o;
    uint32_t sync_n = 0;  // workgroup barrier after every sync_n-th output row (a power of two; 0 = never): option "terrain_sync"
    XD_HD void begin_row(int i) {
        // byte offset of output row i: o0 + i * ostride with the product formed on the SCALAR unit (i and ostride are
        // wave-uniform) -- one v_add_u32 per row.  (Left to itself hipcc forms a quarter-rate v_mad_u64_u32 per row, or, for the
        // running-offset form `i == 0 ? o0 : o + ostride`, a VOP2 v_cndmask on VCC: ~20 cycles on gfx950, tools/ubench.hip.)
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t rowoff = (uint32_t)__builtin_amdgcn_readfirstlane(i) * (uint32_t)__builtin_amdgcn_readfirstlane((int)ostride);
        o = o0 + rowoff;
        // keep `o` an opaque 32-bit VGPR: stops loop-strength-reduction from turning every plane into its own 64-bit
        // running pointer (11 VGPR pairs + one 64-bit add per store)
        asm volatile("" : "+v"(o));
#else
        o = o0 + (uint32_t)i * ostride;
#endif
    }
    template <int K> XD_HD void put(TOUT v) {
#if defined(XD_NOSTORE)  // (measurement builds: all the math, no output traffic)
        if (v != (TOUT)12345.678)  return;
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(XD_PLAINSTORE)
        // Streaming, write-through plane stores (`global_store_dword ... nt sc1`; XD_STORE_BITS overrides the bits in
        // measurement builds): the planes are written once and never read back by this kernel.  Measured at 40000^2 in one
        // session: plain 15.76 ms, `nt` 15.29, `nt sc1` 15.02, `nt sc0 sc1` 15.05, `sc0 sc1` 15.38.
#ifndef XD_STORE_BITS
#define XD_STORE_BITS "nt sc1"
#endif
        if constexpr (sizeof(TOUT) == 4) {
            // The plane pointer goes through a scalar copy inside the asm: a pointer the compiler has just restored from a spill
            // with v_readlane (VALU write of an SGPR) may not be read by a VMEM instruction for 5 wait states -- the compiler
            // inserts them for its own instructions, not for inline asm, and the runtime-mask kernels faulted without them.
            // A SALU read of a VALU-written SGPR and a VMEM read of a SALU-written SGPR need none; `s_nop 4` instead of the
            // copy (XD_STORE_NOP) measured 1.2 % slower (14.50 vs 14.33 ms).
#if defined(XD_STORE_NOP)
            asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2 " XD_STORE_BITS ::"v"(o), "v"(v), "s"(org.p[K]) : "memory");
#else
            if constexpr (PTR_COPY) {
                uint64_t ptmp;
                asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dword %1, %2, %0 " XD_STORE_BITS : "=&s"(ptmp) : "v"(o), "v"(v), "s"(org.p[K]) : "memory");
            } else {
                asm volatile("global_store_dword %0, %1, %2 " XD_STORE_BITS ::"v"(o), "v"(v), "s"(org.p[K]) : "memory");
            }
#endif
        } else {
            __builtin_nontemporal_store(v, reinterpret_cast<TOUT*>(reinterpret_cast<char*>(org.p[K]) + o));
        }
#else
        *reinterpret_cast<TOUT*>(reinterpret_cast<char*>(org.p[K]) + o) = v;
#endif
    }
    XD_HD void end_row(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
        // keeps the four waves of the workgroup on the same raster row: their four 256-byte segments of a plane's 1 KiB row
        // piece then reach the memory controller together (every thread of the workgroup marches, see terrain_tile_kernel)
        if (sync_n && ((uint32_t)(i + 1) & (sync_n - 1)) == 0) __builtin_amdgcn_s_barrier();
#else
        (void)i;
#endif
    }
};

// Compile-time specialisation knobs.  CMASK != 0 fixes the attribute mask: every `if (mask & ...)` folds away
// and the independent attribute chains land in ONE basic block the scheduler can interleave; DIR / DEG /
// WILSON / ZF1 = -1 mean "read the runtime flag".  Spec<0,-1,-1,-1,-1> is the fully general kernel.
template <uint32_t CMASK_, int DIR_, int DEG_, int WILSON_, int ZF1_ = -1, int F64TAIL_ = 0> struct Spec {
    static constexpr uint32_t CMASK = CMASK_;
    static constexpr int DIR = DIR_, DEG = DEG_, WILSON = WILSON_, ZF1 = ZF1_;  // ZF1: hillshade z_factor == 1
    static constexpr int F64TAIL = F64TAIL_;  // float32 in / float32 out: 0 mixed tail (round 2), 1 float64 attribute math, 2 lean tail (round 3)
};
typedef Spec<0, -1, -1, -1, -1> SpecRuntime;
typedef Spec<0, -1, -1, -1, -1, 1> SpecRuntimeF64;
constexpr uint32_t MASK_FULL11 = 0xFFFu & ~A_CURVATURE;  // (bits 0-11)  // the 11-attribute headline set (no deprecated 'curvature')
constexpr uint32_t MASK_SAH_WIN = A_SLOPE | A_ASPECT | A_HILLSHADE | A_TPI | A_TRI;

// ---- attributes from the five derivative estimates ----------------------------------------------------
// Invalid windows arrive with zx (and zxx) already NaN ("poisoned" by the marcher), so NaN propagates through
// every formula below without per-attribute selects; comparisons with NaN are false, which keeps each
// special-case branch (flat, tiny, steep, quadrant) on its arithmetic path.
template <bool CURV, class SP, class SINK>
XD_HD void surface_pixel(double zx, double zy, double zxx, double zyy, double zxy, const TerrainParams& P, SINK& sk) {
    typedef typename SINK::out_t TOUT;
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
        sk.template put<P_SLOPE>((TOUT)(v));
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
        sk.template put<P_ASPECT>((TOUT)(v));
    }
    if (m & A_HILLSHADE) {
        // 1.5 + 254 (sin(alt) cos(s') + cos(alt) sin(s') sin(az' - aspect)), s' = atan(zf * g), all algebraic
        double rwz = rw;  // z_factor 1 (the default): cos(s') = cos(slope)
        if (SP::ZF1 < 0 ? (P.hs_zf2 != 1.0) : (SP::ZF1 == 0)) rwz = rsqrt_pos(fma(P.hs_zf2, g2, 1.0));
        // 1.5 + 254 * shade; the factor 254 is folded into the three sun coefficients on the host (fill_params)
        TOUT v = (TOUT)fma_c(rwz, P.hs_sin_alt + fma(P.hs_ky, zy, P.hs_kx * zx), 1.5);
        if (P.hs_clip) v = v < (TOUT)0 ? (TOUT)0 : (v > (TOUT)255 ? (TOUT)255 : v);   // (only this float64-tail form knows the unclipped mode)
        sk.template put<P_HILLSHADE>((TOUT)(v));
    }
    if (!CURV) return;
    if (m & A_CURVATURE) sk.template put<P_CURVATURE>((TOUT)((TOUT)(-2.0 * (zxx + zyy) * 100.0)));
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
            sk.template put<P_PROFILE>((TOUT)(v));
        }
        const double t_dir = -n_tan * rg2;
        if (m & A_TANGENTIAL) sk.template put<P_TANGENTIAL>((TOUT)((dir ? t_dir : t_dir * rw)));
        if (m & A_PLANFORM) sk.template put<P_PLANFORM>((TOUT)((TOUT)(t_dir * rg_t)));
        if (m & A_FLOWLINE) {
            const double n_flow = fma(zxzy, zxx - zyy, -zxy * (zx2 - zy2));
            const double v = dir ? n_flow * rg2 * rg : n_flow * rg2 * rg_t * rw;
            sk.template put<P_FLOWLINE>((TOUT)(v));
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
            if (m & A_MAXC) sk.template put<P_MAXC>((TOUT)(flat ? 0.0 : vmax));
            if (m & A_MINC) sk.template put<P_MINC>((TOUT)(flat ? 0.0 : vmin));
        }
    }
}

// ---- mixed-precision tail: float32 DEM -> float32 attributes ---------------------------------------------
// The derivative estimates arrive as the float32 values the reference rounds them to.  Everything that can cancel or that
// is raised to a power stays in float64 -- squares, the 1/sqrt factors and their powers, the curvature numerators, the
// discriminant of max/min curvature and its root, the hillshade -- so those outputs stay (almost always) bit-identical to
// the reference's float64 evaluation.  What runs in float32 are the two arcsine polynomials of slope and aspect with the
// octant assembly (float64 arguments rounded once, result within ~2 ulp) and the TRI sums of window3_pixel_mixed, i.e. the
// parts whose float64 form costs the most cycles for digits the float32 output cannot hold.
// FULLRANGE = false is the hot path: valid for squared gradients in [1e-13, 1e16] (no flat-ground selects, nothing can
// under- or overflow).  FULLRANGE = true adds the reference's flat-ground rules (surfit.py:749-805, 825-937) and is what
// march_column's cold path runs for the pixels outside that range (exactly flat ground, the reference's 1e-15
// cancellation residues, absurd slopes: mixed_tail_out_of_range); it uses the very same constants as the hot path, so
// hoisting them out of the row loop costs no extra scalar registers.
template <bool CURV, class SP, bool FULLRANGE, class SINK>
XD_HD void surface_pixel_mixed(float zxf, float zyf, float zxxf, float zyyf, float zxyf, const TerrainParams& P, SINK& sk) {
    const uint32_t m = SP::CMASK ? SP::CMASK : P.mask;
    const bool deg = SP::DEG < 0 ? (P.degrees != 0) : (SP::DEG != 0);
    const bool zf_not_1 = SP::ZF1 < 0 ? (P.hs_zf2 != 1.0) : (SP::ZF1 == 0);
    const double zx = (double)zxf, zy = (double)zyf;
    const double zx2 = zx * zx, zy2 = zy * zy;
    const double g2 = zx2 + zy2;
    const double opg = 1.0 + g2;
    const bool flat = FULLRANGE && (g2 == 0.0);
#if defined(XD_RAW_RSQ)  // (measurement builds: hardware seeds without the Newton step, ~2^-26)
    const double rw = rsq_seed(opg);
    const double rg = flat ? 0.0 : rsq_seed(g2);
#else
    const double rw = rsqrt_pos(opg);                              // cos(slope)
    const double rg = flat ? 0.0 : rsqrt_pos(g2);                  // 1 / |grad|  (0 on flat ground: kills every x/g term)
#endif
    const float rgf = (float)rg;
    // the two arcsine arguments: min(sin, cos) of the slope; the smaller normalised gradient component of the aspect's octant
    const float ax = fabsf(zxf), ay = fabsf(zyf);
    float as_slope = 0.0f, as_aspect = 0.0f;
    {
        const float xs = (float)fmin((g2 * rg) * rw, rw), xa = fminf(ax, ay) * rgf;
        if ((m & A_SLOPE) && (m & A_ASPECT)) asin32_pair(xs, xa, as_slope, as_aspect);
        else if (m & A_SLOPE) as_slope = asin32(xs);
        else if (m & A_ASPECT) as_aspect = asin32(xa);
    }
    if (m & A_SLOPE) {
        // slope = atan(g) = asin(g*rw) below 45 deg, pi/2 - asin(rw) above
        const float a0 = as_slope;
        float a = select_gt(g2, 1.0, (1.57079637f - a0) + -4.37113883e-08f, a0);   // pi/2 in two float32 pieces
        if (deg) a = a * DegScale<float>::v();
        sk.template put<P_SLOPE>(a);
    }
    if (m & A_ASPECT) {
        // aspect = atan2(zx, zy) mod 2pi = n * pi/2 +- a0 with the octant's quarter-turn count n and a0 in [0, pi/4]:
        // octants in order (sx, sy, xbig) = 000 001 011 010 110 111 101 100 -> n = 0 1 1 2 2 3 3 4, a0 subtracted in the odd
        // ones.  Sign bits and integer ops instead of compares + selects (x + 0 turns -0 into +0: "< 0" semantics).
        const float a0 = as_aspect;
        const uint32_t sx = f32_bits(zxf + 0.0f) >> 31, sy = f32_bits(zyf + 0.0f) >> 31;
        const uint32_t xb = f32_bits(ay - ax) >> 31;   // |zx| > |zy|
        const uint32_t q = sx ^ sy;
        const uint32_t odd = q ^ xb;                    // octant index parity (Gray code -> binary, lowest bit)
        const uint32_t oct = 4u * sx + 2u * q + odd;
        const float nf = (float)(int)((oct + 1u) >> 1);
        const float a0s = bits_f32(f32_bits(a0) ^ (odd << 31));
        const float t = fmaf(nf, -4.37113883e-08f, a0s);
        float a = fmaf(nf, 1.57079637f, t);
        if (deg) a = a * DegScale<float>::v();
        sk.template put<P_ASPECT>(a);
    }
    if (m & A_HILLSHADE) {
        // 1.5 + 254 cos(s') (sin(alt) + cos(alt) zf (zy sin(az') - zx cos(az'))): the sun term can cancel -> float64
        double rwz = rw;
        if (zf_not_1) rwz = rsqrt_pos(fma(P.hs_zf2, g2, 1.0));
        const float v = (float)fma_c(rwz, fma(P.hs_ky, zy, fma(P.hs_kx, zx, P.hs_sin_alt)), 1.5);
        sk.template put<P_HILLSHADE>(clamp_keep_nan(v, 0.0f, 255.0f));
    }
    if (!CURV) return;
    const double zxx = (double)zxxf, zyy = (double)zyyf, zxy = (double)zxyf;
    if (m & A_CURVATURE) sk.template put<P_CURVATURE>((float)(-2.0 * (zxx + zyy) * 100.0));
    if (m & (A_ANY_CURV & ~A_CURVATURE)) {
        const bool dir = SP::DIR < 0 ? (P.curv_directional != 0) : (SP::DIR != 0);
        const double zxzy = zx * zy;
        const double cross = 2.0 * zxy * zxzy;
        const double n_prof = fma(zyy, zy2, fma(zxx, zx2, cross));     // zxx zx^2 + 2 zxy zx zy + zyy zy^2
        const double n_tan = fma(zyy, zx2, fma(zxx, zy2, -cross));      // zxx zy^2 - 2 zxy zx zy + zyy zx^2
        const double rg2c = (rg * rg) * 100.0;                          // 100 / g2
        const double rg_t = (FULLRANGE && g2 < 10e-15) ? 0.0 : rg;      // planform / flowline: zero below 1e-14 (surfit.py:749-805)
        const double rw3 = (rw * rw) * rw;                              // 1 / (1 + g2)^1.5
        if (m & A_PROFILE) sk.template put<P_PROFILE>((float)(-(n_prof * (dir ? rg2c : rg2c * rw3))));
        if (m & A_TANGENTIAL) sk.template put<P_TANGENTIAL>((float)(-(n_tan * (dir ? rg2c : rg2c * rw))));
        if (m & A_PLANFORM) sk.template put<P_PLANFORM>((float)(-(n_tan * (rg2c * rg_t))));
        if (m & A_FLOWLINE) {
            const double n_flow = fma(zxzy, zxx - zyy, -zxy * (zx2 - zy2));
            sk.template put<P_FLOWLINE>((float)(n_flow * (dir ? (rg2c * rg) : (rg2c * rg_t) * rw)));
        }
        if (m & (A_MAXC | A_MINC)) {
            double vmax, vmin;
            if (dir) {
                const double half_tr = 50.0 * (zxx + zyy);
                const double hd = 50.0 * (zxx - zyy), sxy = 100.0 * zxy;
                const double rad = sqrt_nr_signed(fma(hd, hd, sxy * sxy));
                vmax = rad - half_tr;
                vmin = -half_tr - rad;
            } else {
                // mean = -h / w^3, gauss = K / w^4 (w^2 = 1 + g2, h = half the mean-curvature numerator): the
                // discriminant mean^2 - gauss = (h^2 - K w^2) / w^6 and the sums -h +- root stay in float64
                const double h = 0.5 * ((zxx + zyy) + n_tan);
                const double disc = fma(h, h, -(fma(zxx, zyy, -zxy * zxy) * opg));
                const double root = sqrt_nr_signed(disc);   // negative radicand -> NaN like the reference's ** 0.5
                const double rw3c = rw3 * 100.0;
                vmax = (root - h) * rw3c;
                vmin = (-h - root) * rw3c;
            }
            if (m & A_MAXC) sk.template put<P_MAXC>((float)(flat ? 0.0 : vmax));
            if (m & A_MINC) sk.template put<P_MINC>((float)(flat ? 0.0 : vmin));
        }
    }
}

// ---- lean tail (round 3): float32 DEM -> float32 attributes, float64 only where terms cancel --------------------------------
// The bar is 1e-6 TRUE relative error against the reference's float64 evaluation; the mixed tail above spends none of it on
// eight of the eleven planes.  Here float64 is kept for exactly the quantities whose terms cancel -- the squared gradient, the
// three curvature numerators, the discriminant of max / min curvature with its root and the two sums (+-root - h), the sun term
// of the hillshade -- and everything that only SCALES them runs in float32: both reciprocal square roots (v_rsq_f32 + one
// Newton step on the float32-rounded argument, ~9e-8), their powers and products, the final multiplications.  Worst-case bound
// of a curvature: 2 eps(rg) + 3 eps(rw) + 7 float32 roundings ~ 8.7e-7; measured maxima are about half (tools/ulp_report.py).
// v_rsq_f64 and its four float64 Newton operations cost as much as 13 float32 operations (tools/ubench.hip), and a float64
// multiply 1.8 float32 ones: this tail is ~25 float32-operation times (8 %) shorter per pixel than the mixed one.
// Hillshade = 1.5 + cos(slope') * S: the product nearly cancels 1.5 for almost fully shadowed pixels, where float32 factors
// would leave an ABSOLUTE error of 3e-7 on a value near 0 -- such pixels (value below 0.3 before clipping, and not safely
// negative) are reported back (return value) and recomputed by the float64 cold path of march_rows, like flat ground.
// Valid for squared gradients in [1e-13, 1e16] like the mixed hot path; everything else is the cold path's.
// Returns flag bits: TAIL_WANT_F64 (hillshade plane wants float64), TAIL_COLD (the pixel belongs to the cold path: a first
// derivative cancelled exactly or the squared gradient is outside the range above -- decided here from values the tail has
// anyway: min(|zx|, |zy|) and the rounded g2), TAIL_COLD_KNOWN (always set: the caller need not test again).
// (round 6: the two decisions leave as two booleans -- compare results the compiler keeps as scalar lane masks -- instead of flag
// bits in a vector register that had to be materialised with a select and tested again with a compare)
template <bool CURV, class SP, class SINK>
XD_HD void surface_pixel_lean(float zxf, float zyf, float zxxf, float zyyf, float zxyf, const TerrainParams& P, SINK& sk,
                              bool& cold_out, bool& want_f64_out, uint64_t& cold_any) {
    const uint32_t m = SP::CMASK ? SP::CMASK : P.mask;
    const bool deg = SP::DEG < 0 ? (P.degrees != 0) : (SP::DEG != 0);
    const bool zf_not_1 = SP::ZF1 < 0 ? (P.hs_zf2 != 1.0) : (SP::ZF1 == 0);
    const double zx = (double)zxf, zy = (double)zyf;
    const double zx2 = zx * zx, zy2 = zy * zy;   // exact (24-bit factors)
    const double g2 = zx2 + zy2;
    const float g2f = (float)g2;
    const float opgf = 1.0f + g2f;
    float rwf, rgf;                              // cos(slope), 1 / |grad|
    // (a launch that asks for the hillshade alone never uses 1 / |grad|: one seed and one plain Newton step instead of the pair --
    //  the same operations on the half it keeps, so the plane is bit-identical)
    if (m & (A_SLOPE | A_ASPECT | (A_ANY_CURV & ~A_CURVATURE))) rsq32_pair(opgf, g2f, rwf, rgf);
    else { rwf = rsq32(opgf); rgf = 0.0f; }
    bool want_f64 = false;
    const float ax = fabsf(zxf), ay = fabsf(zyf);
    const float amin = fminf(ax, ay);
    // (NaN: the range tests are false and NaN propagates by itself; fminf(0, NaN) = 0 sends a zero derivative next to a
    // NaN one to the cold path, which returns NaN for it as well)
    cold_out = (amin == 0.0f) | (g2f < 1e-13f) | (g2f > 1e16f);
    // "does any lane of the wave go to the cold path" = the three compares OR-ed on the scalar unit.  (Asked as
    // ballot(cold_out) != 0 hipcc selects 0 / 1 into a vector register and compares that again: two vector instructions per row
    // for a value that already sits in a scalar register pair.  `cold_out` itself is only read inside the cold branch.)
#if defined(__HIP_DEVICE_COMPILE__)
    {
        uint64_t k0, k1, k2;
        asm("v_cmp_eq_f32_e64 %0, 0, %1" : "=s"(k0) : "v"(amin));
        asm("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(k1) : "s"(1e-13f), "v"(g2f));
        asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(k2) : "s"(1e16f), "v"(g2f));
        cold_any = k0 | k1 | k2;
    }
#else
    cold_any = cold_out ? 1u : 0u;
#endif
    float as_slope = 0.0f, as_aspect = 0.0f;
    {
        const float xs = fminf((g2f * rgf) * rwf, rwf), xa = amin * rgf;   // min(sin, cos) of the slope; aspect octant
        if ((m & A_SLOPE) && (m & A_ASPECT)) asin32_pair(xs, xa, as_slope, as_aspect);
        else if (m & A_SLOPE) as_slope = asin32(xs);
        else if (m & A_ASPECT) as_aspect = asin32(xa);
    }
    if ((m & A_SLOPE) && SP::DEG == 1) {
        // degrees at compile time (round 6): the angle is scaled FIRST and the quarter turn is the exact constant 90 -- one
        // subtraction instead of the two-piece pi/2 and the scaling after it
        const float d = as_slope * DegScale<float>::v();
        sk.template put<P_SLOPE>(select_gt32(g2f, 1.0f, 90.0f - d, d));
    } else if (m & A_SLOPE) {
        const float a0 = as_slope;
        float a = select_gt32(g2f, 1.0f, (1.57079637f - a0) + -4.37113883e-08f, a0);
        if (deg) a = a * DegScale<float>::v();
        sk.template put<P_SLOPE>(a);
    }
    if ((m & A_ASPECT) && SP::DEG == 1) {
        // aspect = atan2(zx, zy) mod 360 from the octant angle d in [0, 45]: three reflections about exact constants (90, 180,
        // 360 degrees), each a subtraction and a select on a scalar mask -- 10 instructions where the sign-bit assembly of the
        // radian form below takes 16.  (-0 counts as not negative, like the reference's `< 0`; a zero derivative never reaches
        // this tail: such pixels are the cold path's.)
        const float d = as_aspect * DegScale<float>::v();
        float a = select_absgt32(zxf, zyf, 90.0f - d, d);
        a = select_gt32(0.0f, zyf, 180.0f - a, a);
        a = select_gt32(0.0f, zxf, 360.0f - a, a);
        sk.template put<P_ASPECT>(a);
    } else if (m & A_ASPECT) {
        // (the octant assembly of surface_pixel_mixed)
        const float a0 = as_aspect;
        const uint32_t sx = f32_bits(zxf + 0.0f) >> 31, sy = f32_bits(zyf + 0.0f) >> 31;
        const uint32_t xb = f32_bits(ay - ax) >> 31;
        const uint32_t q = sx ^ sy;
        const uint32_t odd = q ^ xb;
        const uint32_t oct = 4u * sx + 2u * q + odd;
        const float nf = (float)(int)((oct + 1u) >> 1);
        const float a0s = bits_f32(f32_bits(a0) ^ (odd << 31));
        const float t = fmaf(nf, -4.37113883e-08f, a0s);
        float a = fmaf(nf, 1.57079637f, t);
        if (deg) a = a * DegScale<float>::v();
        sk.template put<P_ASPECT>(a);
    }
    if (m & A_HILLSHADE) {
        float rwz = rwf;
        if (zf_not_1) rwz = rsq32(fmaf((float)P.hs_zf2, g2f, 1.0f));
        const float sun = (float)fma(P.hs_ky, zy, fma(P.hs_kx, zx, P.hs_sin_alt));   // can cancel: float64
        const float v = fmaf(rwz, sun, 1.5f);
        want_f64 = fabsf(v - 0.15f) < 0.15001f;   // -1e-5 < v < 0.30001: the float64 cold path decides (exact zero / tiny values)
        sk.template put<P_HILLSHADE>(clamp_keep_nan(v, 0.0f, 255.0f));
    }
    want_f64_out = want_f64;
    if (!CURV) return;
    const double zxx = (double)zxxf, zyy = (double)zyyf, zxy = (double)zxyf;
    if (m & A_CURVATURE) sk.template put<P_CURVATURE>((float)(-2.0 * (zxx + zyy) * 100.0));
    if (m & (A_ANY_CURV & ~A_CURVATURE)) {
        const bool dir = SP::DIR < 0 ? (P.curv_directional != 0) : (SP::DIR != 0);
        const double zxzy = zx * zy;
        const double cross = 2.0 * zxy * zxzy;
        const double n_prof = fma(zyy, zy2, fma(zxx, zx2, cross));
        const double n_tan = fma(zyy, zx2, fma(zxx, zy2, -cross));
        const float rg2c = (rgf * rgf) * 100.0f;     // 100 / g2
        const float rw2 = rwf * rwf;
        const float ntf = (float)n_tan;
        if (m & A_PROFILE) sk.template put<P_PROFILE>(-((float)n_prof * (dir ? rg2c : (rg2c * rwf) * rw2)));
        if (m & A_TANGENTIAL) sk.template put<P_TANGENTIAL>(-(ntf * (dir ? rg2c : rg2c * rwf)));
        const float rg3c = rg2c * rgf;               // 100 / g2^1.5
        if (m & A_PLANFORM) sk.template put<P_PLANFORM>(-(ntf * rg3c));
        if (m & A_FLOWLINE) {
            const double n_flow = fma(zxzy, zxx - zyy, -zxy * (zx2 - zy2));
            sk.template put<P_FLOWLINE>((float)n_flow * (dir ? rg3c : rg3c * rwf));
        }
        if (m & (A_MAXC | A_MINC)) {
            float vmax, vmin;
            if (dir) {
                // (inline constants only -- 0.5 in float64, the 100.0f the other curvatures use: the streaming kernels have no
                // scalar register to spare, and a spilled plane pointer is what their inline-asm stores must never meet)
                const double half_tr = 0.5 * (zxx + zyy);
                const double hd = 0.5 * (zxx - zyy);
                const double rad = sqrt_nr_signed(fma(hd, hd, zxy * zxy));
                vmax = (float)(rad - half_tr) * 100.0f;
                vmin = (float)(-half_tr - rad) * 100.0f;
            } else {
                // (h, the discriminant and +-root - h in float64 as in the mixed tail; only the common factor 100 / w^3 is float32)
                const double h = 0.5 * ((zxx + zyy) + n_tan);
                const double disc = fma(h, h, -(fma(zxx, zyy, -zxy * zxy) * (1.0 + g2)));
                const double root = sqrt_nr_signed(disc);
                const float rw3c = (rw2 * rwf) * 100.0f;
                vmax = (float)(root - h) * rw3c;
                vmin = (float)(-h - root) * rw3c;
            }
            if (m & A_MAXC) sk.template put<P_MAXC>(vmax);
            if (m & A_MINC) sk.template put<P_MINC>(vmin);
        }
    }
}

// TPI / TRI of a 3x3 window given as raw values (row-major n0..n8, n4 = centre).  Plain IEEE propagation.
template <class SP, class SINK>
XD_HD void window3_pixel(const double (&n)[9], double sum9, const TerrainParams& P, SINK& sk) {
    typedef typename SINK::out_t TOUT;
    const uint32_t m = SP::CMASK ? SP::CMASK : P.mask;
    const bool wilson = SP::WILSON < 0 ? (P.tri_wilson != 0) : (SP::WILSON != 0);
    const double c = n[4];
    if (m & A_TPI) sk.template put<P_TPI>((TOUT)((TOUT)fma_ks(sum9 - c, -0.125, c)));  // c - (sum9 - c) / 8, exact scaling
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
        sk.template put<P_ROUGH>(has_nan ? (TOUT)NAN : (TOUT)(mx - mn));
    }
    if (m & A_TRI) {
        double acc = c - c;  // the centre's own term: 0, or NaN when the centre is +-Inf (IEEE, like the reference)
        if (wilson) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (k != 4) acc += fabs(n[k] - c);
            sk.template put<P_TRI>((TOUT)((TOUT)(acc * 0.125)));
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (k != 4) {
                    const double d = n[k] - c;
                    acc = fma(d, d, acc);
                }
            sk.template put<P_TRI>((TOUT)((TOUT)sqrt_nr(acc)));
        }
    }
}

// Same for the mixed-precision tail: the window arrives as the raw float32 pixels.  Roughness (a float32 subtraction IS the
// reference's float64 difference rounded once) is bit-exact, TPI keeps its exact float64 sum (c - mean cancels), the TRI
// sums -- positive terms only, nothing cancels -- run in float32 (a few float32 roundings).
template <class SP, class SINK>
XD_HD void window3_pixel_mixed(const float (&n)[9], double sum9, const TerrainParams& P, SINK& sk) {
    const uint32_t m = SP::CMASK ? SP::CMASK : P.mask;
    const bool wilson = SP::WILSON < 0 ? (P.tri_wilson != 0) : (SP::WILSON != 0);
    const float c = n[4];
    if (m & A_TPI) {
        const double cd = (double)c;
        sk.template put<P_TPI>((float)fma_ks(sum9 - cd, -0.125, cd));
    }
    if (m & A_ROUGH) {
        float mx = n[0], mn = n[0];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            mx = (n[k] > mx) ? n[k] : mx;
            mn = (n[k] < mn) ? n[k] : mn;
        }
        const bool has_nan = (n[0] != n[0]) | (n[1] != n[1]) | (n[2] != n[2]) | (n[3] != n[3]) | (n[4] != n[4]) |
                             (n[5] != n[5]) | (n[6] != n[6]) | (n[7] != n[7]) | (n[8] != n[8]);
        sk.template put<P_ROUGH>(has_nan ? (float)NAN : (mx - mn));
    }
    if (m & A_TRI) {
        float acc = c - c;  // 0, or NaN when the centre is +-Inf (IEEE, like the reference)
        if (wilson) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (k != 4) acc += fabsf(n[k] - c);
            sk.template put<P_TRI>(acc * 0.125f);
        } else {
            sk.template put<P_TRI>(sqrt32_hw(tri_sumsq8(n, c, acc)));
        }
    }
}

// The reference's own accumulation for one derivative estimate: scipy.ndimage.convolve adds w * value in double over the
// non-zero weights in row-major order (xdem/spatialstats.py:2521-2525), the weights being table / divider in double, and
// rounds the sum to the input dtype.  The marcher below sums with INTEGER weights (exact for float32 pixels) and scales
// once; the two differ only where the terms cancel exactly -- flat or planar ground, where the reference returns its
// rounding residue (|zx| ~ 1e-15, hence an "aspect" of 198.43 deg on a flat Florinsky window) instead of 0.  Such pixels
// (an exact 0 from the fast sum) are recomputed here so that they carry the reference's value.  `win` = top-left pixel of
// the M x M window in the LDS tile (tile row `top`, `left` columns from the lane's own; rows through the row addresser of
// march_column).  No fused multiply-add: SciPy's loop does not contract.
template <int M, typename TIN, class ROWS>
XD_HD TIN ref_order_sum(const ROWS& rows, int top, int left, const double* w) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    double acc = 0.0;
    // (deliberately not unrolled: a cold path, kept small so that the hot loop stays resident in the instruction cache)
#if defined(__clang__)
#pragma clang loop unroll(disable)
#endif
    for (int a = 0; a < M; ++a) {
        const auto win = rows.ptr(top + a) + left;
#if defined(__clang__)
#pragma clang loop unroll(disable)
#endif
        for (int b = 0; b < M; ++b) {
            const double wk = w[a * M + b];
            if (wk != 0.0) {
                const double t = wk * (double)win[b];
                acc = acc + t;
            }
        }
    }
    return (TIN)acc;
}

// squared gradient outside the validity range of the mixed-precision tail (NaN: false -- NaN propagates by itself)
XD_HD bool mixed_tail_out_of_range(float zxf, float zyf) {
    const float g2f = fmaf(zyf, zyf, zxf * zxf);
    return (g2f < 1e-13f) | (g2f > 1e16f);
}
// a first derivative that cancelled exactly (the product also underflows for two tiny derivatives: a few extra visits of
// the cold path, which is correct for every pixel)
template <typename T> XD_HD bool first_derivative_zero(T zx, T zy) { return (zx * zy) == (T)0; }
template <typename T> XD_HD bool any_zero5(T a, T b, T c, T d, T e) {
    return fmin(fmin(fabs((double)a), fabs((double)b)), fmin(fabs((double)c), fmin(fabs((double)d), fabs((double)e)))) == 0.0;
}
template <> XD_HD bool any_zero5<float>(float a, float b, float c, float d, float e) {
    return fminf(fminf(fabsf(a), fabsf(b)), fminf(fabsf(c), fminf(fabsf(d), fabsf(e)))) == 0.0f;
}

// ---- the column marcher -------------------------------------------------------------------------------
// `col` points at the tile element of this thread's column in the first tile row; tile row t holds raster
// row (first output row - HALO + t); element col[t * pitch + d] is the pixel d columns to the right.
// Emits n_out output rows; the plane pointers in `out` are already offset to the tile's first output pixel
// (wave-uniform, so stores use the scalar-base + 32-bit VGPR offset addressing form) and the BYTE offset of
// output row i is o0 + i * ostride (32-bit: a tile spans far less than 4 GiB per plane).
template <int FIT> struct Halo { static constexpr int v = (FIT == 2) ? 2 : 1; };
template <typename A, typename B> struct SameT { static constexpr bool v = false; };
template <typename A> struct SameT<A, A> { static constexpr bool v = true; };

// hot-path tail; returns true if it has decided the cold path itself (lean tail only: `cold` = the pixel belongs to the cold
// path, `want_f64` = its nearly black hillshade wants the float64 factors)
template <bool MIXED, bool CURV, class SP, typename TIN, class SINK> struct SurfaceTail {
    static XD_HD bool go(TIN zx, TIN zy, TIN zxx, TIN zyy, TIN zxy, const TerrainParams& P, SINK& sk, bool&, bool&, uint64_t&) {
        surface_pixel<CURV, SP, SINK>((double)zx, (double)zy, (double)zxx, (double)zyy, (double)zxy, P, sk);
        return false;
    }
};
template <bool CURV, class SP, class SINK> struct SurfaceTail<true, CURV, SP, float, SINK> {
    static XD_HD bool go(float zx, float zy, float zxx, float zyy, float zxy, const TerrainParams& P, SINK& sk, bool& cold, bool& want_f64,
                         uint64_t& cold_any) {
        if (SP::F64TAIL == 2) {
            surface_pixel_lean<CURV, SP, SINK>(zx, zy, zxx, zyy, zxy, P, sk, cold, want_f64, cold_any);
            return true;
        }
        surface_pixel_mixed<CURV, SP, false, SINK>(zx, zy, zxx, zyy, zxy, P, sk);
        return false;
    }
};
template <bool MIXED, bool CURV, class SP, typename TIN, class SINK> struct ColdTail {
    static XD_HD void go(TIN zx, TIN zy, TIN zxx, TIN zyy, TIN zxy, const TerrainParams& P, SINK& sk) {
        surface_pixel<CURV, SP, SINK>((double)zx, (double)zy, (double)zxx, (double)zyy, (double)zxy, P, sk);
    }
};
template <bool CURV, class SP, class SINK> struct ColdTail<true, CURV, SP, float, SINK> {
    static XD_HD void go(float zx, float zy, float zxx, float zyy, float zxy, const TerrainParams& P, SINK& sk) {
        surface_pixel_mixed<CURV, SP, true, SINK>(zx, zy, zxx, zyy, zxy, P, sk);
    }
};
template <bool MIXED, class SP, typename TIN, class SINK> struct WindowTail {
    static XD_HD void go(const TIN (&n)[9], double sum9, const TerrainParams& P, SINK& sk) {
        const double nd[9] = {(double)n[0], (double)n[1], (double)n[2], (double)n[3], (double)n[4],
                              (double)n[5], (double)n[6], (double)n[7], (double)n[8]};
        window3_pixel<SP, SINK>(nd, sum9, P, sk);
    }
};
template <class SP, class SINK> struct WindowTail<true, SP, float, SINK> {
    static XD_HD void go(const float (&n)[9], double sum9, const TerrainParams& P, SINK& sk) {
        window3_pixel_mixed<SP, SINK>(n, sum9, P, sk);
    }
};

// hillshade of one pixel in float64 (the formula of surface_pixel_mixed): the lean tail's fix-up for nearly black pixels
template <class SP, typename TIN, class SINK> struct HillshadeF64 {
    static XD_HD void go(TIN, TIN, const TerrainParams&, SINK&) {}
};
template <class SP, class SINK> struct HillshadeF64<SP, float, SINK> {
    static XD_HD void go(float zxf, float zyf, const TerrainParams& P, SINK& sk) {
        const bool zf_not_1 = SP::ZF1 < 0 ? (P.hs_zf2 != 1.0) : (SP::ZF1 == 0);
        const double zx = (double)zxf, zy = (double)zyf;
        const double g2 = zx * zx + zy * zy;
        const double rwz = rsqrt_pos(zf_not_1 ? fma(P.hs_zf2, g2, 1.0) : 1.0 + g2);
        const float v = (float)fma_c(rwz, fma(P.hs_ky, zy, fma(P.hs_kx, zx, P.hs_sin_alt)), 1.5);
        sk.template put<P_HILLSHADE>(clamp_keep_nan(v, 0.0f, 255.0f));
    }
};

// Row addressers: where tile row t of this lane's column lives.  RowsLinear: a row-major tile (`col` = the lane's column in
// tile row 0).  The streaming kernel of terrain_tile.h supplies a per-wave ring refilled by LDS-DMA; `step(r)` is its hook
// at the start of march step r (before the step's read of tile row r + 1).
template <typename TIN> struct RowsLinear {
    const TIN* col;
    int pitch;
    XD_HD const TIN* ptr(int t) const { return col + (int64_t)t * pitch; }
    XD_HD void step(int) {}
};

template <int FIT, bool CURV, bool WIN, class SP, typename TIN, class SINK, class ROWS>
XD_HD void march_rows(ROWS& rows, int n_out, const TerrainParams& P, SINK& sk);

template <int FIT, bool CURV, bool WIN, class SP, typename TIN, class SINK>
XD_HD void march_column(const TIN* col, int pitch, int n_out, const TerrainParams& P, SINK& sk) {
    RowsLinear<TIN> rows{col, pitch};
    march_rows<FIT, CURV, WIN, SP, TIN, SINK, RowsLinear<TIN>>(rows, n_out, P, sk);
}

template <int FIT, bool CURV, bool WIN, class SP, typename TIN, class SINK, class ROWS>
XD_HD void march_rows(ROWS& rows, int n_out, const TerrainParams& P, SINK& sk) {
    typedef typename SINK::out_t TOUT;
    constexpr int HALO = Halo<FIT>::v;
    constexpr int NS = 2 * HALO + 1;  // rotating window slots
    constexpr bool MIXED = SameT<TIN, float>::v && SameT<TOUT, float>::v && (SP::F64TAIL != 1);
    const int nrows = n_out + 2 * HALO;
    const uint32_t m = SP::CMASK ? SP::CMASK : P.mask;

    // Florinsky (round 6): FORWARD accumulation.  Output row i needs tile rows i .. i + 4; instead of keeping seven per-row
    // partials for five rows (35 float64 values) and combining them when the window is complete, every new row is pushed at
    // once into the five sums of each of the five outputs it belongs to -- slot (i mod 5) holds the running sum of output i, a
    // row of age a = r - i in that output's window adds its weight-a share.  Same operation count, 20 live float64 values
    // instead of 33 between two rows: the streaming kernel drops from 164 to <= 128 VGPRs (a fourth wave per SIMD).
    double Zx[NS], Zy[NS], Zxx[NS], Zyy[NS], Zxy[NS], Pz[NS];
    // ... and the raw pixels of the 3 x 3 window of TPI / TRI are READ AGAIN from the tile (LDS: an idle pipe in this kernel) when
    // their output row is emitted, through the row addresses kept from the prefetches (four live 32-bit values instead of the
    // twelve pixels of four rows)
    typedef decltype(rows.ptr(0)) rowptr_t;
    constexpr bool REREAD = (FIT == 2) && WIN;
    rowptr_t Pn[NS];
    // per-row partials -- 3x3 fits
    double Dr[NS], S[NS], Zc[NS];
    // 3-wide row sums and the raw pixels of the three centre columns for TPI / TRI (and the 3x3 detector)
    double R3[NS];
    TIN Nl[NS], Nc[NS], Nr[NS];

    if (FIT == 2) {   // (the sums of the "outputs" above the band's first one are formed and never emitted: give them a value)
#pragma unroll
        for (int k = 0; k < NS; ++k) Zx[k] = Zy[k] = Zxx[k] = Zyy[k] = Zxy[k] = Pz[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) Pn[k] = rows.ptr(0) - 2;
    // the tile row of the NEXT step is fetched from LDS one step ahead, so its latency hides behind a whole row of math
    TIN nx0 = (TIN)0, nx1, nx2, nx3, nx4 = (TIN)0;
    {
        const auto row0 = rows.ptr(0);
        nx1 = row0[-1]; nx2 = row0[0]; nx3 = row0[1];
        if (FIT == 2) { nx0 = row0[-2]; nx4 = row0[2]; }
    }
    for (int r0 = 0; r0 < nrows; r0 += NS) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int r = r0 + k;
            if (r < nrows) {
                const TIN t0 = nx0, tl = nx1, tc = nx2, tr = nx3, t4 = nx4;
                // the prefetched row is CONSUMED (widened) before the next row's LDS reads are issued: the compiler otherwise
                // hoists those reads above the conversions and has to wait for them on the spot (s_waitcnt lgkmcnt(0) right
                // after the reads: a full LDS round trip per row); in this order the wait at the top of a row is for reads
                // issued a whole row of arithmetic earlier
                const double zl = (double)tl, zc = (double)tc, zr = (double)tr;
                const double z0 = FIT == 2 ? (double)t0 : 0.0, z4 = FIT == 2 ? (double)t4 : 0.0;
                XD_SCHED_FENCE();
                rows.step(r);
                TIN w9[9] = {};
                {
                    const auto row = rows.ptr((r + 1 < nrows) ? r + 1 : r);
                    nx1 = row[-1]; nx2 = row[0]; nx3 = row[1];
                    if (FIT == 2) { nx0 = row[-2]; nx4 = row[2]; }
                    if (REREAD) {
                        // (addresses are kept two columns to the left: an LDS instruction adds an UNSIGNED offset to its address register)
                        // rows r - 3 .. r - 1 = the 3 x 3 window of the output this step emits (centre row r - 2); their addresses are
                        // the ones of the prefetches of steps r - 4 .. r - 2.  (Both tile forms keep those rows readable: the streaming
                        // ring refills a half only once its newest row is more than four steps old.)
                        const rowptr_t a1 = Pn[(k + 2) % NS], a2 = Pn[(k + 3) % NS], a3 = Pn[(k + 4) % NS];
                        w9[0] = a1[1]; w9[1] = a1[2]; w9[2] = a1[3];
                        w9[3] = a2[1]; w9[4] = a2[2]; w9[5] = a2[3];
                        w9[6] = a3[1]; w9[7] = a3[2]; w9[8] = a3[3];
                        Pn[(k + 1) % NS] = row - 2;
                    }
                }
                XD_SCHED_FENCE();
                if (FIT == 2) {
                    // Row partials.  With p = z0 + z4, q = zl + zr, A = zr - zl, B = z4 - z0 (columns -2 .. +2 of this row):
                    //   D  = A + 2 B                      the row's share of zxy (weights -2 -1 0 1 2)
                    //   R  = p + q + zc                   ... of zyy (plain row sum)
                    //   Wr = 2 p - q - 2 zc               ... of zxx (weights 2 -1 -2 -1 2)
                    //   Ua = 44 p + 62 q + 68 zc = 56 R - 6 Wr,   Ub = -31 p + 5 q + 17 zc = -7 R - 12 Wr      (zy: rows +-1 / +-2)
                    //   X1 = 44 A - 31 B,   X2 = 62 A + 5 B = X1 + 18 D,   X3 = 68 A + 17 B = X2 + 6 D        (zx: rows +-2 / +-1 / 0)
                    // -- the second forms are identities of Florinsky's least-squares weights (surfit.py:204-252): 4 + 4 operations
                    // where the weighted sums taken literally cost 6 + 6.  Integer weights on float32 pixels: every sum is exact.
                    const double p = z0 + z4, q = zl + zr;
                    const double Ar = zr - zl, Br = z4 - z0;
                    const double Dr2 = fma_2(Br, Ar);
                    double Rr = 0.0, Wrr = 0.0, Uar, Ubr;
                    if (CURV) {
                        Rr = (p + q) + zc;
                        Wrr = fma(2.0, p - zc, -q);
                        Uar = fma(56.0, Rr, -6.0 * Wrr);
                        Ubr = fma(-7.0, Rr, -12.0 * Wrr);
                    } else {   // (without second derivatives nothing else needs R and Wr: the literal forms are the cheaper ones)
                        Uar = fma(68.0, zc, fma(62.0, q, 44.0 * p));
                        Ubr = fma(17.0, zc, fma(5.0, q, -31.0 * p));
                    }
                    const double X1 = fma(44.0, Ar, -31.0 * Br);
                    const double X2 = fma(18.0, Dr2, X1);
                    const double X3 = fma(6.0, Dr2, X2);
                    // this row has age a in the window of output r - a, whose sums live in slot (k - a) mod 5
                    const int s0 = k % NS, s1 = (k + 4) % NS, s2 = (k + 3) % NS, s3 = (k + 2) % NS, s4 = (k + 1) % NS;
                    Zx[s4] += X1; Zx[s3] += X2; Zx[s2] += X3; Zx[s1] += X2; Zx[s0] = X1;
                    // zy = (Ua[+1] - Ua[-1]) + (Ub[+2] - Ub[-2]): the slot holds +Ub[-2] after age 0, the signs are settled at age 1
                    Zy[s4] += Ubr; Zy[s3] += Uar; Zy[s1] = -Zy[s1] - Uar; Zy[s0] = Ubr;
                    if (CURV) {
                        Zxx[s4] += Wrr; Zxx[s3] += Wrr; Zxx[s2] += Wrr; Zxx[s1] += Wrr; Zxx[s0] = Wrr;
                        // zyy: 2 R[-2] - R[-1] - 2 R[0] - R[+1] + 2 R[+2]
                        Zyy[s4] = fma(2.0, Rr, Zyy[s4]); Zyy[s3] -= Rr; Zyy[s2] = fma(-2.0, Rr, Zyy[s2]); Zyy[s1] = fma(2.0, Zyy[s1], -Rr); Zyy[s0] = Rr;
                        // zxy: 2 D[-2] + D[-1] - D[+1] - 2 D[+2]
                        Zxy[s4] = fma(-2.0, Dr2, Zxy[s4]); Zxy[s3] -= Dr2; Zxy[s1] = fma_2(Zxy[s1], Dr2); Zxy[s0] = Dr2;
                    } else {
                        Pz[s2] = Uar;   // the centre row's partial holds the centre pixel (see `poison` below)
                    }
                    if (WIN) R3[k] = q + zc;
                } else {
                    Dr[k] = zr - zl;
                    S[k] = zl + zr;
                    Zc[k] = zc;
                    R3[k] = (zl + zr) + zc;
                }
                if (WIN && !REREAD) { Nl[k] = tl; Nc[k] = tc; Nr[k] = tr; }

                const int i = r - 2 * HALO;  // output row whose window is now complete
                if (i >= 0) {
                    sk.begin_row(i);
                    // slot of window row (centre + d): the newest row (slot k) is centre + HALO
#define XD_SLOT(d) ((k + NS - HALO + (d)) % NS)
                    TIN zx, zy, zxx = (TIN)0, zyy = (TIN)0, zxy = (TIN)0;   // rounded to the input dtype like the reference
                    // `det` = a sum over every pixel of the window: non-finite <=> some pixel non-finite or outside the raster;
                    // det - det (0 or NaN) rides on the scaling fma of zx (and zxx): invalid windows come out NaN at no cost
                    double det, poison;
                    if (FIT == 2) {
                        const int se = (k + 1) % NS;   // the output whose fifth row this was
                        const double sx = Zx[se], sy = Zy[se];
                        if (CURV) {
                            det = Zxx[se];
                            poison = det - det;
                        } else {
                            // Without second derivatives there is no sum over the whole window to borrow, and one of its own cost seven
                            // float64 operations per pixel (row sums, their sum, det - det) in launches that are bound by instruction issue.
                            // The two first-derivative sums already see every pixel but the centre one with a NONZERO weight -- sx all
                            // rows of the four outer columns, sy all columns of the four outer rows -- and Ua of the centre row holds the
                            // centre pixel (weight 68): their sum is non-finite exactly when some pixel of the window is (a finite raster
                            // cannot overflow float64 here), so v - v is the same 0 / NaN as det - det, in three operations.
                            det = 0.0;
                            const double v = (sx + sy) + Pz[se];
                            poison = v - v;
                        }
                        zx = (TIN)fma(-sx, P.s1, poison);
                        zy = (TIN)(sy * P.s1);
                        if (CURV) {
                            // (every curvature but the deprecated `curvature` = -2 (zxx + zyy) depends on zx, which carries the poison)
                            zxx = (m & A_CURVATURE) ? (TIN)fma(det, P.sxx, poison) : (TIN)(det * P.sxx);
                            zyy = (TIN)(Zyy[se] * P.sxx);
                            zxy = (TIN)(Zxy[se] * P.sxy);
                        }
                    } else {
                        const int m1 = XD_SLOT(-1), c0 = XD_SLOT(0), p1 = XD_SLOT(1);
                        det = (R3[m1] + R3[c0]) + R3[p1];
                        poison = det - det;
                        if (FIT == 0) {  // Horn: [1 2 1] smoothing across the derivative direction
                            zx = (TIN)fma(-(fma(2.0, Dr[c0], Dr[m1] + Dr[p1])), P.s1, poison);
                            zy = (TIN)((fma(2.0, Zc[p1] - Zc[m1], S[p1] - S[m1])) * P.s1);
                        } else {         // Zevenbergen-Thorne: central differences
                            zx = (TIN)fma(-Dr[c0], P.s1, poison);
                            zy = (TIN)((Zc[p1] - Zc[m1]) * P.s1);
                            if (CURV) {
                                zxx = (TIN)fma(fma(-2.0, Zc[c0], S[c0]), P.sxx, poison);
                                zyy = (TIN)(fma(-2.0, Zc[c0], Zc[m1] + Zc[p1]) * P.sxx);
                                zxy = (TIN)((Dr[m1] - Dr[p1]) * P.sxy);
                            }
                        }
                    }
                    if (REREAD) {
                        // TPI / TRI first: their nine pixels are dead before the surface tail's temporaries come alive
                        const int w1 = XD_SLOT(-1), w0 = XD_SLOT(0), w2 = XD_SLOT(1);
                        WindowTail<MIXED, SP, TIN, SINK>::go(w9, (R3[w1] + R3[w0]) + R3[w2], P, sk);
                        XD_SCHED_FENCE();
                    }
                    bool tail_decided = false, tail_is_cold = false, tail_cold = false;
                    uint64_t tail_cold_any = 0;
                    if (m & ~A_ANY_WIN) {
                        // hot path: every lane, straight-line (one basic block per output row)
                        tail_decided = SurfaceTail<MIXED, CURV, SP, TIN, SINK>::go(zx, zy, zxx, zyy, zxy, P, sk, tail_is_cold, tail_cold, tail_cold_any);
                    }
                    if (WIN && !REREAD) {
                        const int w1 = XD_SLOT(-1), w0 = XD_SLOT(0), w2 = XD_SLOT(1);
                        const TIN n[9] = {Nl[w1], Nc[w1], Nr[w1], Nl[w0], Nc[w0], Nr[w0], Nl[w2], Nc[w2], Nr[w2]};
                        WindowTail<MIXED, SP, TIN, SINK>::go(n, (R3[w1] + R3[w0]) + R3[w2], P, sk);
                    }
                    if (m & ~A_ANY_WIN) {
                        // Cold path, entered by the whole wave only if some lane needs it (a scalar branch: the hot path above stays
                        // free of exec-masked regions), executed by the lanes that need it, which overwrite what the hot path stored:
                        //  * exact cancellation of a derivative sum (flat / planar ground): the pixel gets the reference's own
                        //    residue (ref_order_sum) and the float64 tail;
                        //  * mixed-precision tail outside its validity range: float64 tail.
                        // (An exactly cancelling SECOND derivative alone does not send a pixel here: its residue only adds ~1e-15 of
                        // the other curvature terms -- float64 rounding noise the reference's own result carries as well.)
                        bool cold;
                        if (tail_decided) {
                            cold = tail_is_cold;
                        } else {
                            cold = first_derivative_zero<TIN>(zx, zy);
                            if (MIXED) cold |= mixed_tail_out_of_range((float)zx, (float)zy);
                        }
                        // lean tail: a nearly black hillshade is the only plane its float32 factors cannot carry -- that one plane is
                        // recomputed in float64 (its own wave-uniform branch: ~3 % of the wave rows of steep terrain come here,
                        // the full cold tail below would cost them ten times as much)
                        if (MIXED && SP::F64TAIL == 2) {
#if defined(__HIP_DEVICE_COMPILE__)
                            if (__builtin_expect(__builtin_amdgcn_ballot_w64(tail_cold) != 0, 0))
#endif
                            {
                                if (tail_cold && !cold) HillshadeF64<SP, TIN, SINK>::go(zx, zy, P, sk);
                            }
                        }
#if defined(XD_NO_COLD)  // (instruction-count analysis builds only: tools/isa_stats.py)
                        cold = false;
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(XD_NO_COLD)
                        if (__builtin_expect(tail_decided ? (tail_cold_any != 0) : (__builtin_amdgcn_ballot_w64(cold) != 0), 0))
#elif defined(__HIP_DEVICE_COMPILE__)
                        if (__builtin_expect(__builtin_amdgcn_ballot_w64(cold) != 0, 0))
#endif
                        {
                            if (cold) {
                                // (window of output row i = tile rows i .. i + 2 HALO, columns -HALO .. +HALO around the lane's own)
                                if (zx == (TIN)0) zx = ref_order_sum<NS, TIN, ROWS>(rows, i, -HALO, P.wref[0]);
                                if (zy == (TIN)0) zy = ref_order_sum<NS, TIN, ROWS>(rows, i, -HALO, P.wref[1]);
                                if (CURV) {
                                    if (zxx == (TIN)0) zxx = ref_order_sum<NS, TIN, ROWS>(rows, i, -HALO, P.wref[2]);
                                    if (zyy == (TIN)0) zyy = ref_order_sum<NS, TIN, ROWS>(rows, i, -HALO, P.wref[3]);
                                    if (zxy == (TIN)0) zxy = ref_order_sum<NS, TIN, ROWS>(rows, i, -HALO, P.wref[4]);
                                }
                                const TIN pz = (TIN)poison;
                                ColdTail<MIXED, CURV, SP, TIN, SINK>::go(zx + pz, zy, CURV ? zxx + pz : zxx, zyy, zxy, P, sk);
                            }
                        }
                    }
                    sk.end_row(i);
#undef XD_SLOT
                }
            }
        }
    }
}

}  // namespace xd
