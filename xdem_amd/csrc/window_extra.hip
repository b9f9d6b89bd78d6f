// window_extra.hip -- the two windowed indexes outside the fused tile kernel: rugosity and fractal roughness.
//
// Both are per-window callbacks in the reference (xdem/terrain/window.py:316-401 `_fractal_roughness_func`,
// 466-563 `_rugosity_func`; run per pixel by scipy.ndimage.generic_filter(mode="constant", cval=nan) or by the Numba
// loop, window.py:873-923) whose intermediates live in `out_dtype` arrays and whose reductions are Python `sum`
// (left to right).  The kernels below keep that arithmetic type and that order, so float32 outputs are reproduced
// operation by operation (rugosity: bit-exact; fractal roughness: up to the float32 logarithm, see below).
//
//   rugosity           3x3 window: 16 half segment lengths sqrt(dz^2 + dl^2)/2, 8 Heron triangle areas, sum / res^2.
//                      One thread per pixel, neighbours through L1/L2 (8 B/pixel algorithmic: HBM-bound).
//   fractal roughness  w x w window (default 13): voxel heights V = clip(z - z_c, 0, w); for every divisor q of w//2
//                      the ((w-1)/q)^2 block maxima of V are summed and divided by q; output = -slope of log Ns over
//                      log q.  ~1.4 k float ops per pixel: VALU-bound.  The DEM patch of a 64x4-pixel workgroup is
//                      staged in LDS once; block maxima are taken on the raw elevations (clip o shift is monotone,
//                      so max V = clip(max z - z_c)); NaN bookkeeping rides on the q = 1 pass, which visits every
//                      pixel of the (w-1)^2 region that all box sizes cover.  The default window has a fully
//                      unrolled single-pass form (every pixel read once from LDS, box maxima of all q kept in
//                      registers); any other odd window runs the same arithmetic with runtime loops.
//                      log() is evaluated in float64 and rounded to out_dtype (NumPy's float32 log is a SIMD routine
//                      that is not correctly rounded; parity for this attribute is 1e-6 relative, not bit-exact).
#include <math.h>

#include <string.h>
#include <vector>

#include "common.h"

#pragma clang fp contract(off)

namespace xd {

constexpr int FR_MAXQ = 24;   // divisors of w//2 (w <= 1023 has at most 24)
constexpr int FR_COLS = 64, FR_ROWS = 4;

template <typename T> struct RugParams {
    T dl2_straight, dl2_diag, inv_area_den;  // T(L)^2, T(sqrt(2) L)^2 (each squared in T), T(L^2)
};

template <typename T> struct FracParams {
    int w, n;
    int q[FR_MAXQ];
    T qf[FR_MAXQ];   // T(q)
    T x[FR_MAXQ];    // float16(log q), exactly representable in T
    T n_t, m_x, ss_xx;
};

template <typename TIN, typename T> __device__ __forceinline__ T height_diff(TIN z, TIN c) {
    // reference: float64 difference of the float64 window buffer, stored into an out_dtype array.  For float32 in and
    // out the float32 subtraction is the same number (the exact difference rounded once).
    if (sizeof(TIN) == 4 && sizeof(T) == 4) return (T)(z - c);
    return (T)((double)z - (double)c);
}

template <typename T> __device__ __forceinline__ T sqrt_t(T v);
template <> __device__ __forceinline__ float sqrt_t<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double sqrt_t<double>(double v) { return sqrt(v); }

template <typename TIN, typename T>
__global__ __launch_bounds__(256) void rugosity_kernel(const TIN* dem, int64_t H, int64_t W, int64_t stride,
                                                       int64_t halo_top, int64_t halo_bottom, RugParams<T> P, T* out) {
    const int64_t x = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t y = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    TIN z[9];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int64_t yy = y + dy;
        const bool rowok = (yy >= -halo_top) && (yy < H + halo_bottom);
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int64_t xx = x + dx;
            z[(dy + 1) * 3 + dx + 1] = (rowok && xx >= 0 && xx < W) ? dem[(yy + halo_top) * stride + xx] : (TIN)NAN;
        }
    }
    // half surface lengths of the 16 segments (window.py:478-515)
    T hsl[16];
    const int nb[8] = {0, 1, 2, 3, 5, 6, 7, 8};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const T dz = height_diff<TIN, T>(z[4], z[nb[i]]);
        const bool diag = (nb[i] == 0 || nb[i] == 2 || nb[i] == 6 || nb[i] == 8);
        hsl[i] = sqrt_t<T>(dz * dz + (diag ? P.dl2_diag : P.dl2_straight)) * (T)0.5;
    }
    const int ea[8] = {0, 1, 6, 7, 0, 3, 2, 5}, eb[8] = {1, 2, 7, 8, 3, 6, 5, 8};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const T dz = height_diff<TIN, T>(z[ea[i]], z[eb[i]]);
        hsl[8 + i] = sqrt_t<T>(dz * dz + P.dl2_straight) * (T)0.5;
    }
    // 8 triangles, Heron's formula (window.py:517-553)
    const int ta[8] = {3, 0, 1, 2, 4, 7, 6, 5}, tb[8] = {0, 1, 2, 4, 7, 6, 5, 3}, tc[8] = {12, 8, 9, 14, 15, 11, 10, 13};
    T total = (T)0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const T a = hsl[ta[i]], b = hsl[tb[i]], c = hsl[tc[i]];
        const T hs = ((a + b) + c) * (T)0.5;
        const T area = sqrt_t<T>(((hs * (hs - a)) * (hs - b)) * (hs - c));
        total = (i == 0) ? area : total + area;
    }
    out[y * W + x] = total / P.inv_area_den;
}

template <typename T> __device__ __forceinline__ T clip_w(T v, T wmax) {
    return fmin(fmax(v, (T)0), wmax);  // NaN is tracked separately by the caller
}

// regression of log Ns on log q (window.py:380-401), out_dtype arithmetic.  The two sums follow np.add.reduce: left to
// right below 8 terms; from 8 terms on, 8 interleaved accumulators over the leading multiple of 8, combined as a tree,
// then the remaining terms one by one.
template <typename T> struct Regress {
    T sy[8], sxy[8];
    T ty = (T)0, txy = (T)0;   // running totals once the tree is combined (or from the start when n < 8)
    int i = 0, n_main = 0;     // n_main = 8 * (n / 8) when n >= 8, else 0
    __device__ __forceinline__ void begin(int n) { n_main = (n >= 8) ? (n & ~7) : 0; }
    __device__ __forceinline__ void add(T sum_q, T qf, T x) {
        const T y = (T)log((double)(sum_q / qf));
        const T yx = y * x;
        if (i < n_main) {
            const int slot = i & 7;
            const bool fresh = i < 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (slot == j) {
                    sy[j] = fresh ? y : sy[j] + y;
                    sxy[j] = fresh ? yx : sxy[j] + yx;
                }
            }
            if (i == n_main - 1) {
                ty = ((sy[0] + sy[1]) + (sy[2] + sy[3])) + ((sy[4] + sy[5]) + (sy[6] + sy[7]));
                txy = ((sxy[0] + sxy[1]) + (sxy[2] + sxy[3])) + ((sxy[4] + sxy[5]) + (sxy[6] + sxy[7]));
            }
        } else {
            ty = (i == 0) ? y : ty + y;
            txy = (i == 0) ? yx : txy + yx;
        }
        ++i;
    }
    __device__ __forceinline__ T result(const T n_t, const T m_x, const T ss_xx) const {
        const T m_y = ty / n_t;
        const T ss_xy = txy - (n_t * m_y) * m_x;
        return -(ss_xy / ss_xx);
    }
};

template <int WC, typename TIN, typename T>
__global__ __launch_bounds__(256) void fractal_kernel(const TIN* dem, int64_t H, int64_t W, int64_t stride,
                                                      int64_t halo_top, int64_t halo_bottom, FracParams<T> P, T* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TIN* tile = reinterpret_cast<TIN*>(smem);
    const int w = WC ? WC : P.w;
    const int hw = w / 2;
    const int pitch = FR_COLS + w - 1, trows = FR_ROWS + w - 1;
    const int64_t x0 = (int64_t)blockIdx.x * FR_COLS, y0 = (int64_t)blockIdx.y * FR_ROWS;
    for (int idx = threadIdx.x; idx < pitch * trows; idx += 256) {
        const int r = idx / pitch, cidx = idx - r * pitch;
        const int64_t gy = y0 - hw + r, gx = x0 - hw + cidx;
        const bool ok = (gy >= -halo_top) && (gy < H + halo_bottom) && gx >= 0 && gx < W;
        tile[idx] = ok ? dem[(gy + halo_top) * stride + gx] : (TIN)NAN;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int64_t x = x0 + lx, y = y0 + ly;
    if (x >= W || y >= H) return;
    const TIN* t = tile + ly * pitch + lx;  // top-left pixel of this thread's window
    const TIN c = t[hw * pitch + hw];
    const T wmax = (T)w;
    bool bad = false;
    Regress<T> reg;
    reg.begin(WC == 13 ? 4 : P.n);
    if (WC == 13) {
        // single pass: q = 1, 2, 3, 6 box maxima side by side
        T s1 = (T)0, s2 = (T)0, s3 = (T)0, s6 = (T)0;
        TIN m2[6], m3[4], m6[2];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const TIN z = t[j * pitch + k];
                const T v = height_diff<TIN, T>(z, c);
                bad |= (v != v);
                const T vc = clip_w<T>(v, wmax);
                s1 = (j == 0 && k == 0) ? vc : s1 + vc;
                m2[k / 2] = (j % 2 == 0 && k % 2 == 0) ? z : fmax(m2[k / 2], z);
                m3[k / 3] = (j % 3 == 0 && k % 3 == 0) ? z : fmax(m3[k / 3], z);
            }
            if (j % 2 == 1) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const T vc = clip_w<T>(height_diff<TIN, T>(m2[k], c), wmax);
                    s2 = (j == 1 && k == 0) ? vc : s2 + vc;
                }
            }
            if (j % 3 == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const T vc = clip_w<T>(height_diff<TIN, T>(m3[k], c), wmax);
                    s3 = (j == 2 && k == 0) ? vc : s3 + vc;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const TIN pair = fmax(m3[2 * k], m3[2 * k + 1]);
                    m6[k] = (j % 6 == 2) ? pair : fmax(m6[k], pair);
                }
            }
            if (j % 6 == 5) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const T vc = clip_w<T>(height_diff<TIN, T>(m6[k], c), wmax);
                    s6 = (j == 5 && k == 0) ? vc : s6 + vc;
                }
            }
        }
        reg.add(s1, P.qf[0], P.x[0]);
        reg.add(s2, P.qf[1], P.x[1]);
        reg.add(s3, P.qf[2], P.x[2]);
        reg.add(s6, P.qf[3], P.x[3]);
    } else {
        for (int qi = 0; qi < P.n; ++qi) {
            const int q = P.q[qi];
            const int nq = (w - 1) / q;
            T acc = (T)0;
            for (int j = 0; j < nq; ++j)
                for (int k = 0; k < nq; ++k) {
                    const TIN* blk = t + (j * q) * pitch + k * q;
                    TIN m = blk[0];
                    for (int a = 0; a < q; ++a)
                        for (int b = 0; b < q; ++b) m = fmax(m, blk[a * pitch + b]);
                    const T v = height_diff<TIN, T>(m, c);
                    if (q == 1) bad |= (v != v);
                    const T vc = clip_w<T>(v, wmax);
                    acc = (j == 0 && k == 0) ? vc : acc + vc;
                }
            reg.add(acc, P.qf[qi], P.x[qi]);
        }
    }
    const T d = reg.result(P.n_t, P.m_x, P.ss_xx);
    out[y * W + x] = bad ? (T)NAN : d;
}

// ---- host side: float16 regression constants exactly as NumPy produces them (window.py:362-393) ---------------------
static float f16r(float v) { return (float)(_Float16)v; }

// np.add.reduce over a contiguous float buffer: 8 running accumulators for n >= 8, plain loop below
static float np_sum_f32(const float* a, int n) {
    if (n < 8) {
        float r = 0.f;  // (NumPy starts from the first element; 0 + a0 is exact)
        for (int i = 0; i < n; ++i) r = (i == 0) ? a[0] : r + a[i];
        return r;
    }
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

int fractal_constants(int w, int* qs, double* x, double* m_x, double* ss_xx) {
    const int hw = w / 2;
    int n = 0;
    float xf[FR_MAXQ], xx[FR_MAXQ];
    for (int q = 1; q <= hw && n < FR_MAXQ; ++q)
        if (hw % q == 0) {
            qs[n] = q;
            xf[n] = f16r(logf((float)q));      // np.log(uint8 array) -> float16
            xx[n] = f16r(xf[n] * xf[n]);       // float16 multiply
            x[n] = xf[n];
            ++n;
        }
    if (n == 0) { *m_x = NAN; *ss_xx = NAN; return 0; }
    const float mx = f16r(np_sum_f32(xf, n) / (float)n);          // np.mean: float32 accumulate, float16 result
    const float sxx = f16r(np_sum_f32(xx, n));                     // np.sum of float16: float32 accumulate
    const float nm = f16r(f16r((float)n * mx) * mx);               // n * m_x * m_x in float16
    *m_x = mx;
    *ss_xx = f16r(sxx - nm);
    return n;
}

template <typename TIN, typename T>
static int launch_extra_typed(xdemhip_ctx* ctx, const TerrainLaunch& L) {
    const TIN* dem = static_cast<const TIN*>(L.dem);
    const dim3 grid((unsigned)((L.W + 63) / 64), (unsigned)((L.H + 3) / 4));
    if (L.H > (int64_t)4 * 65535) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "raster too tall for one launch of the windowed kernels");
    if (L.planes[P_RUGOSITY_IDX]) {
        RugParams<T> P;
        const double res = L.resolution;
        const T dl_s = (T)(1.0 * res), dl_d = (T)(sqrt(2.0) * res);
        P.dl2_straight = dl_s * dl_s;
        P.dl2_diag = dl_d * dl_d;
        P.inv_area_den = (T)(res * res);
        hipLaunchKernelGGL((rugosity_kernel<TIN, T>), grid, dim3(256), 0, ctx->stream, dem, L.H, L.W, L.row_stride,
                           L.halo_top, L.halo_bottom, P, static_cast<T*>(L.planes[P_RUGOSITY_IDX]));
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    if (L.planes[P_FRACTAL_IDX]) {
        FracParams<T> P;
        memset(&P, 0, sizeof P);
        const int w = L.window_size;
        int qs[FR_MAXQ];
        double x[FR_MAXQ], mx, ssxx;
        P.w = w;
        P.n = fractal_constants(w, qs, x, &mx, &ssxx);
        for (int i = 0; i < P.n; ++i) { P.q[i] = qs[i]; P.qf[i] = (T)qs[i]; P.x[i] = (T)x[i]; }
        P.n_t = (T)P.n; P.m_x = (T)mx; P.ss_xx = (T)ssxx;
        const size_t lds = (size_t)(FR_COLS + w - 1) * (size_t)(FR_ROWS + w - 1) * sizeof(TIN);
        if (lds > 64 * 1024) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "fractal roughness window too large for the LDS tile");
        T* out = static_cast<T*>(L.planes[P_FRACTAL_IDX]);
        if (w == 13)
            hipLaunchKernelGGL((fractal_kernel<13, TIN, T>), grid, dim3(256), lds, ctx->stream, dem, L.H, L.W,
                               L.row_stride, L.halo_top, L.halo_bottom, P, out);
        else
            hipLaunchKernelGGL((fractal_kernel<0, TIN, T>), grid, dim3(256), lds, ctx->stream, dem, L.H, L.W,
                               L.row_stride, L.halo_top, L.halo_bottom, P, out);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    return XDEMHIP_OK;
}

int launch_window_extra(xdemhip_ctx* ctx, const TerrainLaunch& L) {
    if (L.dem_dtype == XDEMHIP_F32 && L.out_dtype == XDEMHIP_F32) return launch_extra_typed<float, float>(ctx, L);
    if (L.dem_dtype == XDEMHIP_F64 && L.out_dtype == XDEMHIP_F64) return launch_extra_typed<double, double>(ctx, L);
    if (L.dem_dtype == XDEMHIP_F32 && L.out_dtype == XDEMHIP_F64) return launch_extra_typed<float, double>(ctx, L);
    if (L.dem_dtype == XDEMHIP_F64 && L.out_dtype == XDEMHIP_F32) return launch_extra_typed<double, float>(ctx, L);
    return xd_fail(ctx, XDEMHIP_EINVAL, "unsupported dtype combination");
}

}  // namespace xd
