// convolve.hip -- the reference's generic image x filter convolution:
//     xdem.spatialstats.convolution(imgs, filters, method)          xdem/spatialstats.py:2558-2594
// n_N images (N1 x N2) against n_M filters (M1 x M2) -> float64 (n_N, n_M, N1, N2).  Two engines, both restated here:
//   method 0 "scipy"  (_scipy_convolution, spatialstats.py:2512-2525): scipy.ndimage.convolve(img, filter, mode="constant",
//     cval=nan) per (image, filter).  A true convolution, out[r, c] = sum_ab k[a, b] * img[r + M1/2 - a, c + M2/2 - b] (the
//     same centre for even sizes: SciPy flips the kernel and moves its origin by one); a double accumulator starts at 0 and
//     adds weight * value over the taps whose |weight| > DBL_EPSILON in row-major order of the FLIPPED kernel (= increasing
//     image offset); taps outside the image read the NaN border value; the sum is rounded to the IMAGE dtype and then widened
//     into the float64 output.  A non-finite pixel under a zero weight therefore leaves the output alone.
//   method 1 "numba"  (_numba_convolution, spatialstats.py:2528-2555, on the NaN-padded images of 2582-2585): a correlation
//     (no flip), out[r, c] = sum_ab img[r - (M1-1)/2 + a, c - (M2-1)/2 + b] * k[a, b] over EVERY tap in row-major order, zero
//     weights included (0 x Inf = NaN arises from the arithmetic), no rounding to the image dtype; with an even filter size the
//     padded image is one row / column short, so the last output row / column keeps the zeros it was initialised with.
// No fused multiply-add anywhere (this unit is compiled with -ffp-contract=off): the products are rounded before they are
// added, as SciPy's C loop and Numba's LLVM code (no fastmath) do.
// One workgroup = a 64 x 16 block of output pixels of one image; the block plus the filter margin is staged in LDS once
// (outside the image: NaN) and every thread walks the taps of its 4 pixels, filter after filter -- the image is read once
// for all filters.  HBM traffic: sizeof(T) read + 8 * n_M written per pixel.  Filters whose LDS patch would exceed 64 KiB
// take the same loop on global memory (bounds tested per tap).
// The sizes the reference itself uses (3 x 3 and 5 x 5 stencil tables, surfit.py:1107; 7 x 7) take convolve_window_kernel: a
// thread owns RP = 4 CONSECUTIVE rows of one column, reads the (RP + M1 - 1) x M2 window they share from LDS once -- 40 reads for
// 5 x 5 instead of 100 per filter -- and keeps it in registers as float64 for all filters; the tap loops are unrolled, weights
// and the per-filter bit mask of the taps that count (SciPy's footprint) are wave-uniform scalars.  Same sums in the same order.
#include "common.h"

#include <float.h>
#include <math.h>

#include <vector>

namespace xd {

constexpr int CV_TX = 64, CV_TY = 16;
constexpr size_t CV_LDS_MAX = 64 * 1024;

struct CvArgs {
    int64_t H, W;
    int n_f, dy_min, dx_min, M1, M2, round_to_t;
    int64_t vr, vc;         // rows / columns that receive a sum (the rest stays 0: the Numba engine with even filter sizes)
};

// (the tap lists and the output are top-level __restrict__ parameters: only then are the wave-uniform tap reads scalar loads that
// the compiler may issue ahead of the stores -- as members of the argument block they came out as one vector load + full wait per tap)
// tap_yx: (dy, dx) image offset of a tap relative to the output pixel; taps of filter f: [f_start[f], f_start[f + 1]); out: (n_f, H, W)
template <typename T, bool LDS>
__global__ __launch_bounds__(256) void convolve_kernel(const T* __restrict__ img, const int2* __restrict__ tap_yx,
                                                       const double* __restrict__ tap_w, const int* __restrict__ f_start,
                                                       double* __restrict__ out, CvArgs a) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    T* s = reinterpret_cast<T*>(s_raw);
    const int64_t x0 = (int64_t)blockIdx.x * CV_TX, y0 = (int64_t)blockIdx.y * CV_TY;
    const int pw = CV_TX + a.M2 - 1, ph = CV_TY + a.M1 - 1;
    const T nan_t = (T)NAN;
    if (LDS) {
        for (int k = threadIdx.x; k < pw * ph; k += 256) {
            const int r = k / pw, c = k - r * pw;
            const int64_t gy = y0 + a.dy_min + r, gx = x0 + a.dx_min + c;
            s[k] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? img[gy * a.W + gx] : nan_t;
        }
        __syncthreads();
    }
    const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
    const int64_t gx = x0 + tx;
    const int64_t plane = a.H * a.W;
    for (int f = 0; f < a.n_f; ++f) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        const int k0 = f_start[f], k1 = f_start[f + 1];
        for (int k = k0; k < k1; ++k) {
            const int2 yx = tap_yx[k];
            const double w = tap_w[k];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ty = ty0 + 4 * q;
                T v;
                if (LDS) {
                    v = s[(ty + yx.x - a.dy_min) * pw + (tx + yx.y - a.dx_min)];
                } else {
                    const int64_t yy = y0 + ty + yx.x, xx = gx + yx.y;
                    v = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? img[yy * a.W + xx] : nan_t;
                }
                acc[q] = acc[q] + (double)v * w;
            }
        }
        if (gx < a.W) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t gy = y0 + ty0 + 4 * q;
                if (gy >= a.H) continue;
                const double r = a.round_to_t ? (double)(T)acc[q] : acc[q];
                out[(int64_t)f * plane + gy * a.W + gx] = (gy < a.vr && gx < a.vc) ? r : 0.0;
            }
        }
    }
}

// Small filters of compile-time size: the window of a thread's 4 consecutive rows lives in registers.
// w: (n_f, M1, M2) weights in TAP order (SciPy: the flipped kernel), zeros included; mask: per filter, bit a * M2 + b set = the tap
// takes part in the sum
template <typename T, int M1, int M2, int RP>
__global__ __launch_bounds__(256) void convolve_window_kernel(const T* __restrict__ img, const double* __restrict__ w_all,
                                                              const unsigned long long* __restrict__ mask,
                                                              double* __restrict__ out, CvArgs a) {
    constexpr int TY = 4 * RP, PW = CV_TX + M2 - 1, PH = TY + M1 - 1;   // RP consecutive rows per thread, 4 row groups per workgroup
    __shared__ T s[PH * PW];
    const int64_t x0 = (int64_t)blockIdx.x * CV_TX, y0 = (int64_t)blockIdx.y * TY;
    const T nan_t = (T)NAN;
    for (int k = threadIdx.x; k < PW * PH; k += 256) {
        const int r = k / PW, c = k - r * PW;
        const int64_t gy = y0 + a.dy_min + r, gx = x0 + a.dx_min + c;
        s[k] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? img[gy * a.W + gx] : nan_t;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * RP;
    double win[RP + M1 - 1][M2];
#pragma unroll
    for (int r = 0; r < RP + M1 - 1; ++r)
#pragma unroll
        for (int b = 0; b < M2; ++b) win[r][b] = (double)s[(r0 + r) * PW + tx + b];
    const int64_t gx = x0 + tx, plane = a.H * a.W;
    for (int f = 0; f < a.n_f; ++f) {
        const unsigned long long m = mask[f];
        const double* __restrict__ wf = w_all + (size_t)f * (M1 * M2);
        double acc[RP];
#pragma unroll
        for (int q = 0; q < RP; ++q) acc[q] = 0.0;
        // weights as wave-uniform scalars: a whole filter up front where it fits the scalar registers (<= 25 taps: one wait per filter),
        // else one tap row at a time (one wait per row)
        constexpr int WB = (M1 * M2 <= 25) ? M1 : 1;   // tap rows per batch of weights
#pragma unroll
        for (int r0w = 0; r0w < M1; r0w += WB) {
            double wr[WB * M2];
#pragma unroll
            for (int t = 0; t < WB * M2; ++t) wr[t] = wf[r0w * M2 + t];
#pragma unroll
            for (int t = 0; t < WB * M2; ++t) {
                const int r = r0w + t / M2, b = t % M2;
                if ((m >> (r * M2 + b)) & 1ull) {   // (wave-uniform)
#pragma unroll
                    for (int q = 0; q < RP; ++q) acc[q] = acc[q] + win[q + r][b] * wr[t];
                }
            }
        }
        if (gx < a.W) {
#pragma unroll
            for (int q = 0; q < RP; ++q) {
                const int64_t gy = y0 + r0 + q;
                if (gy >= a.H) continue;
                const double r = a.round_to_t ? (double)(T)acc[q] : acc[q];
                out[(int64_t)f * plane + gy * a.W + gx] = (gy < a.vr && gx < a.vc) ? r : 0.0;
            }
        }
    }
}

// rows per thread: 4 for all three sizes.  Two rows for 5 x 5 (78 instead of 104 registers: six waves per SIMD instead of four) were
// measured SLOWER, 4.6 against 3.5 ms for the five Florinsky tables at 16384^2 -- the shared window shrinks from 8 to 6 rows but serves
// half the pixels (15 instead of 10 LDS reads per pixel), and tile margin, weight loads and tap branches are paid per 8 rows instead of 16
// (profiles/r06ao_conv_probe.txt)
constexpr int cv_rows_per_thread(int) { return 4; }

template <typename T>
static bool launch_window(int M1, int M2, int64_t H, int64_t W, hipStream_t st, const T* src, const double* w, const unsigned long long* mask,
                          double* out, const CvArgs& a) {
    auto grid = [&](int rp) { return dim3((unsigned)((W + CV_TX - 1) / CV_TX), (unsigned)((H + 4 * rp - 1) / (4 * rp))); };
    if (M1 == 3 && M2 == 3) hipLaunchKernelGGL((convolve_window_kernel<T, 3, 3, cv_rows_per_thread(3)>), grid(cv_rows_per_thread(3)), dim3(256), 0, st, src, w, mask, out, a);
    else if (M1 == 5 && M2 == 5) hipLaunchKernelGGL((convolve_window_kernel<T, 5, 5, cv_rows_per_thread(5)>), grid(cv_rows_per_thread(5)), dim3(256), 0, st, src, w, mask, out, a);
    else if (M1 == 7 && M2 == 7) hipLaunchKernelGGL((convolve_window_kernel<T, 7, 7, cv_rows_per_thread(7)>), grid(cv_rows_per_thread(7)), dim3(256), 0, st, src, w, mask, out, a);
    else return false;
    return true;
}

}  // namespace xd

extern "C" int xdemhip_convolution(xdemhip_ctx* ctx, const void* imgs, int dtype, int64_t n_img, int64_t H, int64_t W,
                                   const double* filters, int n_f, int M1, int M2, int method, double* out, int memspace) {
    using namespace xd;
    if (!ctx) return XDEMHIP_EINVAL;
    if (!imgs || !filters || !out || n_img < 1 || H < 1 || W < 1 || n_f < 1 || M1 < 1 || M2 < 1)
        return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (dtype != XDEMHIP_F32 && dtype != XDEMHIP_F64) return xd_fail(ctx, XDEMHIP_EINVAL, "dtype must be float32 or float64");
    if (method != 0 && method != 1) return xd_fail(ctx, XDEMHIP_EINVAL, "method: 0 scipy, 1 numba");
    if (M1 > 4096 || M2 > 4096) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "filters of more than 4096 rows / columns are refused");
    // tap lists in the engine's accumulation order
    std::vector<int2> yx;
    std::vector<double> wt;
    std::vector<int> start(1, 0);
    std::vector<double> dense((size_t)n_f * M1 * M2);        // weights in tap order, zeros included (the window kernels)
    std::vector<unsigned long long> mask((size_t)n_f, 0ull);
    const bool windowed = (M1 == M2) && (M1 == 3 || M1 == 5 || M1 == 7);
    const int dy_min = method == 0 ? -(M1 - 1 - M1 / 2) : -((M1 - 1) / 2);
    const int dx_min = method == 0 ? -(M2 - 1 - M2 / 2) : -((M2 - 1) / 2);
    for (int f = 0; f < n_f; ++f) {
        const double* k = filters + (size_t)f * M1 * M2;
        for (int a = 0; a < M1; ++a)
            for (int b = 0; b < M2; ++b) {
                // scipy: the flipped kernel in row-major order; numba: the kernel as it stands
                const double w = method == 0 ? k[(size_t)(M1 - 1 - a) * M2 + (M2 - 1 - b)] : k[(size_t)a * M2 + b];
                dense[((size_t)f * M1 + a) * M2 + b] = w;
                if (method == 0 && !(fabs(w) > DBL_EPSILON)) continue;   // (a NaN weight fails the test too, as in SciPy's footprint)
                if (windowed) mask[f] |= 1ull << (a * M2 + b);
                yx.push_back(make_int2(dy_min + a, dx_min + b));
                wt.push_back(w);
            }
        start.push_back((int)yx.size());
    }
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t es = dtype == XDEMHIP_F32 ? 4 : 8, n = (size_t)H * (size_t)W;
    const size_t ntap = yx.size() ? yx.size() : 1;
    int2* d_yx = nullptr;
    double *d_w = nullptr, *d_dense = nullptr;
    unsigned long long* d_mask = nullptr;
    int* d_start = nullptr;
    void* d_img = nullptr;
    double* d_out = nullptr;
    auto release = [&]() {
        if (d_yx) (void)hipFree(d_yx);
        if (d_w) (void)hipFree(d_w);
        if (d_start) (void)hipFree(d_start);
        if (d_dense) (void)hipFree(d_dense);
        if (d_mask) (void)hipFree(d_mask);
        if (memspace == XDEMHIP_HOST) {
            if (d_img) (void)hipFree(d_img);
            if (d_out) (void)hipFree(d_out);
        }
    };
    if (hipMalloc(reinterpret_cast<void**>(&d_yx), ntap * sizeof(int2)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d_w), ntap * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d_start), start.size() * sizeof(int)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d_dense), dense.size() * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d_mask), mask.size() * sizeof(unsigned long long)) != hipSuccess) {
        release();
        return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    }
    if ((yx.size() && (hipMemcpy(d_yx, yx.data(), yx.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess ||
                       hipMemcpy(d_w, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)) ||
        hipMemcpy(d_start, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_dense, dense.data(), dense.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_mask, mask.data(), mask.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess) {
        release();
        return xd_fail(ctx, XDEMHIP_EHIP, "upload of the filter taps failed");
    }
    if (memspace == XDEMHIP_HOST) {   // one image and its n_f planes on the device at a time
        if (hipMalloc(&d_img, n * es) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&d_out), n * 8 * (size_t)n_f) != hipSuccess) {
            release();
            return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
        }
    }
    CvArgs a;
    a.H = H; a.W = W; a.n_f = n_f;
    a.dy_min = dy_min; a.dx_min = dx_min; a.M1 = M1; a.M2 = M2; a.round_to_t = method == 0 ? 1 : 0;
    a.vr = (method == 1 && !(M1 & 1)) ? H - 1 : H;
    a.vc = (method == 1 && !(M2 & 1)) ? W - 1 : W;
    const size_t lds = (size_t)(CV_TX + M2 - 1) * (size_t)(CV_TY + M1 - 1) * es;
    const bool use_lds = lds <= CV_LDS_MAX;
    const dim3 grid((unsigned)((W + CV_TX - 1) / CV_TX), (unsigned)((H + CV_TY - 1) / CV_TY));
    int rc = XDEMHIP_OK;
    (void)hipEventRecord(ctx->ev_start, ctx->stream);
    for (int64_t i = 0; i < n_img && rc == XDEMHIP_OK; ++i) {
        const void* src = static_cast<const unsigned char*>(imgs) + (size_t)i * n * es;
        double* dst = out + (size_t)i * (size_t)n_f * n;
        if (memspace == XDEMHIP_HOST) {
            if (hipMemcpyAsync(d_img, src, n * es, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
                rc = xd_fail(ctx, XDEMHIP_EHIP, "H2D copy of an image failed");
                break;
            }
            src = d_img;
        }
        double* o = memspace == XDEMHIP_HOST ? d_out : dst;
        if (windowed) {
            if (dtype == XDEMHIP_F32) launch_window<float>(M1, M2, H, W, ctx->stream, static_cast<const float*>(src), d_dense, d_mask, o, a);
            else launch_window<double>(M1, M2, H, W, ctx->stream, static_cast<const double*>(src), d_dense, d_mask, o, a);
        } else if (dtype == XDEMHIP_F32) {
            if (use_lds) hipLaunchKernelGGL((convolve_kernel<float, true>), grid, dim3(256), lds, ctx->stream, static_cast<const float*>(src), d_yx, d_w, d_start, o, a);
            else hipLaunchKernelGGL((convolve_kernel<float, false>), grid, dim3(256), 0, ctx->stream, static_cast<const float*>(src), d_yx, d_w, d_start, o, a);
        } else {
            if (use_lds) hipLaunchKernelGGL((convolve_kernel<double, true>), grid, dim3(256), lds, ctx->stream, static_cast<const double*>(src), d_yx, d_w, d_start, o, a);
            else hipLaunchKernelGGL((convolve_kernel<double, false>), grid, dim3(256), 0, ctx->stream, static_cast<const double*>(src), d_yx, d_w, d_start, o, a);
        }
        if (hipGetLastError() != hipSuccess) {
            rc = xd_fail(ctx, XDEMHIP_EHIP, "convolution kernel launch failed");
            break;
        }
        if (memspace == XDEMHIP_HOST &&
            (hipMemcpyAsync(dst, d_out, n * 8 * (size_t)n_f, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
             hipStreamSynchronize(ctx->stream) != hipSuccess))
            rc = xd_fail(ctx, XDEMHIP_EHIP, "convolution kernel / D2H failed");
    }
    (void)hipEventRecord(ctx->ev_stop, ctx->stream);
    ctx->timed = true;
    // the tap lists are read by launches that may still run: wait for them before the lists go (hipFree would wait as well)
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == XDEMHIP_OK) rc = xd_fail(ctx, XDEMHIP_EHIP, "convolution kernel failed");
    release();
    return rc;
}
