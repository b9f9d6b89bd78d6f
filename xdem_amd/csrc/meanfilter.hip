// meanfilter.hip -- NaN-ignoring mean filter with a square or circular kernel: the dense step of the
// patches method.  Replaces mean_filter_nan (xdem/spatialstats.py:2597-2655), i.e. its two
// scipy.ndimage.convolve(..., mode="constant", cval=nan) calls (spatialstats.py:2512-2525, 2626-2644):
//   * the kernel is ones((p, p)) or the p x p mask of _create_circular_mask (spatialstats.py:880-904), uint8; a true
//     convolution: tap (a, b) reads pixel (r + p/2 - a, c + p/2 - b); zero weights are skipped;
//   * SUM image: non-finite pixels count as 0, float64 accumulation over the window in row-major image order (SciPy walks
//     the flipped kernel), rounded to the image dtype, widened to float64; a non-zero tap outside the raster adds the NaN
//     border value;
//   * COUNT image: int8 ones (0 where non-finite) into an int8 result -- a border window's NaN accumulator casts to 0;
//     beyond 127 kernel pixels the reference's int8 count wraps (-112 for a 12 x 12 square), so such kernels are refused;
//   * mean = sum / count in float64 (0 / 0 -> NaN).
// One workgroup = a 64 x 16 block of output pixels; the block plus its window margin is staged in LDS once (value + a
// state byte: finite / nodata / outside the raster) and every thread walks the taps of its 4 pixels in SciPy's order, so
// the float64 sums are the reference's bit for bit.  HBM traffic: one read of the image, 16 B written per pixel.
#include "common.h"

#include <math.h>

#include <vector>

namespace xd {

constexpr int MF_TX = 64, MF_TY = 16, MF_MAXP = 13, MF_MAXTAPS = 127;
struct MfTaps {
    int n, lo, hi;                // taps, smallest / largest offset over both axes
    signed char dy[MF_MAXTAPS + 1], dx[MF_MAXTAPS + 1];
};

template <typename T>
__global__ __launch_bounds__(256) void mean_filter_kernel(const T* __restrict__ img, int64_t H, int64_t W, MfTaps taps,
                                                          double* __restrict__ mean_out, double* __restrict__ nvalid_out) {
    constexpr int PW = MF_TX + 2 * MF_MAXP, PH = MF_TY + 2 * MF_MAXP;
    __shared__ double s_val[PH * PW];          // pixel value, non-finite -> 0
    __shared__ signed char s_state[PH * PW];   // 1 finite, 0 nodata, -1 outside the raster
    const int64_t x0 = (int64_t)blockIdx.x * MF_TX, y0 = (int64_t)blockIdx.y * MF_TY;
    const int lo = taps.lo, span = taps.hi - taps.lo + 1;   // LDS patch: rows / columns lo .. TY - 1 + hi
    const int pw = MF_TX + span - 1, ph = MF_TY + span - 1;
    for (int k = threadIdx.x; k < pw * ph; k += 256) {
        const int r = k / pw, c = k - r * pw;
        const int64_t gy = y0 + lo + r, gx = x0 + lo + c;
        double v = 0.0;
        signed char st = -1;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const T t = img[gy * W + gx];
            const bool fin = fabs((double)t) <= 1.79769313486231570e308;   // np.isfinite
            v = fin ? (double)t : 0.0;
            st = fin ? 1 : 0;
        }
        s_val[r * PW + c] = v;
        s_state[r * PW + c] = st;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
#pragma unroll 1
    for (int q = 0; q < MF_TY / 4; ++q) {
        const int ty = ty0 + 4 * q;
        const int64_t gy = y0 + ty, gx = x0 + tx;
        if (gy >= H || gx >= W) continue;
        double s = 0.0;
        int n = 0;
        bool outside = false;
        const int base = (ty - lo) * PW + (tx - lo);
        for (int k = 0; k < taps.n; ++k) {   // SciPy's order: one float64 add per non-zero tap
            const int o = base + taps.dy[k] * PW + taps.dx[k];
            const signed char st = s_state[o];
            s = s + s_val[o];
            outside |= st < 0;
            n += st > 0;
        }
        const double sum = outside ? (double)NAN : (double)(T)s;
        const double cnt = outside ? 0.0 : (double)n;
        mean_out[gy * W + gx] = sum / cnt;
        nvalid_out[gy * W + gx] = cnt;
    }
}

// kernel mask -> tap list in accumulation order; returns the number of non-zero kernel pixels
static int build_taps(int p, int circular, MfTaps& t) {
    std::vector<unsigned char> k((size_t)p * p, 1);
    if (circular) {
        // _create_circular_mask((p, p)): centre (p / 2, p / 2), radius min(centre, p - centre) = p / 2, dist < radius
        const int c = p / 2;
        const double radius = (double)(c < p - c ? c : p - c);
        for (int y = 0; y < p; ++y)
            for (int x = 0; x < p; ++x) k[(size_t)y * p + x] = sqrt((double)((x - c) * (x - c) + (y - c) * (y - c))) < radius ? 1 : 0;
    }
    int n = 0;
    for (int i = 0; i < p * p; ++i) n += k[i];
    if (n > MF_MAXTAPS) return n;
    t.n = 0; t.lo = 0; t.hi = 0;
    // increasing image offset: dy = p/2 - a ascending <=> a descending; same for b
    for (int a = p - 1; a >= 0; --a)
        for (int b = p - 1; b >= 0; --b)
            if (k[(size_t)a * p + b]) {
                const int dy = p / 2 - a, dx = p / 2 - b;
                t.dy[t.n] = (signed char)dy; t.dx[t.n] = (signed char)dx;
                ++t.n;
                t.lo = dy < t.lo ? dy : t.lo; t.lo = dx < t.lo ? dx : t.lo;
                t.hi = dy > t.hi ? dy : t.hi; t.hi = dx > t.hi ? dx : t.hi;
            }
    return n;
}

}  // namespace xd

extern "C" int xdemhip_mean_filter_nan(xdemhip_ctx* ctx, const void* img, int dtype, int64_t H, int64_t W, int kernel_size,
                                       int kernel_shape, double* mean_out, double* nvalid_out, int* n_kernel_px, int memspace) {
    using namespace xd;
    if (!ctx) return XDEMHIP_EINVAL;
    if (!img || !mean_out || !nvalid_out || H < 1 || W < 1) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (dtype != XDEMHIP_F32 && dtype != XDEMHIP_F64) return xd_fail(ctx, XDEMHIP_EINVAL, "dtype must be float32 or float64");
    if (kernel_shape != 0 && kernel_shape != 1) return xd_fail(ctx, XDEMHIP_EINVAL, "kernel_shape: 0 square, 1 circular");
    if (kernel_size < 1) return xd_fail(ctx, XDEMHIP_EINVAL, "kernel_size must be >= 1");
    MfTaps taps;
    const int npx = kernel_size <= MF_MAXP + 1 ? build_taps(kernel_size, kernel_shape, taps) : MF_MAXTAPS + 1;
    if (n_kernel_px) *n_kernel_px = npx;
    if (npx > MF_MAXTAPS || kernel_size > MF_MAXP)
        return xd_fail(ctx, XDEMHIP_EUNSUPPORTED,
                       "mean_filter_nan: kernels of more than 127 pixels are refused -- the reference counts valid pixels in an int8 image, "
                       "which wraps beyond 127 (xdem/spatialstats.py:2637-2646)");
    if (npx == 0) {   // (p = 1 circular: an empty mask) every window is empty: sum 0, count 0
        taps.n = 0; taps.lo = taps.hi = 0;
    }
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t es = dtype == XDEMHIP_F32 ? 4 : 8, n = (size_t)H * (size_t)W;
    void* d_img = const_cast<void*>(img);
    double *d_mean = mean_out, *d_nv = nvalid_out;
    if (memspace == XDEMHIP_HOST) {
        d_img = nullptr; d_mean = d_nv = nullptr;
        if (hipMalloc(&d_img, n * es) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&d_mean), n * 8) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&d_nv), n * 8) != hipSuccess) {
            if (d_img) (void)hipFree(d_img);
            if (d_mean) (void)hipFree(d_mean);
            return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
        }
        (void)hipMemcpyAsync(d_img, img, n * es, hipMemcpyHostToDevice, ctx->stream);
    }
    const dim3 grid((unsigned)((W + MF_TX - 1) / MF_TX), (unsigned)((H + MF_TY - 1) / MF_TY));
    XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    if (dtype == XDEMHIP_F32)
        hipLaunchKernelGGL((mean_filter_kernel<float>), grid, dim3(256), 0, ctx->stream, static_cast<const float*>(d_img), H, W, taps, d_mean, d_nv);
    else
        hipLaunchKernelGGL((mean_filter_kernel<double>), grid, dim3(256), 0, ctx->stream, static_cast<const double*>(d_img), H, W, taps, d_mean, d_nv);
    (void)hipEventRecord(ctx->ev_stop, ctx->stream);
    ctx->timed = true;
    int rc = XDEMHIP_OK;
    if (hipGetLastError() != hipSuccess) rc = xd_fail(ctx, XDEMHIP_EHIP, "mean filter kernel launch failed");
    if (memspace == XDEMHIP_HOST) {
        if (rc == XDEMHIP_OK && (hipMemcpyAsync(mean_out, d_mean, n * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                                 hipMemcpyAsync(nvalid_out, d_nv, n * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                                 hipStreamSynchronize(ctx->stream) != hipSuccess))
            rc = xd_fail(ctx, XDEMHIP_EHIP, "mean filter kernel / D2H failed");
        (void)hipFree(d_img);
        (void)hipFree(d_mean);
        (void)hipFree(d_nv);
    }
    return rc;
}
