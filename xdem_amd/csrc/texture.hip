// texture.hip -- texture shading (Brown 2010): fractional Laplacian |f|^alpha applied in the frequency domain.
//
// Replaces  _texture_shading_fft(dem, alpha)   xdem/terrain/freq.py:63-148  (called from terrain.py:637-644):
//   non-finite pixels filled with the NaN-ignoring mean (freq.py:88-96), symmetric padding to the next 2/3/5/7-smooth size,
//   centred (98-117), rfft2, multiply by hypot(fx, fy)^alpha with the DC term zeroed when alpha > 0 (119-137), irfft2, crop,
//   NaN restored where the input was invalid (140-146).
// The transform is the library's job (hipFFT, loaded lazily with dlopen so that nothing else in libxdemhip.so depends on
// it); it runs in the DEM's precision like scipy.fft does.  Fill / pad, spectral filter and crop are streaming kernels.
#include <dlfcn.h>
#include <math.h>

#include "common.h"

namespace xd {

// ---- minimal hipFFT binding (hipfft/hipfft.h: hipfftPlan2d, hipfftExec*, hipfftSetStream, hipfftDestroy) ---------------
typedef struct hipfftHandle_t* fft_handle;
enum { FFT_R2C = 0x2a, FFT_C2R = 0x2c, FFT_D2Z = 0x6a, FFT_Z2D = 0x6c };
struct FftApi {
    int (*plan2d)(fft_handle*, int, int, int) = nullptr;
    int (*set_stream)(fft_handle, hipStream_t) = nullptr;
    int (*exec_r2c)(fft_handle, float*, void*) = nullptr;
    int (*exec_c2r)(fft_handle, void*, float*) = nullptr;
    int (*exec_d2z)(fft_handle, double*, void*) = nullptr;
    int (*exec_z2d)(fft_handle, void*, double*) = nullptr;
    int (*destroy)(fft_handle) = nullptr;
    bool ok = false;
};

static FftApi& fft_api() {
    static FftApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    void* h = dlopen("libhipfft.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libhipfft.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/libhipfft.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return api;
    api.plan2d = reinterpret_cast<decltype(api.plan2d)>(dlsym(h, "hipfftPlan2d"));
    api.set_stream = reinterpret_cast<decltype(api.set_stream)>(dlsym(h, "hipfftSetStream"));
    api.exec_r2c = reinterpret_cast<decltype(api.exec_r2c)>(dlsym(h, "hipfftExecR2C"));
    api.exec_c2r = reinterpret_cast<decltype(api.exec_c2r)>(dlsym(h, "hipfftExecC2R"));
    api.exec_d2z = reinterpret_cast<decltype(api.exec_d2z)>(dlsym(h, "hipfftExecD2Z"));
    api.exec_z2d = reinterpret_cast<decltype(api.exec_z2d)>(dlsym(h, "hipfftExecZ2D"));
    api.destroy = reinterpret_cast<decltype(api.destroy)>(dlsym(h, "hipfftDestroy"));
    api.ok = api.plan2d && api.set_stream && api.exec_r2c && api.exec_c2r && api.exec_d2z && api.exec_z2d && api.destroy;
    return api;
}

// next FFT length of freq.py:32-60: power of two up to 1024, else the next 7-smooth integer
static int64_t nextprod_fft(int64_t n) {
    if (n <= 1) return 1;
    if (n <= 1024) {
        int64_t p = 1;
        while (p < n) p <<= 1;
        return p;
    }
    for (int64_t c = n;; ++c) {
        int64_t t = c;
        for (int f : {2, 3, 5, 7})
            while (t % f == 0) t /= f;
        if (t == 1) return c;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void tex_stats_kernel(const T* dem, int64_t n, double* sum, unsigned long long* counts /* [non-NaN, finite] */) {
    double s = 0.0;
    unsigned long long c_nn = 0, c_fin = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const T v = dem[p];
        if (v == v) { s += (double)v; ++c_nn; }                      // np.nanmean skips NaN only: +-Inf poison the mean, as upstream
        if (fabs((double)v) <= 1.79769313486231570e308) ++c_fin;
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off); c_nn += __shfl_down(c_nn, off); c_fin += __shfl_down(c_fin, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(sum, s); atomicAdd(&counts[0], c_nn); atomicAdd(&counts[1], c_fin); }
}

// padded[r][c] = filled DEM at the symmetric reflection of (r - pad_r, c - pad_c)   (np.pad(mode="symmetric"))
template <typename T>
__global__ __launch_bounds__(256) void tex_pad_kernel(const T* dem, int64_t H, int64_t W, int64_t FH, int64_t FW, int64_t pad_r,
                                                      int64_t pad_c, T fill, T* out) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= FW) return;
    int64_t jc = (c - pad_c) % (2 * W);
    if (jc < 0) jc += 2 * W;
    if (jc >= W) jc = 2 * W - 1 - jc;
    for (int64_t r = blockIdx.y; r < FH; r += gridDim.y) {
        int64_t jr = (r - pad_r) % (2 * H);
        if (jr < 0) jr += 2 * H;
        if (jr >= H) jr = 2 * H - 1 - jr;
        const T v = dem[jr * W + jc];
        out[r * FW + c] = (fabs((double)v) <= 1.79769313486231570e308) ? v : fill;
    }
}

// spectrum[r][c] *= hypot(fx, fy)^alpha / (FH * FW)   (the inverse transform of hipFFT is unnormalised)
template <typename T>
__global__ __launch_bounds__(256) void tex_filter_kernel(T* spec /* interleaved complex */, int64_t FH, int64_t FW, int64_t FC, double alpha) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= FC) return;
    const double fx = (double)c / (double)FW;                       // rfftfreq
    const double norm = 1.0 / ((double)FH * (double)FW);
    for (int64_t r = blockIdx.y; r < FH; r += gridDim.y) {
        const int64_t k = (r < (FH + 1) / 2) ? r : r - FH;            // fftfreq: 0 .. (n-1)//2, -(n//2) .. -1
        const double fy = (double)k / (double)FH;
        double mag = hypot(fx, fy);
        if (r == 0 && c == 0) mag = 1.0;
        double f = pow(mag, alpha);
        if (r == 0 && c == 0 && alpha > 0.0) f = 0.0;
        const double g = f * norm;
        T* z = spec + 2 * (r * FC + c);
        z[0] = (T)((double)z[0] * g);
        z[1] = (T)((double)z[1] * g);
    }
}

template <typename T, typename TOUT>
__global__ __launch_bounds__(256) void tex_crop_kernel(const T* padded, const T* dem, int64_t H, int64_t W, int64_t FW, int64_t pad_r,
                                                       int64_t pad_c, TOUT* out) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= W) return;
    for (int64_t r = blockIdx.y; r < H; r += gridDim.y) {
        const T v = dem[r * W + c];
        const bool valid = fabs((double)v) <= 1.79769313486231570e308;
        out[r * W + c] = valid ? (TOUT)padded[(r + pad_r) * FW + (c + pad_c)] : (TOUT)NAN;
    }
}

template <typename TOUT> __global__ void tex_fill_nan_kernel(TOUT* out, int64_t n) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) out[p] = (TOUT)NAN;
}

template <typename T, typename TOUT>
static int texture_typed(xdemhip_ctx* ctx, const T* d_dem, int64_t H, int64_t W, double alpha, TOUT* d_out) {
    FftApi& api = fft_api();
    if (!api.ok) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "libhipfft.so could not be loaded (needed by texture shading only)");
    const int64_t n = H * W;
    double* d_sum = nullptr;
    T *d_pad = nullptr, *d_spec = nullptr;
    fft_handle fwd = nullptr, inv = nullptr;
    auto cleanup = [&]() {
        if (fwd) api.destroy(fwd);
        if (inv) api.destroy(inv);
        if (d_sum) (void)hipFree(d_sum);
        if (d_pad) (void)hipFree(d_pad);
        if (d_spec) (void)hipFree(d_spec);
    };
    if (hipMalloc(reinterpret_cast<void**>(&d_sum), 24) != hipSuccess) return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    (void)hipMemsetAsync(d_sum, 0, 24, ctx->stream);
    const int g1 = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL((tex_stats_kernel<T>), dim3(g1), dim3(256), 0, ctx->stream, d_dem, n, d_sum,
                       reinterpret_cast<unsigned long long*>(d_sum + 1));
    struct { double sum; unsigned long long nn, fin; } st;
    hipError_t e = hipMemcpyAsync(&st, d_sum, 24, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { cleanup(); return xd_fail(ctx, XDEMHIP_EHIP, std::string("texture statistics failed: ") + hipGetErrorString(e)); }
    if (st.fin == 0) {  // no valid pixel: all NaN (freq.py:88-89)
        hipLaunchKernelGGL((tex_fill_nan_kernel<TOUT>), dim3(g1), dim3(256), 0, ctx->stream, d_out, n);
        cleanup();
        return XDEMHIP_OK;
    }
    const T fill = (T)(st.sum / (double)st.nn);  // np.nanmean in the DEM dtype (accumulated in float64 here)
    const int64_t FH = nextprod_fft(H), FW = nextprod_fft(W), FC = FW / 2 + 1;
    if (FH > 0x7fffffff || FW > 0x7fffffff) { cleanup(); return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "raster too large for the FFT"); }
    const int64_t pad_r = (FH - H) / 2, pad_c = (FW - W) / 2;
    if (hipMalloc(reinterpret_cast<void**>(&d_pad), (size_t)FH * FW * sizeof(T)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d_spec), (size_t)FH * FC * 2 * sizeof(T)) != hipSuccess) {
        cleanup();
        return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc(FFT buffers) failed");
    }
    auto rows_grid = [&](int64_t cols, int64_t rows) {
        int64_t gy = rows < 1024 ? rows : 1024;
        return dim3((unsigned)((cols + 255) / 256), (unsigned)(gy < 1 ? 1 : gy));
    };
    hipLaunchKernelGGL((tex_pad_kernel<T>), rows_grid(FW, FH), dim3(256), 0, ctx->stream, d_dem, H, W, FH, FW, pad_r, pad_c, fill, d_pad);
    const bool f32 = sizeof(T) == 4;
    if (api.plan2d(&fwd, (int)FH, (int)FW, f32 ? FFT_R2C : FFT_D2Z) != 0 || api.plan2d(&inv, (int)FH, (int)FW, f32 ? FFT_C2R : FFT_Z2D) != 0) {
        cleanup();
        return xd_fail(ctx, XDEMHIP_EHIP, "hipfftPlan2d failed");
    }
    api.set_stream(fwd, ctx->stream);
    api.set_stream(inv, ctx->stream);
    int rc = f32 ? api.exec_r2c(fwd, reinterpret_cast<float*>(d_pad), d_spec) : api.exec_d2z(fwd, reinterpret_cast<double*>(d_pad), d_spec);
    if (rc == 0) {
        hipLaunchKernelGGL((tex_filter_kernel<T>), rows_grid(FC, FH), dim3(256), 0, ctx->stream, d_spec, FH, FW, FC, alpha);
        rc = f32 ? api.exec_c2r(inv, d_spec, reinterpret_cast<float*>(d_pad)) : api.exec_z2d(inv, d_spec, reinterpret_cast<double*>(d_pad));
    }
    if (rc != 0) { cleanup(); return xd_fail(ctx, XDEMHIP_EHIP, "hipFFT execution failed"); }
    hipLaunchKernelGGL((tex_crop_kernel<T, TOUT>), rows_grid(W, H), dim3(256), 0, ctx->stream, d_pad, d_dem, H, W, FW, pad_r, pad_c, d_out);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return xd_fail(ctx, XDEMHIP_EHIP, std::string("texture shading failed: ") + hipGetErrorString(e));
    return XDEMHIP_OK;
}

}  // namespace xd

using namespace xd;

extern "C" int xdemhip_texture_shading(xdemhip_ctx* ctx, const void* dem, int dem_dtype, int64_t H, int64_t W, double alpha, int out_dtype,
                                       void* out, int memspace) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (!dem || !out || H <= 0 || W <= 0) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (!(alpha >= 0.0 && alpha <= 2.0)) return xd_fail(ctx, XDEMHIP_EINVAL, "Alpha must be between 0 and 2");
    if ((dem_dtype != XDEMHIP_F32 && dem_dtype != XDEMHIP_F64) || (out_dtype != XDEMHIP_F32 && out_dtype != XDEMHIP_F64))
        return xd_fail(ctx, XDEMHIP_EINVAL, "dtype must be XDEMHIP_F32 or XDEMHIP_F64");
    if (memspace != XDEMHIP_HOST && memspace != XDEMHIP_DEVICE) return xd_fail(ctx, XDEMHIP_EINVAL, "bad memspace");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t in_es = dem_dtype == XDEMHIP_F32 ? 4 : 8, out_es = out_dtype == XDEMHIP_F32 ? 4 : 8;
    const size_t n = (size_t)H * (size_t)W;
    void *d_dem = const_cast<void*>(dem), *d_out = out;
    if (memspace == XDEMHIP_HOST) {
        d_dem = d_out = nullptr;
        if (hipMalloc(&d_dem, n * in_es) != hipSuccess || hipMalloc(&d_out, n * out_es) != hipSuccess) {
            if (d_dem) (void)hipFree(d_dem);
            return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
        }
        if (hipMemcpyAsync(d_dem, dem, n * in_es, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
            (void)hipFree(d_dem); (void)hipFree(d_out);
            return xd_fail(ctx, XDEMHIP_EHIP, "H2D copy failed");
        }
    }
    int rc;
    if (dem_dtype == XDEMHIP_F32)
        rc = out_dtype == XDEMHIP_F32 ? texture_typed<float, float>(ctx, static_cast<const float*>(d_dem), H, W, alpha, static_cast<float*>(d_out))
                                      : texture_typed<float, double>(ctx, static_cast<const float*>(d_dem), H, W, alpha, static_cast<double*>(d_out));
    else
        rc = out_dtype == XDEMHIP_F32 ? texture_typed<double, float>(ctx, static_cast<const double*>(d_dem), H, W, alpha, static_cast<float*>(d_out))
                                      : texture_typed<double, double>(ctx, static_cast<const double*>(d_dem), H, W, alpha, static_cast<double*>(d_out));
    if (memspace == XDEMHIP_HOST) {
        if (rc == XDEMHIP_OK) {
            hipError_t e = hipMemcpyAsync(out, d_out, n * out_es, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = xd_fail(ctx, XDEMHIP_EHIP, std::string("D2H copy failed: ") + hipGetErrorString(e));
        }
        (void)hipFree(d_dem);
        (void)hipFree(d_out);
    }
    return rc;
}
