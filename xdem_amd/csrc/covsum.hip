// covsum.hip -- double sum of spatially correlated errors,  sum_i sum_j  e_i e_j rho(|x_i - x_j|),  on gfx950.
//
// Replaces the O(N^2) part of the "number of effective samples" estimators built on the fitted variogram models
// (callers of the variogram path):  neff_exact  xdem/spatialstats.py:2175-2236 (scipy pdist + squareform + a dense N x N
// product: 80 GB at N = 1e5) and  neff_hugonnet_approx  2239-2308 (N x subsample).  rho(h) = 1 - sum_m gamma_m(h) / total sill
// with the scikit-gstat model forms restated in xdem_amd/variogram_models.py (spherical, exponential, gaussian, cubic, stable;
// matern needs a modified Bessel function and is refused here).
// One workgroup = 256 A points (registers) x a 4096-point chunk of B streamed through LDS in 256-point tiles; float64
// throughout; per-thread partial sums, a workgroup tree reduction, one float64 atomic per workgroup.  VALU-bound
// (a square root and an exponential per pair).
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace xd {

constexpr int CV_NT = 256, CV_CHUNK = 4096, CV_MAXM = 8;
enum { CV_SPHERICAL = 0, CV_EXPONENTIAL = 1, CV_GAUSSIAN = 2, CV_CUBIC = 3, CV_STABLE = 4 };

struct CovModels {
    int n;
    int type[CV_MAXM];
    double r[CV_MAXM], c0[CV_MAXM], s[CV_MAXM];
    double inv_sill;
};

__device__ __forceinline__ double cov_rho(const CovModels& M, double h) {
    double g = 0.0;
    for (int m = 0; m < M.n; ++m) {
        const double r = M.r[m], c0 = M.c0[m];
        double v;
        switch (M.type[m]) {
            case CV_SPHERICAL: { const double x = h / r; v = h <= r ? c0 * (1.5 * x - 0.5 * x * x * x) : c0; break; }
            case CV_EXPONENTIAL: v = c0 * (1.0 - exp(-h / (r / 3.0))); break;
            case CV_GAUSSIAN: { const double a = r / 2.0; v = c0 * (1.0 - exp(-(h * h) / (a * a))); break; }
            case CV_CUBIC: {
                const double x = h / r, x2 = x * x;
                v = h < r ? c0 * (7.0 * x2 - 8.75 * x2 * x + 3.5 * x2 * x2 * x - 0.75 * x2 * x2 * x2 * x) : c0;
                break;
            }
            default: { const double a = r / pow(3.0, 1.0 / M.s[m]); v = c0 * (1.0 - exp(-pow(h / a, M.s[m]))); break; }
        }
        g += v;
    }
    return 1.0 - g * M.inv_sill;
}

__global__ __launch_bounds__(CV_NT) void cov_sum_kernel(const double* ax, const double* ay, const double* ae, int64_t na, const double* bx,
                                                        const double* by, const double* be, int64_t nb, CovModels M, double* out) {
    __shared__ double s_x[CV_NT], s_y[CV_NT], s_e[CV_NT], s_red[CV_NT / 64];
    const int64_t nchunk = (nb + CV_CHUNK - 1) / CV_CHUNK;
    const int64_t ta = blockIdx.x / nchunk, cb = blockIdx.x - ta * nchunk;
    const int tid = threadIdx.x;
    const int64_t ia = ta * CV_NT + tid;
    const bool have = ia < na;
    const double px = have ? ax[ia] : 0.0, py = have ? ay[ia] : 0.0, pe = have ? ae[ia] : 0.0;
    const int64_t j0 = cb * CV_CHUNK, j1 = (j0 + CV_CHUNK < nb) ? j0 + CV_CHUNK : nb;
    double acc = 0.0;
    for (int64_t j = j0; j < j1; j += CV_NT) {
        __syncthreads();
        const int cnt = (int)((j1 - j) < CV_NT ? (j1 - j) : CV_NT);
        if (tid < cnt) { s_x[tid] = bx[j + tid]; s_y[tid] = by[j + tid]; s_e[tid] = be[j + tid]; }
        __syncthreads();
        if (have)
            for (int k = 0; k < cnt; ++k) {
                const double dx = px - s_x[k], dy = py - s_y[k];
                acc += s_e[k] * cov_rho(M, sqrt(dx * dx + dy * dy));
            }
    }
    acc *= pe;
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((tid & 63) == 0) s_red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < CV_NT / 64; ++w) t += s_red[w];
        atomicAdd(out, t);
    }
}

}  // namespace xd

using namespace xd;

extern "C" int xdemhip_cov_double_sum(xdemhip_ctx* ctx, const double* ax, const double* ay, const double* ae, int64_t na, const double* bx,
                                      const double* by, const double* be, int64_t nb, int n_models, const int* model_type,
                                      const double* range, const double* psill, const double* smooth, double* out_sum, int memspace) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (!ax || !ay || !ae || na <= 0 || !out_sum || !model_type || !range || !psill) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (n_models < 1 || n_models > CV_MAXM) return xd_fail(ctx, XDEMHIP_EINVAL, "1 to 8 variogram models");
    if (memspace != XDEMHIP_HOST && memspace != XDEMHIP_DEVICE) return xd_fail(ctx, XDEMHIP_EINVAL, "bad memspace");
    const bool self = (bx == nullptr);
    if (!self && (!by || !be || nb <= 0)) return xd_fail(ctx, XDEMHIP_EINVAL, "bad B arrays");
    CovModels M;
    memset(&M, 0, sizeof M);
    M.n = n_models;
    double sill = 0.0;
    for (int m = 0; m < n_models; ++m) {
        if (model_type[m] < CV_SPHERICAL || model_type[m] > CV_STABLE)
            return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "variogram model not available on the device (spherical, exponential, gaussian, cubic, stable are)");
        if (!(range[m] > 0.0) || !(psill[m] > 0.0)) return xd_fail(ctx, XDEMHIP_EINVAL, "ranges and partial sills must be positive");
        M.type[m] = model_type[m]; M.r[m] = range[m]; M.c0[m] = psill[m];
        M.s[m] = smooth ? smooth[m] : 1.0;
        if (model_type[m] == CV_STABLE && !(M.s[m] > 0.0)) return xd_fail(ctx, XDEMHIP_EINVAL, "smoothness must be positive");
        sill += psill[m];
    }
    M.inv_sill = 1.0 / sill;
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (self) { bx = ax; by = ay; be = ae; nb = na; }
    std::vector<void*> owned;
    auto cleanup = [&]() { for (void* p : owned) (void)hipFree(p); };
    auto up = [&](const double* src, int64_t n, const double** dst) -> int {
        if (memspace == XDEMHIP_DEVICE) { *dst = src; return XDEMHIP_OK; }
        void* d = nullptr;
        if (hipMalloc(&d, (size_t)n * 8) != hipSuccess) return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
        owned.push_back(d);
        if (hipMemcpyAsync(d, src, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return xd_fail(ctx, XDEMHIP_EHIP, "H2D copy failed");
        *dst = static_cast<const double*>(d);
        return XDEMHIP_OK;
    };
    const double *dax, *day, *dae, *dbx, *dby, *dbe;
    int rc = up(ax, na, &dax);
    if (rc == XDEMHIP_OK) rc = up(ay, na, &day);
    if (rc == XDEMHIP_OK) rc = up(ae, na, &dae);
    if (self) { dbx = dax; dby = day; dbe = dae; }
    else {
        if (rc == XDEMHIP_OK) rc = up(bx, nb, &dbx);
        if (rc == XDEMHIP_OK) rc = up(by, nb, &dby);
        if (rc == XDEMHIP_OK) rc = up(be, nb, &dbe);
    }
    double* d_out = nullptr;
    if (rc == XDEMHIP_OK && hipMalloc(reinterpret_cast<void**>(&d_out), 8) != hipSuccess) rc = xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    if (rc != XDEMHIP_OK) { cleanup(); return rc; }
    owned.push_back(d_out);
    (void)hipMemsetAsync(d_out, 0, 8, ctx->stream);
    const int64_t n_wg = ((na + CV_NT - 1) / CV_NT) * ((nb + CV_CHUNK - 1) / CV_CHUNK);
    if (n_wg * CV_NT >= ((int64_t)1 << 32)) {  // (total work-items of a HIP dispatch are a 32-bit quantity)
        cleanup();
        return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "too many point tiles for one launch");
    }
    (void)hipEventRecord(ctx->ev_start, ctx->stream);
    hipLaunchKernelGGL(cov_sum_kernel, dim3((unsigned)n_wg), dim3(CV_NT), 0, ctx->stream, dax, day, dae, na, dbx, dby, dbe, nb, M, d_out);
    hipError_t e = hipGetLastError();
    (void)hipEventRecord(ctx->ev_stop, ctx->stream);
    ctx->timed = e == hipSuccess;
    if (e == hipSuccess) e = hipMemcpyAsync(out_sum, d_out, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return xd_fail(ctx, XDEMHIP_EHIP, std::string("covariance sum failed: ") + hipGetErrorString(e));
    return XDEMHIP_OK;
}
