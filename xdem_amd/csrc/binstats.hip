// binstats.hip -- N-dimensional binned statistics on gfx950.
//
// Replaces the array work of  nd_binning(values, list_var, ...)   xdem/spatialstats.py:91-216
// i.e. what scipy.stats.binned_statistic / _2d / _dd do there for the statistics "count", np.nanmedian and
// geoutils' nmad (1.4826 * nanmedian(|x - nanmedian(x)|)), the input of the heteroscedasticity inference
// (spatialstats.py:576-631).  Per call:
//   * the joint finiteness mask of values and all variables and every variable's min / max over it (nd_binning
//     drops non-finite rows before SciPy derives its bin edges from the data range)          -> finalize
//   * SciPy's bin numbers (_binned_statistic.py:_bin_numbers): np.digitize against the edges of each dimension, samples
//     on the rightmost edge (after its decimal rounding rule) moved into the last bin, outliers dropped; flattened in
//     C order                                                                                 -> bin_ids_kernel
//   * per-bin count and exact median of the values: radix selection of select_run.h (integer histograms in LDS)
//   * per-bin NMAD: |v - median[bin]| in the value dtype, selected the same way.
// Bin edges themselves (np.linspace in SciPy's dtype rules) are a host matter and come in as float64 numbers.
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "select.h"
#include "select_run.h"

namespace xd {

constexpr int BS_MAXDIM = 8;

struct BinDims {
    int nd;
    int ne[BS_MAXDIM];        // edges per dimension
    int off[BS_MAXDIM];       // offset of the dimension's edges in the LDS edge table
    int stride[BS_MAXDIM];    // C-order stride of the dimension in the flattened bin id
    int decimal[BS_MAXDIM];   // SciPy's rounding precision of the rightmost-edge rule
    int is_f32[BS_MAXDIM];    // arithmetic type of that rule (dtype of SciPy's sample matrix)
    const void* var[BS_MAXDIM];
    int var_f32[BS_MAXDIM];   // storage type of the variable
};

// np.around(x, decimal) in the array dtype (multiarray/calculation.c PyArray_Round): scale by 10^|decimal|, rint, unscale
template <typename T> __device__ __forceinline__ T np_around(T x, int decimal, T p10) {
    if (decimal >= 0) return rint(x * p10) / p10;
    return rint(x / p10) * p10;
}

template <typename TV>
__global__ __launch_bounds__(256) void finite_and_kernel(const TV* v, int64_t n, uint8_t* valid, int first) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const bool f = t_finite<TV>(v[p]);
        valid[p] = first ? (uint8_t)f : (uint8_t)(valid[p] && f);
    }
}

template <typename TV>
__global__ __launch_bounds__(256) void minmax_kernel(const TV* v, const uint8_t* valid, int64_t n, uint64_t* out /* [min key, max key, count] */) {
    typedef typename KeyT<TV>::type K;
    K kmin = ~(K)0, kmax = 0;
    unsigned long long cnt = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        if (!valid[p]) continue;
        const K k = key_of(v[p]);
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
        ++cnt;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const K a = k_shfl_down(kmin, off), b = k_shfl_down(kmax, off);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
        cnt += __shfl_down(cnt, off);
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        k_atomic_min(&out[0], (uint64_t)kmin);
        k_atomic_max(&out[1], (uint64_t)kmax);
        atomicAdd(reinterpret_cast<unsigned long long*>(&out[2]), cnt);
    }
}

__global__ __launch_bounds__(256) void bin_ids_kernel(BinDims D, const double* __restrict__ edges, int n_edges_total,
                                                      const uint8_t* __restrict__ valid, int64_t n, uint16_t* __restrict__ bins) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* e = reinterpret_cast<double*>(smem);
    for (int k = threadIdx.x; k < n_edges_total; k += blockDim.x) e[k] = edges[k];
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        int flat = 0;
        bool ok = valid[p] != 0;
        for (int d = 0; d < D.nd && ok; ++d) {
            const double x = D.var_f32[d] ? (double)static_cast<const float*>(D.var[d])[p] : static_cast<const double*>(D.var[d])[p];
            const double* ed = e + D.off[d];
            const int ne = D.ne[d];
            // np.digitize(x, edges): number of edges <= x
            int lo = 0, hi = ne;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (ed[mid] <= x) lo = mid + 1; else hi = mid;
            }
            int idx = lo;
            const double last = ed[ne - 1];
            if (x >= last) {
                bool same;
                if (D.is_f32[d]) {
                    const float p10 = (float)pow(10.0, (double)abs(D.decimal[d]));
                    same = np_around<float>((float)x, D.decimal[d], p10) == np_around<float>((float)last, D.decimal[d], p10);
                } else {
                    const double p10 = pow(10.0, (double)abs(D.decimal[d]));
                    same = np_around<double>(x, D.decimal[d], p10) == np_around<double>(last, D.decimal[d], p10);
                }
                if (same) idx -= 1;
            }
            if (idx < 1 || idx > ne - 1) ok = false;
            else flat += (idx - 1) * D.stride[d];
        }
        bins[p] = ok ? (uint16_t)flat : (uint16_t)0xFFFF;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void absdev_kernel(const T* __restrict__ v, const uint16_t* __restrict__ bins, const T* __restrict__ med,
                                                     int64_t n, T* __restrict__ out) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const uint16_t b = bins[p];
        T r = (T)NAN;
        if (b != 0xFFFF) {
            const T d = t_sub(v[p], med[b]);
            r = d < (T)0 ? -d : d;  // np.abs
            if (r != r) r = (T)NAN;
        }
        out[p] = r;
    }
}

// ---- multilinear interpolation on a regular grid (scipy.interpolate.RegularGridInterpolator, method="linear",
// bounds_error=False, fill_value=None: linear extrapolation from the edge intervals; NaN in any coordinate -> NaN) ----
struct GridDims {
    int nd;
    int n[BS_MAXDIM], off[BS_MAXDIM], stride[BS_MAXDIM];
    const void* var[BS_MAXDIM];
    int var_f32[BS_MAXDIM];
};

__global__ __launch_bounds__(256) void interp_grid_kernel(GridDims G, const double* __restrict__ axes, int n_axes_total,
                                                          const double* __restrict__ gv, int n_grid, int grid_in_lds, int64_t n,
                                                          double scale, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ax = reinterpret_cast<double*>(smem);
    double* vals = ax + n_axes_total;
    for (int k = threadIdx.x; k < n_axes_total; k += blockDim.x) ax[k] = axes[k];
    if (grid_in_lds)
        for (int k = threadIdx.x; k < n_grid; k += blockDim.x) vals[k] = gv[k];
    __syncthreads();
    const double* V = grid_in_lds ? vals : gv;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        int base = 0;
        double y[BS_MAXDIM];
        bool isnan_any = false;
#pragma unroll 1
        for (int d = 0; d < G.nd; ++d) {
            const double x = G.var_f32[d] ? (double)static_cast<const float*>(G.var[d])[p] : static_cast<const double*>(G.var[d])[p];
            isnan_any |= (x != x);
            const double* g = ax + G.off[d];
            const int m = G.n[d];
            // interval i with g[i] <= x < g[i+1], clipped to [0, m-2] (x == g[m-1] belongs to the last interval)
            int lo = 0, hi = m - 1;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (g[mid] <= x) lo = mid; else hi = mid;
            }
            y[d] = (x - g[lo]) / (g[lo + 1] - g[lo]);
            base += lo * G.stride[d];
        }
        // hypercube corners in itertools.product order (first dimension slowest), weights multiplied left to right
        double value = 0.0;
        const int corners = 1 << G.nd;
        for (int c = 0; c < corners; ++c) {
            double wgt = 1.0;
            int idx = base;
            for (int d = 0; d < G.nd; ++d) {
                const int up = (c >> (G.nd - 1 - d)) & 1;
                wgt = wgt * (up ? y[d] : (1.0 - y[d]));
                idx += up * G.stride[d];
            }
            value = value + V[idx] * wgt;
        }
        out[p] = isnan_any ? (double)NAN : scale * value;
    }
}

// |v| > limit -> NaN (two_step_standardization's outlier filter), else v; limit = +inf keeps everything
template <typename T>
__global__ __launch_bounds__(256) void clip_abs_kernel(const T* __restrict__ v, int64_t n, T limit, T* __restrict__ out) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const T x = v[p];
        const T a = x < (T)0 ? -x : x;
        out[p] = (a > limit) ? (T)NAN : x;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void absdev1_kernel(const T* __restrict__ v, int64_t n, T med, T* __restrict__ out) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const T d = t_sub(v[p], med);
        out[p] = d < (T)0 ? -d : d;
    }
}

}  // namespace xd

using namespace xd;

struct xdemhip_binstats {
    xdemhip_ctx* ctx = nullptr;
    int dtype = XDEMHIP_F32;
    int64_t n = 0;
    void* values = nullptr;
    bool own_values = false;
    uint8_t* valid = nullptr;
    std::vector<void*> var;
    std::vector<int> var_dtype;
    std::vector<bool> var_own;
    uint16_t* bins = nullptr;
    void* absdev = nullptr;
    bool finalized = false;
};

namespace {

int upload(xdemhip_ctx* ctx, const void* src, size_t bytes, int memspace, void** dst, bool* own) {
    if (memspace == XDEMHIP_DEVICE) { *dst = const_cast<void*>(src); *own = false; return XDEMHIP_OK; }
    if (hipMalloc(dst, bytes) != hipSuccess) return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    *own = true;
    XD_HIP_CHECK(ctx, hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return XDEMHIP_OK;
}

template <typename T>
int run_typed(xdemhip_binstats* P, int nb, int want_nmad, double nfact, int64_t* counts, double* medians, double* nmads) {
    typedef typename KeyT<T>::type K;
    xdemhip_ctx* ctx = P->ctx;
    void* scratch = nullptr;
    if (hipMalloc(&scratch, scratch_size(nb)) != hipSuccess) return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc(scratch) failed");
    unsigned char* base = static_cast<unsigned char*>(scratch);
    std::vector<SelResult<K>> hs;
    SelWorkspace ws;
    if (P->n >= SEL_BRACKET_MIN_N && nb <= MAX_BINS_PER_SWEEP && sel_ws_create(ctx, P->n, sizeof(T), nb, ws) != XDEMHIP_OK) {
        (void)hipFree(scratch);
        return XDEMHIP_ENOMEM;
    }
    int rc = run_select<T>(ctx, static_cast<const T*>(P->values), P->bins, P->n, nb, base, hs, &ws);
    std::vector<T> med(nb);
    if (rc == XDEMHIP_OK)
        for (int k = 0; k < nb; ++k) {
            counts[k] = (int64_t)hs[k].st.count;
            medians[k] = median_from<T>(hs[k]);
            med[k] = (T)medians[k];
        }
    if (rc == XDEMHIP_OK && want_nmad) {
        T* d_med = reinterpret_cast<T*>(base);  // (the edge area of the scratch block is free here)
        void* d_med_big = nullptr;
        if (sizeof(T) * (size_t)nb > OFF_STATS) {
            if (hipMalloc(&d_med_big, sizeof(T) * (size_t)nb) != hipSuccess) { (void)hipFree(scratch); return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed"); }
            d_med = static_cast<T*>(d_med_big);
        }
        hipError_t e = hipMemcpyAsync(d_med, med.data(), sizeof(T) * (size_t)nb, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((absdev_kernel<T>), dim3(grid_for(ctx, P->n, 256, 16)), dim3(256), 0, ctx->stream,
                               static_cast<const T*>(P->values), P->bins, d_med, P->n, static_cast<T*>(P->absdev));
            e = hipGetLastError();
        }
        if (e != hipSuccess) rc = xd_fail(ctx, XDEMHIP_EHIP, std::string("absdev launch failed: ") + hipGetErrorString(e));
        if (rc == XDEMHIP_OK) rc = run_select<T>(ctx, static_cast<const T*>(P->absdev), P->bins, P->n, nb, base, hs, &ws);
        if (rc == XDEMHIP_OK)
            for (int k = 0; k < nb; ++k) {
                const double m = median_from<T>(hs[k]);
                nmads[k] = (double)(T)((T)nfact * (T)m);  // nfact * np.nanmedian(...): the Python float is a weak scalar
            }
        if (d_med_big) (void)hipFree(d_med_big);
    }
    sel_ws_free(ws);
    (void)hipFree(scratch);
    return rc;
}

}  // namespace

extern "C" {

void xdemhip_binstats_destroy(xdemhip_binstats* P) {
    if (!P) return;
    (void)hipSetDevice(P->ctx->device);
    (void)hipStreamSynchronize(P->ctx->stream);
    if (P->own_values && P->values) (void)hipFree(P->values);
    for (size_t i = 0; i < P->var.size(); ++i)
        if (P->var_own[i] && P->var[i]) (void)hipFree(P->var[i]);
    if (P->valid) (void)hipFree(P->valid);
    if (P->bins) (void)hipFree(P->bins);
    if (P->absdev) (void)hipFree(P->absdev);
    delete P;
}

int xdemhip_binstats_create(xdemhip_ctx* ctx, const void* values, int dtype, int64_t n, int memspace, xdemhip_binstats** out) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (!values || !out || n <= 0) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (dtype != XDEMHIP_F32 && dtype != XDEMHIP_F64) return xd_fail(ctx, XDEMHIP_EINVAL, "dtype must be float32 or float64");
    if (memspace != XDEMHIP_HOST && memspace != XDEMHIP_DEVICE) return xd_fail(ctx, XDEMHIP_EINVAL, "bad memspace");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    xdemhip_binstats* P = new xdemhip_binstats;
    P->ctx = ctx; P->dtype = dtype; P->n = n;
    const size_t es = dtype == XDEMHIP_F32 ? 4 : 8;
    int rc = upload(ctx, values, (size_t)n * es, memspace, &P->values, &P->own_values);
    if (rc == XDEMHIP_OK && (hipMalloc(reinterpret_cast<void**>(&P->valid), (size_t)n) != hipSuccess ||
                             hipMalloc(reinterpret_cast<void**>(&P->bins), (size_t)n * 2) != hipSuccess ||
                             hipMalloc(&P->absdev, (size_t)n * es) != hipSuccess))
        rc = xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    if (rc != XDEMHIP_OK) { xdemhip_binstats_destroy(P); return rc; }
    const dim3 g(grid_for(ctx, n, 256, 16));
    if (dtype == XDEMHIP_F32) hipLaunchKernelGGL((finite_and_kernel<float>), g, dim3(256), 0, ctx->stream, static_cast<const float*>(P->values), n, P->valid, 1);
    else hipLaunchKernelGGL((finite_and_kernel<double>), g, dim3(256), 0, ctx->stream, static_cast<const double*>(P->values), n, P->valid, 1);
    *out = P;
    return XDEMHIP_OK;
}

int xdemhip_binstats_add_var(xdemhip_binstats* P, const void* var, int dtype, int memspace) {
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (!var || (dtype != XDEMHIP_F32 && dtype != XDEMHIP_F64) || (memspace != XDEMHIP_HOST && memspace != XDEMHIP_DEVICE))
        return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if ((int)P->var.size() >= BS_MAXDIM) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "at most 8 explanatory variables");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    void* d = nullptr;
    bool own = false;
    const size_t es = dtype == XDEMHIP_F32 ? 4 : 8;
    int rc = upload(ctx, var, (size_t)P->n * es, memspace, &d, &own);
    if (rc != XDEMHIP_OK) return rc;
    P->var.push_back(d); P->var_dtype.push_back(dtype); P->var_own.push_back(own);
    const dim3 g(grid_for(ctx, P->n, 256, 16));
    if (dtype == XDEMHIP_F32) hipLaunchKernelGGL((finite_and_kernel<float>), g, dim3(256), 0, ctx->stream, static_cast<const float*>(d), P->n, P->valid, 0);
    else hipLaunchKernelGGL((finite_and_kernel<double>), g, dim3(256), 0, ctx->stream, static_cast<const double*>(d), P->n, P->valid, 0);
    XD_HIP_CHECK(ctx, hipGetLastError());
    P->finalized = false;
    return (int)P->var.size() - 1;
}

int xdemhip_binstats_finalize(xdemhip_binstats* P, int64_t* n_valid, double* var_min, double* var_max) {
    XdFetchScope fetch_scope_(P ? P->ctx : nullptr);
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (!n_valid || !var_min || !var_max) return xd_fail(ctx, XDEMHIP_EINVAL, "null output");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t nv = P->var.size();
    uint64_t* d_out = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d_out), 24 * (nv + 1)) != hipSuccess) return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    std::vector<uint64_t> h(3 * (nv + 1));
    for (size_t i = 0; i <= nv; ++i) { h[3 * i] = ~(uint64_t)0; h[3 * i + 1] = 0; h[3 * i + 2] = 0; }
    hipError_t e = hipMemcpyAsync(d_out, h.data(), 24 * (nv + 1), hipMemcpyHostToDevice, ctx->stream);
    const dim3 g(grid_for(ctx, P->n, 256, 16));
    for (size_t i = 0; i < nv && e == hipSuccess; ++i) {
        if (P->var_dtype[i] == XDEMHIP_F32) hipLaunchKernelGGL((minmax_kernel<float>), g, dim3(256), 0, ctx->stream, static_cast<const float*>(P->var[i]), P->valid, P->n, d_out + 3 * i);
        else hipLaunchKernelGGL((minmax_kernel<double>), g, dim3(256), 0, ctx->stream, static_cast<const double*>(P->var[i]), P->valid, P->n, d_out + 3 * i);
        e = hipGetLastError();
    }
    if (e == hipSuccess && nv == 0) {  // count only
        if (P->dtype == XDEMHIP_F32) hipLaunchKernelGGL((minmax_kernel<float>), g, dim3(256), 0, ctx->stream, static_cast<const float*>(P->values), P->valid, P->n, d_out);
        else hipLaunchKernelGGL((minmax_kernel<double>), g, dim3(256), 0, ctx->stream, static_cast<const double*>(P->values), P->valid, P->n, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_out, 24 * (nv + 1), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return xd_fail(ctx, XDEMHIP_EHIP, std::string("binstats finalize failed: ") + hipGetErrorString(e));
    *n_valid = (int64_t)h[2];
    for (size_t i = 0; i < nv; ++i) {
        if (h[3 * i + 2] == 0) { var_min[i] = NAN; var_max[i] = NAN; continue; }
        if (P->var_dtype[i] == XDEMHIP_F32) { var_min[i] = (double)val_of((uint32_t)h[3 * i]); var_max[i] = (double)val_of((uint32_t)h[3 * i + 1]); }
        else { var_min[i] = val_of((uint64_t)h[3 * i]); var_max[i] = val_of((uint64_t)h[3 * i + 1]); }
    }
    P->finalized = true;
    return XDEMHIP_OK;
}

int xdemhip_binstats_run(xdemhip_binstats* P, int n_dims, const int* var_ids, const double* edges, const int* n_edges,
                         const int* decimals, int sample_dtype, int want_nmad, double nfact, int64_t* counts, double* medians,
                         double* nmads) {
    XdFetchScope fetch_scope_(P ? P->ctx : nullptr);
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (!P->finalized) return xd_fail(ctx, XDEMHIP_EINVAL, "call xdemhip_binstats_finalize first");
    if (n_dims < 1 || n_dims > BS_MAXDIM || !var_ids || !edges || !n_edges || !decimals || !counts || !medians || (want_nmad && !nmads))
        return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (sample_dtype != XDEMHIP_F32 && sample_dtype != XDEMHIP_F64) return xd_fail(ctx, XDEMHIP_EINVAL, "bad sample dtype");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    BinDims D;
    memset(&D, 0, sizeof D);
    D.nd = n_dims;
    int64_t nb = 1;
    int tot = 0;
    for (int d = 0; d < n_dims; ++d) {
        if (var_ids[d] < 0 || var_ids[d] >= (int)P->var.size()) return xd_fail(ctx, XDEMHIP_EINVAL, "unknown variable id");
        if (n_edges[d] < 2) return xd_fail(ctx, XDEMHIP_EINVAL, "a dimension needs at least 2 edges");
        D.ne[d] = n_edges[d]; D.off[d] = tot; tot += n_edges[d];
        D.decimal[d] = decimals[d]; D.is_f32[d] = sample_dtype == XDEMHIP_F32;
        D.var[d] = P->var[var_ids[d]]; D.var_f32[d] = P->var_dtype[var_ids[d]] == XDEMHIP_F32;
        nb *= (n_edges[d] - 1);
        if (nb > 3072) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "more than 3072 bins in one binning");
    }
    for (int d = n_dims - 1, s = 1; d >= 0; --d) { D.stride[d] = s; s *= (n_edges[d] - 1); }
    if ((size_t)tot * 8 > 60 * 1024) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "too many bin edges for the LDS table");
    double* d_edges = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d_edges), (size_t)tot * 8) != hipSuccess) return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    hipError_t e = hipMemcpyAsync(d_edges, edges, (size_t)tot * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        if ((size_t)tot * 8 > 48 * 1024)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(bin_ids_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tot * 8);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bin_ids_kernel, dim3(grid_for(ctx, P->n, 256, 16)), dim3(256), (size_t)tot * 8, ctx->stream, D, d_edges, tot,
                           P->valid, P->n, P->bins);
        e = hipGetLastError();
    }
    int rc = XDEMHIP_OK;
    if (e != hipSuccess) rc = xd_fail(ctx, XDEMHIP_EHIP, std::string("bin id launch failed: ") + hipGetErrorString(e));
    if (rc == XDEMHIP_OK) {
        const xdemhip_allreduce_fn hook = ctx->allreduce;
        ctx->allreduce = nullptr;  // single-device helper
        rc = P->dtype == XDEMHIP_F32 ? run_typed<float>(P, (int)nb, want_nmad, nfact, counts, medians, nmads)
                                     : run_typed<double>(P, (int)nb, want_nmad, nfact, counts, medians, nmads);
        ctx->allreduce = hook;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_edges);
    return rc;
}

int xdemhip_binstats_bin_numbers(xdemhip_binstats* P, uint16_t* bins_out) {
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (!P->finalized || !P->bins || !bins_out) return xd_fail(ctx, XDEMHIP_EINVAL, "xdemhip_binstats_bin_numbers: after xdemhip_binstats_run, with an output array");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XD_HIP_CHECK(ctx, hipMemcpyAsync(bins_out, P->bins, (size_t)P->n * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return XDEMHIP_OK;
}

}  // extern "C"

template <typename T>
static int nmad_typed(xdemhip_ctx* ctx, const void* values, int64_t n, double nfact, double abs_limit, int memspace, double* median,
                      double* nmad_out, int64_t* count) {
    typedef typename KeyT<T>::type K;
    void *d_v = nullptr, *d_w = nullptr, *scratch = nullptr;
    bool own = false;
    auto cleanup = [&]() { if (own && d_v) (void)hipFree(d_v); if (d_w) (void)hipFree(d_w); if (scratch) (void)hipFree(scratch); };
    int rc = upload(ctx, values, (size_t)n * sizeof(T), memspace, &d_v, &own);
    if (rc == XDEMHIP_OK && (hipMalloc(&d_w, (size_t)n * sizeof(T)) != hipSuccess || hipMalloc(&scratch, scratch_size(1)) != hipSuccess))
        rc = xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    if (rc != XDEMHIP_OK) { cleanup(); return rc; }
    const dim3 g(grid_for(ctx, n, 256, 16));
    const T* src = static_cast<const T*>(d_v);
    if (abs_limit == abs_limit && abs_limit < INFINITY) {
        hipLaunchKernelGGL((clip_abs_kernel<T>), g, dim3(256), 0, ctx->stream, src, n, (T)abs_limit, static_cast<T*>(d_w));
        // the filtered copy becomes the data; the deviations need a second buffer
        void* d_f = d_w;
        d_w = nullptr;
        if (hipMalloc(&d_w, (size_t)n * sizeof(T)) != hipSuccess) { d_w = d_f; cleanup(); return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed"); }
        if (own) (void)hipFree(d_v);
        d_v = d_f; own = true;
        src = static_cast<const T*>(d_v);
    }
    unsigned char* base = static_cast<unsigned char*>(scratch);
    std::vector<SelResult<K>> r;
    const xdemhip_allreduce_fn hook = ctx->allreduce;
    ctx->allreduce = nullptr;
    SelWorkspace ws;
    if (n >= SEL_BRACKET_MIN_N) (void)sel_ws_create(ctx, n, sizeof(T), 1, ws);  // (on failure: plain selection)
    rc = run_select<T>(ctx, src, nullptr, n, 1, base, r, &ws);
    if (rc == XDEMHIP_OK) {
        *count = (int64_t)r[0].st.count;
        const double med = median_from<T>(r[0]);
        *median = med;
        hipLaunchKernelGGL((absdev1_kernel<T>), g, dim3(256), 0, ctx->stream, src, n, (T)med, static_cast<T*>(d_w));
        rc = run_select<T>(ctx, static_cast<const T*>(d_w), nullptr, n, 1, base, r, &ws);
        if (rc == XDEMHIP_OK) *nmad_out = (double)(T)((T)nfact * (T)median_from<T>(r[0]));
    }
    ctx->allreduce = hook;
    (void)hipStreamSynchronize(ctx->stream);
    sel_ws_free(ws);
    cleanup();
    return rc;
}

extern "C" {

int xdemhip_nmad(xdemhip_ctx* ctx, const void* values, int dtype, int64_t n, double nfact, double abs_limit, int memspace,
                 double* median, double* nmad_out, int64_t* count) {
    XdFetchScope fetch_scope_(ctx);
    if (!ctx) return XDEMHIP_EINVAL;
    if (!values || n <= 0 || !median || !nmad_out || !count) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (memspace != XDEMHIP_HOST && memspace != XDEMHIP_DEVICE) return xd_fail(ctx, XDEMHIP_EINVAL, "bad memspace");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (dtype == XDEMHIP_F32) return nmad_typed<float>(ctx, values, n, nfact, abs_limit, memspace, median, nmad_out, count);
    if (dtype == XDEMHIP_F64) return nmad_typed<double>(ctx, values, n, nfact, abs_limit, memspace, median, nmad_out, count);
    return xd_fail(ctx, XDEMHIP_EINVAL, "dtype must be float32 or float64");
}

int xdemhip_interp_grid_linear(xdemhip_ctx* ctx, int n_dims, const double* axes, const int* n_axis, const double* grid_values,
                               const void* const* vars, const int* var_dtypes, int64_t n, double scale, double* out, int memspace) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (n_dims < 1 || n_dims > BS_MAXDIM || !axes || !n_axis || !grid_values || !vars || !var_dtypes || !out || n <= 0)
        return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (memspace != XDEMHIP_HOST && memspace != XDEMHIP_DEVICE) return xd_fail(ctx, XDEMHIP_EINVAL, "bad memspace");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    GridDims G;
    memset(&G, 0, sizeof G);
    G.nd = n_dims;
    int tot = 0;
    int64_t ngrid = 1;
    for (int d = 0; d < n_dims; ++d) {
        if (n_axis[d] < 2) return xd_fail(ctx, XDEMHIP_EINVAL, "every grid axis needs at least 2 points");
        if (var_dtypes[d] != XDEMHIP_F32 && var_dtypes[d] != XDEMHIP_F64) return xd_fail(ctx, XDEMHIP_EINVAL, "bad variable dtype");
        G.n[d] = n_axis[d]; G.off[d] = tot; tot += n_axis[d];
        ngrid *= n_axis[d];
        if (ngrid > (1 << 24)) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "interpolation grid too large");
    }
    for (int d = n_dims - 1, st = 1; d >= 0; --d) { G.stride[d] = st; st *= n_axis[d]; }
    std::vector<void*> owned;
    double *d_axes = nullptr, *d_grid = nullptr, *d_out = nullptr;
    auto cleanup = [&]() {
        for (void* p : owned) (void)hipFree(p);
        if (d_axes) (void)hipFree(d_axes);
        if (d_grid) (void)hipFree(d_grid);
        if (memspace == XDEMHIP_HOST && d_out) (void)hipFree(d_out);
    };
    int rc = XDEMHIP_OK;
    for (int d = 0; d < n_dims && rc == XDEMHIP_OK; ++d) {
        void* p = nullptr;
        bool own = false;
        rc = upload(ctx, vars[d], (size_t)n * (var_dtypes[d] == XDEMHIP_F32 ? 4 : 8), memspace, &p, &own);
        if (own) owned.push_back(p);
        G.var[d] = p; G.var_f32[d] = var_dtypes[d] == XDEMHIP_F32;
    }
    if (rc == XDEMHIP_OK && (hipMalloc(reinterpret_cast<void**>(&d_axes), (size_t)tot * 8) != hipSuccess ||
                             hipMalloc(reinterpret_cast<void**>(&d_grid), (size_t)ngrid * 8) != hipSuccess))
        rc = xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    if (rc == XDEMHIP_OK) {
        if (memspace == XDEMHIP_DEVICE) d_out = out;
        else if (hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n * 8) != hipSuccess) rc = xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    }
    if (rc != XDEMHIP_OK) { cleanup(); return rc; }
    hipError_t e = hipMemcpyAsync(d_axes, axes, (size_t)tot * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_grid, grid_values, (size_t)ngrid * 8, hipMemcpyHostToDevice, ctx->stream);
    const int in_lds = ((size_t)(tot + ngrid) * 8 <= 40 * 1024);
    const size_t lds = (size_t)(tot + (in_lds ? ngrid : 0)) * 8;
    if (e == hipSuccess && lds > 48 * 1024) rc = xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "too many grid axis points");
    if (e == hipSuccess && rc == XDEMHIP_OK) {
        hipLaunchKernelGGL(interp_grid_kernel, dim3(grid_for(ctx, n, 256, 16)), dim3(256), lds, ctx->stream, G, d_axes, tot, d_grid,
                           (int)ngrid, in_lds, n, scale, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess && rc == XDEMHIP_OK && memspace == XDEMHIP_HOST)
        e = hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess && rc == XDEMHIP_OK) rc = xd_fail(ctx, XDEMHIP_EHIP, std::string("grid interpolation failed: ") + hipGetErrorString(e));
    cleanup();
    return rc;
}

}  // extern "C"
