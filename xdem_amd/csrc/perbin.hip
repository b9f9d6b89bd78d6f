// perbin.hip -- per-pixel lookup of a binned statistic (the consumer side of nd_binning):
//     xdem.spatialstats.get_perbin_nd_binning(df, list_var, list_var_names, statistic, min_count)   xdem/spatialstats.py:425-527
// Upstream forms one boolean mask per interval of every variable (var >= left & var < right, spatialstats.py:499-502), walks
// the Cartesian product of the intervals in itertools.product order, and writes the bin's statistic into the pixels of the
// combined mask when the bin's count exceeds min_count (505-525) -- L x n_bins passes over the arrays.  Here every pixel looks
// its own bin up: the host hands over the sorted unique intervals of each variable (edges already in the dtype NumPy compares
// in), the table of statistics over the product of the intervals and a byte per bin: 1 = write, 0 = the count fails, 2 = the
// DataFrame holds no such row (upstream raises IndexError at `.values[0]` as soon as a pixel lies in such a bin: counted and
// reported).  Disjoint intervals (what nd_binning produces): one containing interval per variable, one table read.  Intervals
// that overlap (a hand-made DataFrame): the product is walked in upstream's order per pixel and the last bin that writes wins,
// as the overwriting masks do.  A NaN variable lies in no interval: NaN out.  HBM traffic: the L variables read once, 8 B out.
#include "common.h"

#include <math.h>

namespace xd {

constexpr int PB_MAXVAR = 8;

struct PbArgs {
    const void* var[PB_MAXVAR];
    int dt[PB_MAXVAR], n_int[PB_MAXVAR], off[PB_MAXVAR];
    int n_var;
    int64_t n, n_bins;
    const double *left, *right, *table;
    const unsigned char* pass;
    double* out;
    unsigned long long* missing;
};

__device__ __forceinline__ double pb_load(const void* p, int dt, int64_t i) {
    return dt == XDEMHIP_F32 ? (double)static_cast<const float*>(p)[i] : static_cast<const double*>(p)[i];
}

template <bool DISJOINT>
__global__ __launch_bounds__(256) void perbin_kernel(PbArgs a) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long miss = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        double v[PB_MAXVAR];
        for (int k = 0; k < a.n_var; ++k) v[k] = pb_load(a.var[k], a.dt[k], i);
        double res = (double)NAN;
        if (DISJOINT) {
            int64_t idx = 0;
            bool in = true;
            for (int k = 0; k < a.n_var; ++k) {
                int j = -1;
                for (int jj = 0; jj < a.n_int[k]; ++jj)
                    if (v[k] >= a.left[a.off[k] + jj] && v[k] < a.right[a.off[k] + jj]) j = jj;
                in = in && j >= 0;
                idx = idx * a.n_int[k] + (j < 0 ? 0 : j);
            }
            if (in) {
                const unsigned char p = a.pass[idx];
                if (p == 1) res = a.table[idx];
                miss += p == 2;
            }
        } else {
            for (int64_t b = 0; b < a.n_bins; ++b) {   // itertools.product order: the last variable runs fastest
                int64_t r = b;
                bool in = true;
                for (int k = a.n_var - 1; k >= 0; --k) {
                    const int j = (int)(r % a.n_int[k]);
                    r /= a.n_int[k];
                    in = in && v[k] >= a.left[a.off[k] + j] && v[k] < a.right[a.off[k] + j];
                }
                if (in) {
                    const unsigned char p = a.pass[b];
                    if (p == 1) res = a.table[b];
                    miss += p == 2;
                }
            }
        }
        a.out[i] = res;
    }
    if (miss) atomicAdd(a.missing, miss);
}

}  // namespace xd

extern "C" int xdemhip_perbin_lookup(xdemhip_ctx* ctx, const void* const* vars, const int* var_dtypes, int n_var, int64_t n,
                                     const int* n_intervals, const double* left, const double* right, const double* table,
                                     const unsigned char* pass, int disjoint, double* out, int64_t* n_missing, int memspace) {
    using namespace xd;
    if (!ctx) return XDEMHIP_EINVAL;
    if (!vars || !var_dtypes || !n_intervals || !left || !right || !table || !pass || !out || n < 0 || n_var < 1)
        return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (n_var > PB_MAXVAR) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "at most 8 explanatory variables");
    PbArgs a;
    memset(&a, 0, sizeof a);
    int64_t n_bins = 1, n_edges = 0;
    for (int k = 0; k < n_var; ++k) {
        if (var_dtypes[k] != XDEMHIP_F32 && var_dtypes[k] != XDEMHIP_F64) return xd_fail(ctx, XDEMHIP_EINVAL, "variables must be float32 or float64");
        if (n_intervals[k] < 1 || !vars[k]) return xd_fail(ctx, XDEMHIP_EINVAL, "every variable needs an array and at least one interval");
        a.dt[k] = var_dtypes[k];
        a.n_int[k] = n_intervals[k];
        a.off[k] = (int)n_edges;
        n_edges += n_intervals[k];
        n_bins *= n_intervals[k];
        if (n_bins > ((int64_t)1 << 26)) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "more than 2^26 bins");
    }
    if (n_missing) *n_missing = 0;
    if (n == 0) return XDEMHIP_OK;
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // small tables: edges, statistics, pass bytes, the counter
    double *d_left = nullptr, *d_right = nullptr, *d_table = nullptr, *d_out = nullptr;
    unsigned char* d_pass = nullptr;
    unsigned long long* d_miss = nullptr;
    void* d_var[PB_MAXVAR] = {};
    auto release = [&]() {
        if (d_left) (void)hipFree(d_left);
        if (d_right) (void)hipFree(d_right);
        if (d_table) (void)hipFree(d_table);
        if (d_pass) (void)hipFree(d_pass);
        if (d_miss) (void)hipFree(d_miss);
        if (memspace == XDEMHIP_HOST) {
            for (int k = 0; k < n_var; ++k)
                if (d_var[k]) (void)hipFree(d_var[k]);
            if (d_out) (void)hipFree(d_out);
        }
    };
    bool ok = hipMalloc(reinterpret_cast<void**>(&d_left), n_edges * 8) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&d_right), n_edges * 8) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&d_table), n_bins * 8) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&d_pass), n_bins) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&d_miss), 8) == hipSuccess;
    if (ok && memspace == XDEMHIP_HOST) {
        for (int k = 0; k < n_var && ok; ++k) ok = hipMalloc(&d_var[k], (size_t)n * (var_dtypes[k] == XDEMHIP_F32 ? 4 : 8)) == hipSuccess;
        ok = ok && hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n * 8) == hipSuccess;
    }
    if (!ok) {
        release();
        return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    }
    ok = hipMemcpy(d_left, left, n_edges * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_right, right, n_edges * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_table, table, n_bins * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_pass, pass, n_bins, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemsetAsync(d_miss, 0, 8, ctx->stream) == hipSuccess;
    for (int k = 0; k < n_var && ok; ++k) {
        if (memspace == XDEMHIP_HOST) {
            ok = hipMemcpyAsync(d_var[k], vars[k], (size_t)n * (var_dtypes[k] == XDEMHIP_F32 ? 4 : 8), hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
            a.var[k] = d_var[k];
        } else {
            a.var[k] = vars[k];
        }
    }
    if (!ok) {
        release();
        return xd_fail(ctx, XDEMHIP_EHIP, "upload failed");
    }
    a.n_var = n_var; a.n = n; a.n_bins = n_bins;
    a.left = d_left; a.right = d_right; a.table = d_table; a.pass = d_pass;
    a.out = memspace == XDEMHIP_HOST ? d_out : out;
    a.missing = d_miss;
    const int64_t want = (n + 255) / 256;
    const unsigned blocks = (unsigned)(want < (int64_t)ctx->num_cu * 16 ? want : (int64_t)ctx->num_cu * 16);
    (void)hipEventRecord(ctx->ev_start, ctx->stream);
    if (disjoint) hipLaunchKernelGGL((perbin_kernel<true>), dim3(blocks), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL((perbin_kernel<false>), dim3(blocks), dim3(256), 0, ctx->stream, a);
    (void)hipEventRecord(ctx->ev_stop, ctx->stream);
    ctx->timed = true;
    int rc = XDEMHIP_OK;
    if (hipGetLastError() != hipSuccess) rc = xd_fail(ctx, XDEMHIP_EHIP, "per-bin lookup kernel launch failed");
    unsigned long long miss = 0;
    if (rc == XDEMHIP_OK && memspace == XDEMHIP_HOST && hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = xd_fail(ctx, XDEMHIP_EHIP, "D2H failed");
    if (rc == XDEMHIP_OK && (hipMemcpyAsync(&miss, d_miss, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                             hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = xd_fail(ctx, XDEMHIP_EHIP, "per-bin lookup kernel failed");
    if (n_missing) *n_missing = (int64_t)miss;
    release();
    return rc;
}
