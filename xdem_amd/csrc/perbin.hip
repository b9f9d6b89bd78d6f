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
    int n_var, vec;
    int64_t n, n_bins;
    const double *left, *right, *table;
    const unsigned char* pass;
    double* out;
    unsigned long long* missing;
};

__device__ __forceinline__ double pb_load(const void* p, int dt, int64_t i) {
    return dt == XDEMHIP_F32 ? (double)static_cast<const float*>(p)[i] : static_cast<const double*>(p)[i];
}

constexpr int PB_LDS_EDGES = 2048, PB_LDS_BINS = 2048;   // tables staged in LDS up to these sizes (beyond: read from global memory)
constexpr int PB_U = 4;                                    // pixels per thread and trip: four independent search chains in flight

// Every loop over the variables is unrolled over PB_MAXVAR with a guard, so that the per-variable fields of the argument block
// are indexed statically (scalar registers) instead of living in scratch memory.  LDS: interval ends, statistics and decision
// bytes of the bins are staged in (dynamic) shared memory -- [left | right | table | pass].
// NV: compile-time bound of the variable loops (1, 2, 3, 4 or 8 >= n_var): two variables keep 8 instead of 32 values per thread in
// registers, i.e. more waves per SIMD to hide the dependent LDS reads of the searches behind
template <bool DISJOINT, bool LDS, int NV>
__global__ __launch_bounds__(256) void perbin_kernel(PbArgs a) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    int n_edges = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (k < a.n_var) n_edges += a.n_int[k];
    double* s_left = reinterpret_cast<double*>(s_raw);
    double* s_right = s_left + n_edges;
    double* s_table = s_right + n_edges;
    unsigned char* s_pass = reinterpret_cast<unsigned char*>(s_table + a.n_bins);
    if (LDS) {
        for (int e = threadIdx.x; e < n_edges; e += 256) {
            s_left[e] = a.left[e];
            s_right[e] = a.right[e];
        }
        for (int e = threadIdx.x; e < (int)a.n_bins; e += 256) {
            s_table[e] = a.table[e];
            s_pass[e] = a.pass[e];
        }
        __syncthreads();
    }
    const double* __restrict__ g_left = a.left;
    const double* __restrict__ g_right = a.right;
    auto lo_of = [&](int e) { return LDS ? s_left[e] : g_left[e]; };
    auto hi_of = [&](int e) { return LDS ? s_right[e] : g_right[e]; };
    auto pass_of = [&](int64_t b) { return LDS ? s_pass[b] : a.pass[b]; };
    auto table_of = [&](int64_t b) { return LDS ? s_table[b] : a.table[b]; };
    // a thread owns PB_U CONSECUTIVE pixels per trip: one 16-byte load per float32 variable (two for float64), two 16-byte
    // stores -- when every array is 16-byte aligned (a.vec; the library's and NumPy's / torch's allocations are), else element-wise
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * PB_U;
    unsigned long long miss = 0;
    for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * PB_U; i0 < a.n; i0 += stride) {
        double v[PB_U][NV];
        const bool whole = a.vec && i0 + PB_U <= a.n;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (k < a.n_var && whole) {
                if (a.dt[k] == XDEMHIP_F32) {
                    const float4 t = *reinterpret_cast<const float4*>(static_cast<const float*>(a.var[k]) + i0);
                    v[0][k] = (double)t.x; v[1][k] = (double)t.y; v[2][k] = (double)t.z; v[3][k] = (double)t.w;
                } else {
                    const double2 t0 = *reinterpret_cast<const double2*>(static_cast<const double*>(a.var[k]) + i0);
                    const double2 t1 = *reinterpret_cast<const double2*>(static_cast<const double*>(a.var[k]) + i0 + 2);
                    v[0][k] = t0.x; v[1][k] = t0.y; v[2][k] = t1.x; v[3][k] = t1.y;
                }
            } else {
#pragma unroll
                for (int u = 0; u < PB_U; ++u) v[u][k] = (k < a.n_var && i0 + u < a.n) ? pb_load(a.var[k], a.dt[k], i0 + u) : (double)NAN;
            }
        }
        double res[PB_U];
        if (DISJOINT) {
            // sorted disjoint intervals: the only candidate is the last one whose left end is <= v.  Branch-free binary search
            // with a wave-uniform trip count (a NaN compares false everywhere and finds none).
            int64_t idx[PB_U];
            bool in[PB_U];
#pragma unroll
            for (int u = 0; u < PB_U; ++u) { idx[u] = 0; in[u] = true; }
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (k < a.n_var) {
                    const int base = a.off[k], n = a.n_int[k];
                    int top = 1;
                    while (top <= n) top <<= 1;   // (uniform)
                    int pos[PB_U];
#pragma unroll
                    for (int u = 0; u < PB_U; ++u) pos[u] = 0;
                    for (int len = top >> 1; len > 0; len >>= 1) {
#pragma unroll
                        for (int u = 0; u < PB_U; ++u) {
                            const int cand = pos[u] + len;
                            const int at = cand <= n ? cand : n;   // (stay inside the table; the result is discarded when cand > n)
                            pos[u] = (cand <= n && lo_of(base + at - 1) <= v[u][k]) ? cand : pos[u];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < PB_U; ++u) {
                        const int j = pos[u] - 1;
                        in[u] = in[u] && j >= 0 && v[u][k] < hi_of(base + (j < 0 ? 0 : j));
                        idx[u] = idx[u] * n + (j < 0 ? 0 : j);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < PB_U; ++u) {
                res[u] = (double)NAN;
                if (in[u]) {
                    const unsigned char p = pass_of(idx[u]);
                    if (p == 1) res[u] = table_of(idx[u]);
                    miss += p == 2;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < PB_U; ++u) res[u] = (double)NAN;
            for (int64_t b = 0; b < a.n_bins; ++b) {   // itertools.product order: the last variable runs fastest
                int jk[NV];
                int64_t r = b;
#pragma unroll
                for (int k = NV - 1; k >= 0; --k) {
                    jk[k] = 0;
                    if (k < a.n_var) {
                        jk[k] = (int)(r % a.n_int[k]);
                        r /= a.n_int[k];
                    }
                }
                const unsigned char p = pass_of(b);
                const double t = table_of(b);
#pragma unroll
                for (int u = 0; u < PB_U; ++u) {
                    bool in = true;
#pragma unroll
                    for (int k = 0; k < NV; ++k)
                        if (k < a.n_var) in = in && v[u][k] >= lo_of(a.off[k] + jk[k]) && v[u][k] < hi_of(a.off[k] + jk[k]);
                    if (in) {
                        if (p == 1) res[u] = t;
                        miss += p == 2;
                    }
                }
            }
        }
        if (whole) {
            *reinterpret_cast<double2*>(a.out + i0) = make_double2(res[0], res[1]);
            *reinterpret_cast<double2*>(a.out + i0 + 2) = make_double2(res[2], res[3]);
        } else {
#pragma unroll
            for (int u = 0; u < PB_U; ++u)
                if (i0 + u < a.n) a.out[i0 + u] = res[u];
        }
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
    a.vec = ((uintptr_t)(memspace == XDEMHIP_HOST ? (void*)d_out : (void*)out) & 15) == 0;
    for (int k = 0; k < n_var; ++k) a.vec = a.vec && ((uintptr_t)a.var[k] & 15) == 0;
    a.left = d_left; a.right = d_right; a.table = d_table; a.pass = d_pass;
    a.out = memspace == XDEMHIP_HOST ? d_out : out;
    a.missing = d_miss;
    const int64_t want = (n + 256 * PB_U - 1) / (256 * PB_U);
    const unsigned blocks = (unsigned)(want < (int64_t)ctx->num_cu * 16 ? want : (int64_t)ctx->num_cu * 16);
    (void)hipEventRecord(ctx->ev_start, ctx->stream);
    const bool lds = n_edges <= PB_LDS_EDGES && n_bins <= PB_LDS_BINS;
    const size_t smem = lds ? (size_t)n_edges * 16 + (size_t)n_bins * 9 + 16 : 0;
#define XD_PB_GO(NV)                                                                                                      \
    do {                                                                                                                  \
        if (disjoint && lds) hipLaunchKernelGGL((perbin_kernel<true, true, NV>), dim3(blocks), dim3(256), smem, ctx->stream, a);   \
        else if (disjoint) hipLaunchKernelGGL((perbin_kernel<true, false, NV>), dim3(blocks), dim3(256), 0, ctx->stream, a);       \
        else if (lds) hipLaunchKernelGGL((perbin_kernel<false, true, NV>), dim3(blocks), dim3(256), smem, ctx->stream, a);         \
        else hipLaunchKernelGGL((perbin_kernel<false, false, NV>), dim3(blocks), dim3(256), 0, ctx->stream, a);                    \
    } while (0)
    if (n_var == 1) XD_PB_GO(1);
    else if (n_var == 2) XD_PB_GO(2);
    else if (n_var == 3) XD_PB_GO(3);
    else if (n_var == 4) XD_PB_GO(4);
    else XD_PB_GO(8);
#undef XD_PB_GO
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
