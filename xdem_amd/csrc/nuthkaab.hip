// nuthkaab.hip -- Nuth & Kaab (2011) inner loop on gfx950: everything of one iteration that touches the grids.
//
// Replaces, for raster-raster input (xdem/coreg/affine.py):
//   _nuth_kaab_aux_vars 412-474 (+ zero-slope removal 578-579, valid mask base.py:650-661)   -> nk_aux_kernel (once)
//   _nuth_kaab_iteration_step 477-536:  dh = ref - tba(shifted)                                  -> nk_dh_kernel
//                                       vshift = nanmedian(dh)                                  -> radix select (select.h)
//   _nuth_kaab_bin_fit 358-409:         y = dh / slope_tan, nanmean / nanstd                     -> nk_y_kernel
//                                       binned_statistic(aspect, y, np.nanmedian, 72)            -> bin ids + radix select
// The 72-point curve_fit stays on the host (scipy), exactly as in the reference.
//
// All kernels are streaming passes over row-major grids (coalesced 4-byte lanes, grid-stride), i.e. HBM-bound;
// the per-bin histograms live in LDS (ds_add_u32) and are flushed once per workgroup.  Float arithmetic that the
// reference does in the DEM dtype uses the non-contracting *_rn intrinsics so results are bit-identical to NumPy.
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "select.h"
#include "select_run.h"

namespace xd {

// ---- aux: gradient -> slope tangent, aspect, valid mask --------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nk_aux_kernel(const T* __restrict__ ref, const T* __restrict__ tba,
                                                     const uint8_t* __restrict__ inlier, int64_t H, int64_t W,
                                                     T* __restrict__ slope_tan, T* __restrict__ aspect,
                                                     uint8_t* __restrict__ valid, unsigned long long* n_valid,
                                                     int64_t p0, int64_t p1) {
    // rows [p0 / W, p1 / W) (row-aligned ranges): blockIdx.x tiles the columns, blockIdx.y strides over the rows
    unsigned long long local = 0;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row0 = p0 / W, row1 = p1 / W;
    if (j < W)
    for (int64_t i = row0 + blockIdx.y; i < row1; i += gridDim.y) {
        const int64_t p = i * W + j;
        const T c = ref[p];
        T gy, gx;
        // np.gradient, unit spacing: central differences inside, one-sided on the borders
        if (i == 0) gy = t_sub(ref[p + W], c);
        else if (i == H - 1) gy = t_sub(c, ref[p - W]);
        else gy = t_div(t_sub(ref[p + W], ref[p - W]), (T)2);
        if (j == 0) gx = t_sub(ref[p + 1], c);
        else if (j == W - 1) gx = t_sub(c, ref[p - 1]);
        else gx = t_div(t_sub(ref[p + 1], ref[p - 1]), (T)2);
        T st = t_sqrt(t_add(t_mul(gx, gx), t_mul(gy, gy)));
        // aspect = arctan2(-gx, gy) + pi: correctly rounded to the DEM dtype from the float64 arctangent
        T as = (T)atan2(-(double)gx, (double)gy);
        as = t_add(as, (T)3.14159265358979323846);
        if (fabs((double)st) <= 1e-8) st = (T)NAN;  // np.isclose(slope_tan, 0) -> NaN (affine.py:578-579)
        const bool ok = (inlier ? inlier[p] != 0 : true) && t_finite(c) && t_finite(tba[p]) && t_finite(st) && t_finite(as);
        slope_tan[p] = st;
        aspect[p] = as;
        valid[p] = ok ? 1 : 0;
        local += ok ? 1 : 0;
    }
    // wave reduction then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(n_valid, local);
}

// ---- dh at a shifted position (stated bilinear convention) + first histogram digit of its global median -------
struct DhStats {
    uint64_t asp_min, asp_max;  // order-preserving keys (widened to 64 bit) of min / max aspect among finite dh
};

template <typename T>
__global__ __launch_bounds__(256) void nk_dh_kernel(const T* __restrict__ ref, const T* __restrict__ tba,
                                                    const uint8_t* __restrict__ valid, const T* __restrict__ aspect,
                                                    int64_t H, int64_t W, double dr, double dc, T* __restrict__ dh,
                                                    DhStats* stats, int64_t p0, int64_t p1) {
    typedef typename KeyT<T>::type K;
    K kmin = ~(K)0, kmax = 0;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row0 = p0 / W, row1 = p1 / W;
    // column-invariant part of the bilinear tap position
    const double cc = t_add((double)j, dc);
    const double c0f = floor(cc);
    const double fc = t_sub(cc, c0f);
    const int64_t c0 = (int64_t)c0f;
    const bool col_ok = j < W && c0 >= 0 && c0 + 1 < W;
    const int64_t c0c = col_ok ? c0 : 0;
    if (j < W)
    for (int64_t i = row0 + blockIdx.y; i < row1; i += gridDim.y) {
        const int64_t p = i * W + j;
        const double rr = t_add((double)i, dr);
        const double r0f = floor(rr);
        const double fr = t_sub(rr, r0f);
        const int64_t r0 = (int64_t)r0f;
        const bool in = col_ok && r0 >= 0 && r0 + 1 < H;
        // all loads issued unconditionally (clamped taps) so they overlap; validity is applied afterwards
        const T* q = tba + (in ? r0 : 0) * W + c0c;
        const T a00 = q[0], a01 = q[1], a10 = q[W], a11 = q[W + 1];
        const T rv = ref[p];
        const T av = aspect[p];
        const bool ok = valid[p] && in && t_finite(a00) && t_finite(a01) && t_finite(a10) && t_finite(a11);
        const double v00 = a00, v01 = a01, v10 = a10, v11 = a11;
        const double top = t_add(v00, t_mul(fc, t_sub(v01, v00)));
        const double bot = t_add(v10, t_mul(fc, t_sub(v11, v10)));
        const double val = t_add(top, t_mul(fr, t_sub(bot, top)));
        T out = t_sub(rv, (T)val);
        if (ok && t_finite(out)) {
            const K ka = key_of(av);
            kmin = ka < kmin ? ka : kmin;
            kmax = ka > kmax ? ka : kmax;
        } else {
            out = (T)NAN;
        }
        dh[p] = out;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const K a = k_shfl_down(kmin, off), b = k_shfl_down(kmax, off);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    if ((threadIdx.x & 63) == 0) {
        if (kmin != ~(K)0) k_atomic_min(&stats->asp_min, (uint64_t)kmin);
        if (kmax != 0) k_atomic_max(&stats->asp_max, (uint64_t)kmax);
    }
}

// ---- full-grid translation resample (Coreg.apply for a pure shift) --------------------------------
// out(r, c) = bilinear(src)(r + dr, c + dc) + dz with the same stated tap convention as nk_dh_kernel.
template <typename T>
__global__ __launch_bounds__(256) void shift_bilinear_kernel(const T* __restrict__ src, int64_t H, int64_t W, double dr, double dc,
                                                             T dz, T* __restrict__ out) {
    const int64_t n = H * W;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = p / W, j = p - i * W;
        const double rr = t_add((double)i, dr), cc = t_add((double)j, dc);
        const double r0f = floor(rr), c0f = floor(cc);
        const double fr = t_sub(rr, r0f), fc = t_sub(cc, c0f);
        const int64_t r0 = (int64_t)r0f, c0 = (int64_t)c0f;
        T o = (T)NAN;
        if (r0 >= 0 && r0 + 1 < H && c0 >= 0 && c0 + 1 < W) {
            const T* q = src + r0 * W + c0;
            const T a00 = q[0], a01 = q[1], a10 = q[W], a11 = q[W + 1];
            if (t_finite(a00) && t_finite(a01) && t_finite(a10) && t_finite(a11)) {
                const double v00 = a00, v01 = a01, v10 = a10, v11 = a11;
                const double top = t_add(v00, t_mul(fc, t_sub(v01, v00)));
                const double bot = t_add(v10, t_mul(fc, t_sub(v11, v10)));
                o = t_add((T)t_add(top, t_mul(fr, t_sub(bot, top))), dz);
            }
        }
        out[p] = o;
    }
}

// ---- y = (dh - vshift) / slope_tan, aspect bin id, sums for nanmean / nanstd --------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nk_y_kernel(const T* __restrict__ dh, const T* __restrict__ slope_tan,
                                                   const T* __restrict__ aspect, int64_t n, T vshift,
                                                   const T* __restrict__ edges, int nb, T* __restrict__ y,
                                                   uint16_t* __restrict__ bins, double* sums /* [sum, sumsq] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* e = reinterpret_cast<T*>(smem);
    for (int k = threadIdx.x; k <= nb; k += blockDim.x) e[k] = edges[k];
    __syncthreads();
    const double inv_width = (double)nb / ((double)e[nb] - (double)e[0]);
    double s1 = 0.0, s2 = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const T d = dh[p];
        const T st = slope_tan[p];
        const T x = aspect[p];
        T yv = (T)NAN;
        uint16_t b = 0xFFFF;
        if (d == d) {  // dh is NaN wherever the pixel is unusable
            yv = t_div(t_sub(d, vshift), st);
            // np.digitize(x, edges) - 1 = (number of edges <= x) - 1: arithmetic guess on the uniform grid, then an exact
            // walk against the dtype-rounded edges (a step or two); a sample equal to the last edge goes to the last bin
            int idx = (int)(((double)x - (double)e[0]) * inv_width);
            idx = idx < 0 ? 0 : (idx > nb ? nb : idx);
            while (idx > 0 && !(e[idx] <= x)) --idx;
            while (idx < nb && e[idx + 1] <= x) ++idx;
            if (!(e[0] <= x)) idx = -1;
            if (idx == nb) idx = nb - 1;
            b = (idx >= 0 && idx < nb) ? (uint16_t)idx : 0xFFFF;
            s1 += (double)yv;
            s2 += (double)yv * (double)yv;
        }
        y[p] = yv;
        bins[p] = b;
    }
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sums[0], s1); atomicAdd(&sums[1], s2); }
}

// The same y / bin id computed on the fly for the bracketed selection (select_run.h): its sample and counting passes read
// dh, slope_tan and aspect directly, so the y and bin-id arrays are neither written nor re-read (nk_y_kernel + plain
// selection remain the fallback).  The counting pass, which sees every element once, also accumulates the two sums.
template <typename T> struct NkYSource {
    const T* dh;
    const T* slope_tan;
    const T* aspect;
    T vshift;
    const T* edges;  // [nb + 1], device
    double* sums;    // [sum, sumsq], device
    T* e;            // LDS copy of the edges (setup)
    double inv_width;
    struct Raw { T d, st, x; };
    struct Acc { double s1 = 0.0, s2 = 0.0; };
    static size_t lds_bytes(int nb) { return sizeof(T) * (size_t)(nb + 1) + 8; }
    __device__ __forceinline__ void setup(unsigned char* lds, int nb) {
        e = reinterpret_cast<T*>((reinterpret_cast<uintptr_t>(lds) + 7) & ~(uintptr_t)7);
        for (int k = threadIdx.x; k <= nb; k += blockDim.x) e[k] = edges[k];
        inv_width = (double)nb / ((double)edges[nb] - (double)edges[0]);
    }
    __device__ __forceinline__ void fetch(int64_t p, Raw& r) const { r.d = dh[p]; r.st = slope_tan[p]; r.x = aspect[p]; }
    __device__ __forceinline__ void blank(Raw& r) const { r.d = (T)NAN; r.st = (T)1; r.x = (T)0; }
    template <bool ACC> __device__ __forceinline__ bool eval(const Raw& r, int nb, T& v, uint16_t& b, Acc& acc) const {
        v = (T)NAN;
        b = 0xFFFF;
        if (!(r.d == r.d)) return false;
        const T yv = t_div(t_sub(r.d, vshift), r.st);
        const T x = r.x;
        int idx = (int)(((double)x - (double)e[0]) * inv_width);  // (same digitize as nk_y_kernel)
        idx = idx < 0 ? 0 : (idx > nb ? nb : idx);
        while (idx > 0 && !(e[idx] <= x)) --idx;
        while (idx < nb && e[idx + 1] <= x) ++idx;
        if (!(e[0] <= x)) idx = -1;
        if (idx == nb) idx = nb - 1;
        if (ACC) { acc.s1 += (double)yv; acc.s2 += (double)yv * (double)yv; }
        v = yv;
        if (idx < 0 || idx >= nb) return false;
        b = (uint16_t)idx;
        return yv == yv;
    }
    __device__ __forceinline__ void finish(Acc& acc) const {
        double s1 = acc.s1, s2 = acc.s2;
        for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&sums[0], s1); atomicAdd(&sums[1], s2); }
    }
};

}  // namespace xd

// ================================================================================================================
using namespace xd;

struct xdemhip_nk_plan {
    xdemhip_ctx* ctx = nullptr;
    int dtype = XDEMHIP_F32;
    int64_t H = 0, W = 0;
    int64_t p0 = 0, p1 = 0;               // this rank's pixel range [p0, p1) (whole raster unless xdemhip_nk_set_rows)
    void *ref = nullptr, *tba = nullptr;  // device (owned when own_inputs)
    uint8_t* inlier = nullptr;            // device copy kept for re-partitioning (owned when own_inputs)
    bool own_inputs = false;
    void *slope_tan = nullptr, *aspect = nullptr, *dh = nullptr, *y = nullptr;
    uint8_t* valid = nullptr;
    uint16_t* bins = nullptr;
    void* scratch = nullptr;  // edges, stats, sums, selection states, successor keys, histograms
    size_t scratch_bytes = 0;
    int max_bins = 0;
    long long n_valid0 = 0;
    xd::SelWorkspace ws;  // bracketed selection (select_run.h): sample + candidate buffers
    int bin_stat = XDEMHIP_BINSTAT_MEDIAN;
};

namespace {

// column tiles of 256 x enough row-strided workgroups to fill the chip (~16 workgroups per CU)
dim3 grid2d(const xdemhip_ctx* ctx, int64_t W, int64_t rows) {
    const int64_t gx = (W + 255) / 256;
    int64_t gy = ((int64_t)ctx->num_cu * 16 + gx - 1) / gx;
    gy = gy < 1 ? 1 : (gy > rows ? (rows > 0 ? rows : 1) : gy);
    return dim3((unsigned)gx, (unsigned)gy);
}

// np.linspace(smin, smax, nb + 1) in double (k * step + start, end point forced), cast to T -- SciPy's _bin_edges
template <typename T> void make_edges(double smin, double smax, int nb, std::vector<T>& e) {
    if (smin == smax) { smin -= 0.5; smax += 0.5; }
    e.resize(nb + 1);
    const double step = (smax - smin) / nb;
    for (int k = 0; k <= nb; ++k) e[k] = (T)((double)k * step + smin);
    e[nb] = (T)smax;
}

// aux variables + valid mask for this rank's pixel range; global valid count through the hook
template <typename T> int nk_aux_typed(xdemhip_nk_plan* P) {
    xdemhip_ctx* ctx = P->ctx;
    const int64_t n = P->p1 - P->p0;
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(P->scratch) + OFF_STATS);
    XD_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
    if (n > 0) {
        hipLaunchKernelGGL((nk_aux_kernel<T>), grid2d(ctx, P->W, n / P->W), dim3(256), 0, ctx->stream,
                           static_cast<const T*>(P->ref), static_cast<const T*>(P->tba), P->inlier, P->H, P->W,
                           static_cast<T*>(P->slope_tan), static_cast<T*>(P->aspect), P->valid, d_cnt, P->p0, P->p1);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    int rc = xd_allreduce_device(ctx, d_cnt, 1, XDEMHIP_RED_SUM_U64);
    if (rc) return rc;
    unsigned long long c = 0;
    XD_HIP_CHECK(ctx, hipMemcpyAsync(&c, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    P->n_valid0 = (long long)c;
    return XDEMHIP_OK;
}

template <typename T>
int nk_step_typed(xdemhip_nk_plan* P, double shift_x, double shift_y, double res_x, double res_y, int nb, double* vshift,
                  int64_t* n_valid, double* y_mean, double* y_std, double* edges_out, int64_t* counts, double* medians) {
    typedef typename KeyT<T>::type K;
    xdemhip_ctx* ctx = P->ctx;
    const int64_t n = P->p1 - P->p0;
    unsigned char* base = static_cast<unsigned char*>(P->scratch);
    DhStats* d_stats = reinterpret_cast<DhStats*>(base + OFF_STATS);
    double* d_sums = reinterpret_cast<double*>(base + OFF_SUMS);
    const T* dh = static_cast<const T*>(P->dh) + P->p0;
    T* y = static_cast<T*>(P->y) + P->p0;
    uint16_t* bins = P->bins + P->p0;

    // 1. dh at the shifted position: tba sampled at (row - shift_y / res_y, col + shift_x / res_x)
    DhStats hs0;
    hs0.asp_min = ~(uint64_t)0;
    hs0.asp_max = 0;
    XD_HIP_CHECK(ctx, hipMemcpyAsync(d_stats, &hs0, sizeof hs0, hipMemcpyHostToDevice, ctx->stream));
    const double dr = -shift_y / res_y, dc = shift_x / res_x;
    if (n > 0) {
        hipLaunchKernelGGL((nk_dh_kernel<T>), grid2d(ctx, P->W, n / P->W), dim3(256), 0, ctx->stream,
                           static_cast<const T*>(P->ref), static_cast<const T*>(P->tba), P->valid, static_cast<const T*>(P->aspect),
                           P->H, P->W, dr, dc, static_cast<T*>(P->dh), d_stats, P->p0, P->p1);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    int rc = xd_allreduce_device(ctx, &d_stats->asp_min, 1, XDEMHIP_RED_MIN_U64);
    if (rc) return rc;
    rc = xd_allreduce_device(ctx, &d_stats->asp_max, 1, XDEMHIP_RED_MAX_U64);
    if (rc) return rc;

    // 2. vertical shift = exact nanmedian(dh)
    std::vector<SelResult<K>> g;
    rc = run_select<T>(ctx, dh, nullptr, n, 1, base, g, &P->ws);
    if (rc) return rc;
    *n_valid = (int64_t)g[0].st.count;
    if (g[0].st.count == 0) return xd_fail(ctx, XDEMHIP_EINVAL, "The subsample contains no more valid values.");
    const double vs = median_from<T>(g[0]);
    *vshift = vs;
    XD_HIP_CHECK(ctx, hipMemcpyAsync(&hs0, d_stats, sizeof hs0, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));

    // 3. y = (dh - vshift) / slope_tan, bin ids on SciPy's edges, sums
    std::vector<T> edges;
    make_edges<T>((double)val_of((K)hs0.asp_min), (double)val_of((K)hs0.asp_max), nb, edges);
    T* d_edges = reinterpret_cast<T*>(base);
    XD_HIP_CHECK(ctx, hipMemcpyAsync(d_edges, edges.data(), sizeof(T) * (nb + 1), hipMemcpyHostToDevice, ctx->stream));
    if (P->bin_stat == XDEMHIP_BINSTAT_MEAN) {
        // 4'. bin_statistic = np.nanmean: one pass, per-bin float64 sums and counts (and the two global sums)
        if (nb > 3072) return xd_fail(ctx, XDEMHIP_EINVAL, "n_bins too large for the mean statistic");
        double* d_bsum = reinterpret_cast<double*>(base + off_hist(nb));
        unsigned long long* d_bcnt = reinterpret_cast<unsigned long long*>(d_bsum + nb);
        XD_HIP_CHECK(ctx, hipMemsetAsync(d_sums, 0, 16, ctx->stream));
        NkYSource<T> src{dh, static_cast<const T*>(P->slope_tan) + P->p0, static_cast<const T*>(P->aspect) + P->p0, (T)vs, d_edges, d_sums,
                         nullptr, 0.0};
        rc = run_bin_sums<T, NkYSource<T>>(ctx, src, n, nb, d_bsum, d_bcnt);
        if (rc) return rc;
        rc = xd_allreduce_device(ctx, d_sums, 2, XDEMHIP_RED_SUM_F64);
        if (rc) return rc;
        std::vector<double> bs(nb);
        std::vector<unsigned long long> bc(nb);
        double sums[2];
        XD_HIP_CHECK(ctx, hipMemcpyAsync(bs.data(), d_bsum, 8 * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream));
        XD_HIP_CHECK(ctx, hipMemcpyAsync(bc.data(), d_bcnt, 8 * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream));
        XD_HIP_CHECK(ctx, hipMemcpyAsync(sums, d_sums, 16, hipMemcpyDeviceToHost, ctx->stream));
        XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        const double cnt = (double)g[0].st.count;
        const double mean = sums[0] / cnt;
        const double var = sums[1] / cnt - mean * mean;
        *y_mean = mean;
        *y_std = var > 0 ? sqrt(var) : 0.0;
        for (int k = 0; k < nb; ++k) {
            counts[k] = (int64_t)bc[k];
            medians[k] = bc[k] ? (double)(T)(bs[k] / (double)bc[k]) : NAN;  // np.nanmean returns the sample dtype
        }
        for (int k = 0; k <= nb; ++k) edges_out[k] = (double)edges[k];
        return XDEMHIP_OK;
    }

    // 4. per-bin exact medians.  Bracketed route: y and the bin ids are computed on the fly by the sample / counting passes
    // (NkYSource), the counting pass accumulates the sums.  Otherwise (small grids, plain mode, a missed bracket): y and
    // bin-id arrays + plain digit passes.
    std::vector<SelResult<K>> hs;
    bool done = false;
    XD_HIP_CHECK(ctx, hipMemsetAsync(d_sums, 0, 16, ctx->stream));
    {
        NkYSource<T> src{dh, static_cast<const T*>(P->slope_tan) + P->p0, static_cast<const T*>(P->aspect) + P->p0, (T)vs, d_edges, d_sums,
                         nullptr, 0.0};
        rc = run_select_bracketed<T, NkYSource<T>>(ctx, src, n, nb, base, hs, &P->ws, &done);
        if (rc) return rc;
    }
    if (!done) {
        XD_HIP_CHECK(ctx, hipMemsetAsync(d_sums, 0, 16, ctx->stream));
        if (n > 0) {
            hipLaunchKernelGGL((nk_y_kernel<T>), dim3(grid_for(ctx, n, 256, 16)), dim3(256), sizeof(T) * (nb + 1), ctx->stream, dh,
                               static_cast<const T*>(P->slope_tan) + P->p0, static_cast<const T*>(P->aspect) + P->p0, n, (T)vs, d_edges,
                               nb, y, bins, d_sums);
            XD_HIP_CHECK(ctx, hipGetLastError());
        }
        rc = run_select_core<T>(ctx, y, bins, n, nb, base, hs, SEL_MEDIAN, nullptr);
        if (rc) return rc;
    }
    rc = xd_allreduce_device(ctx, d_sums, 2, XDEMHIP_RED_SUM_F64);
    if (rc) return rc;
    double sums[2];
    XD_HIP_CHECK(ctx, hipMemcpyAsync(sums, d_sums, 16, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const double cnt = (double)g[0].st.count;
    const double mean = sums[0] / cnt;
    const double var = sums[1] / cnt - mean * mean;
    *y_mean = mean;
    *y_std = var > 0 ? sqrt(var) : 0.0;
    for (int k = 0; k < nb; ++k) {
        counts[k] = (int64_t)hs[k].st.count;
        medians[k] = median_from<T>(hs[k]);
    }
    for (int k = 0; k <= nb; ++k) edges_out[k] = (double)edges[k];
    return XDEMHIP_OK;
}

}  // namespace

extern "C" {

void xdemhip_nk_destroy(xdemhip_nk_plan* P) {
    if (!P) return;
    (void)hipSetDevice(P->ctx->device);
    if (P->own_inputs) { (void)hipFree(P->ref); (void)hipFree(P->tba); if (P->inlier) (void)hipFree(P->inlier); }
    void* bufs[] = {P->slope_tan, P->aspect, P->dh, P->y, P->valid, P->bins, P->scratch};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    xd::sel_ws_free(P->ws);
    delete P;
}

int xdemhip_nk_create(xdemhip_ctx* ctx, const void* ref, const void* tba, const uint8_t* inlier, int dtype, int64_t H, int64_t W,
                      int memspace, xdemhip_nk_plan** out_plan, int64_t* n_valid) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (!ref || !tba || !out_plan) return xd_fail(ctx, XDEMHIP_EINVAL, "null argument");
    if (H < 2 || W < 2) return xd_fail(ctx, XDEMHIP_EINVAL, "Shape of array too small to calculate a numerical gradient, at least 2 elements are required.");
    if (dtype != XDEMHIP_F32 && dtype != XDEMHIP_F64) return xd_fail(ctx, XDEMHIP_EINVAL, "dtype must be float32 or float64");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t es = dtype == XDEMHIP_F32 ? 4 : 8;
    const size_t n = (size_t)H * (size_t)W;
    xdemhip_nk_plan* P = new xdemhip_nk_plan();
    P->ctx = ctx; P->dtype = dtype; P->H = H; P->W = W; P->p0 = 0; P->p1 = (int64_t)n;
    auto fail = [&](int code, const char* msg) { xdemhip_nk_destroy(P); return xd_fail(ctx, code, msg); };
    if (memspace == XDEMHIP_HOST) {
        P->own_inputs = true;
        if (hipMalloc(&P->ref, n * es) != hipSuccess || hipMalloc(&P->tba, n * es) != hipSuccess) return fail(XDEMHIP_ENOMEM, "hipMalloc failed");
        if (hipMemcpyAsync(P->ref, ref, n * es, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(P->tba, tba, n * es, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(XDEMHIP_EHIP, "H2D copy failed");
        if (inlier) {
            if (hipMalloc(reinterpret_cast<void**>(&P->inlier), n) != hipSuccess) return fail(XDEMHIP_ENOMEM, "hipMalloc failed");
            if (hipMemcpyAsync(P->inlier, inlier, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(XDEMHIP_EHIP, "H2D copy failed");
        }
    } else {
        P->ref = const_cast<void*>(ref);
        P->tba = const_cast<void*>(tba);
        P->inlier = const_cast<uint8_t*>(inlier);
    }
    P->max_bins = 1024;
    P->scratch_bytes = scratch_size(P->max_bins);
    if (hipMalloc(&P->slope_tan, n * es) != hipSuccess || hipMalloc(&P->aspect, n * es) != hipSuccess ||
        hipMalloc(&P->dh, n * es) != hipSuccess || hipMalloc(&P->y, n * es) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&P->valid), n) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&P->bins), n * 2) != hipSuccess || hipMalloc(&P->scratch, P->scratch_bytes) != hipSuccess)
        return fail(XDEMHIP_ENOMEM, "hipMalloc failed");
    if ((int64_t)n >= SEL_BRACKET_MIN_N && sel_ws_create(ctx, (int64_t)n, es, MAX_BINS_PER_SWEEP, P->ws) != XDEMHIP_OK)
        return fail(XDEMHIP_ENOMEM, "hipMalloc failed");
    const xdemhip_allreduce_fn hook = ctx->allreduce;
    ctx->allreduce = nullptr;  // the whole-raster pass at creation is local; xdemhip_nk_set_rows re-partitions with the hook
    int rc = dtype == XDEMHIP_F32 ? nk_aux_typed<float>(P) : nk_aux_typed<double>(P);
    ctx->allreduce = hook;
    if (rc != XDEMHIP_OK) { xdemhip_nk_destroy(P); return rc; }
    if (n_valid) *n_valid = P->n_valid0;
    *out_plan = P;
    return XDEMHIP_OK;
}

int xdemhip_nk_set_rows(xdemhip_nk_plan* P, int64_t row_begin, int64_t row_end, int64_t* n_valid) {
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (row_begin < 0 || row_end < row_begin || row_end > P->H) return xd_fail(ctx, XDEMHIP_EINVAL, "bad row range");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    P->p0 = row_begin * P->W;
    P->p1 = row_end * P->W;
    // valid mask / aux rasters outside the range are never read by this rank; recount the global number of valid pixels
    int rc = P->dtype == XDEMHIP_F32 ? nk_aux_typed<float>(P) : nk_aux_typed<double>(P);
    if (rc) return rc;
    if (n_valid) *n_valid = P->n_valid0;
    return XDEMHIP_OK;
}

int xdemhip_nk_set_statistic(xdemhip_nk_plan* P, int bin_stat) {
    if (!P) return XDEMHIP_EINVAL;
    if (bin_stat != XDEMHIP_BINSTAT_MEDIAN && bin_stat != XDEMHIP_BINSTAT_MEAN) return xd_fail(P->ctx, XDEMHIP_EINVAL, "bin statistic: 0 median, 1 mean");
    P->bin_stat = bin_stat;
    return XDEMHIP_OK;
}

int xdemhip_nk_get_aux(xdemhip_nk_plan* P, void* slope_tan, void* aspect, uint8_t* valid) {
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t es = P->dtype == XDEMHIP_F32 ? 4 : 8, n = (size_t)P->H * (size_t)P->W;
    if (slope_tan) XD_HIP_CHECK(ctx, hipMemcpy(slope_tan, P->slope_tan, n * es, hipMemcpyDeviceToHost));
    if (aspect) XD_HIP_CHECK(ctx, hipMemcpy(aspect, P->aspect, n * es, hipMemcpyDeviceToHost));
    if (valid) XD_HIP_CHECK(ctx, hipMemcpy(valid, P->valid, n, hipMemcpyDeviceToHost));
    return XDEMHIP_OK;
}

int xdemhip_nk_step(xdemhip_nk_plan* P, double shift_x, double shift_y, double res_x, double res_y, int n_bins, double* vshift,
                    int64_t* n_valid, double* y_mean, double* y_std, double* edges, int64_t* counts, double* medians) {
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (!vshift || !n_valid || !y_mean || !y_std || !edges || !counts || !medians) return xd_fail(ctx, XDEMHIP_EINVAL, "null output");
    if (n_bins < 1 || n_bins > P->max_bins) return xd_fail(ctx, XDEMHIP_EINVAL, "n_bins out of range (1..1024)");
    if (!(res_x > 0) || !(res_y > 0)) return xd_fail(ctx, XDEMHIP_EINVAL, "resolution must be > 0");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    int rc = P->dtype == XDEMHIP_F32
                 ? nk_step_typed<float>(P, shift_x, shift_y, res_x, res_y, n_bins, vshift, n_valid, y_mean, y_std, edges, counts, medians)
                 : nk_step_typed<double>(P, shift_x, shift_y, res_x, res_y, n_bins, vshift, n_valid, y_mean, y_std, edges, counts, medians);
    (void)hipEventRecord(ctx->ev_stop, ctx->stream);
    ctx->timed = (rc == XDEMHIP_OK);
    return rc;
}

// resample a raster shifted by (shift_col, shift_row) pixels (+ dz) back onto its own grid -- the
// translation case of Coreg.apply(resample=True): _reproject_horizontal_shift_samecrs, xdem/coreg/base.py:1615-1655.
int xdemhip_shift_bilinear(xdemhip_ctx* ctx, const void* src, int dtype, int64_t H, int64_t W, double shift_row_px,
                           double shift_col_px, double dz, void* out, int memspace) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (!src || !out || H < 1 || W < 1) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (dtype != XDEMHIP_F32 && dtype != XDEMHIP_F64) return xd_fail(ctx, XDEMHIP_EINVAL, "dtype must be float32 or float64");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t es = dtype == XDEMHIP_F32 ? 4 : 8, bytes = (size_t)H * (size_t)W * es;
    void *d_src = const_cast<void*>(src), *d_out = out;
    if (memspace == XDEMHIP_HOST) {
        d_src = d_out = nullptr;
        if (hipMalloc(&d_src, bytes) != hipSuccess || hipMalloc(&d_out, bytes) != hipSuccess) {
            if (d_src) (void)hipFree(d_src);
            return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
        }
        (void)hipMemcpyAsync(d_src, src, bytes, hipMemcpyHostToDevice, ctx->stream);
    }
    const int64_t n = H * W;
    XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    if (dtype == XDEMHIP_F32)
        hipLaunchKernelGGL((shift_bilinear_kernel<float>), dim3(grid_for(ctx, n, 256, 16)), dim3(256), 0, ctx->stream,
                           static_cast<const float*>(d_src), H, W, shift_row_px, shift_col_px, (float)dz, static_cast<float*>(d_out));
    else
        hipLaunchKernelGGL((shift_bilinear_kernel<double>), dim3(grid_for(ctx, n, 256, 16)), dim3(256), 0, ctx->stream,
                           static_cast<const double*>(d_src), H, W, shift_row_px, shift_col_px, dz, static_cast<double*>(d_out));
    (void)hipEventRecord(ctx->ev_stop, ctx->stream);
    ctx->timed = true;
    int rc = XDEMHIP_OK;
    if (hipGetLastError() != hipSuccess) rc = xd_fail(ctx, XDEMHIP_EHIP, "shift kernel launch failed");
    if (memspace == XDEMHIP_HOST) {
        if (rc == XDEMHIP_OK && (hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                                 hipStreamSynchronize(ctx->stream) != hipSuccess))
            rc = xd_fail(ctx, XDEMHIP_EHIP, "shift kernel / D2H failed");
        (void)hipFree(d_src);
        (void)hipFree(d_out);
    }
    return rc;
}

// Stand-alone binned nanmedian: binned_statistic(x, y, np.nanmedian, n_bins) + counts on SciPy's edges.
// (xdem/spatialstats.py:143-157 for one explanatory variable).  Non-finite (x, y) pairs are dropped like nd_binning does.
int xdemhip_binned_median(xdemhip_ctx* ctx, const void* x, const void* y, int dtype, int64_t n, int n_bins, double* edges,
                          int64_t* counts, double* medians) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (!x || !y || !edges || !counts || !medians || n <= 0) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (n_bins < 1 || n_bins > 1024) return xd_fail(ctx, XDEMHIP_EINVAL, "n_bins out of range (1..1024)");
    if (dtype != XDEMHIP_F32 && dtype != XDEMHIP_F64) return xd_fail(ctx, XDEMHIP_EINVAL, "dtype must be float32 or float64");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t es = dtype == XDEMHIP_F32 ? 4 : 8;
    // finite filter + min / max on the host, like nd_binning (O(n); this helper is not a hot path)
    double smin = INFINITY, smax = -INFINITY;
    std::vector<unsigned char> xs((size_t)n * es), ys((size_t)n * es);
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        double xv, yv;
        if (dtype == XDEMHIP_F32) { xv = static_cast<const float*>(x)[i]; yv = static_cast<const float*>(y)[i]; }
        else { xv = static_cast<const double*>(x)[i]; yv = static_cast<const double*>(y)[i]; }
        if (!std::isfinite(xv) || !std::isfinite(yv)) continue;
        memcpy(&xs[(size_t)m * es], static_cast<const unsigned char*>(x) + (size_t)i * es, es);
        memcpy(&ys[(size_t)m * es], static_cast<const unsigned char*>(y) + (size_t)i * es, es);
        smin = xv < smin ? xv : smin;
        smax = xv > smax ? xv : smax;
        ++m;
    }
    if (m == 0) { for (int k = 0; k < n_bins; ++k) { counts[k] = 0; medians[k] = NAN; } return XDEMHIP_OK; }
    void *d_x = nullptr, *d_y = nullptr, *d_one = nullptr, *d_yb = nullptr, *scratch = nullptr;
    uint16_t* d_bins = nullptr;
    auto cleanup = [&]() { void* b[] = {d_x, d_y, d_one, d_yb, scratch, d_bins}; for (void* p : b) if (p) (void)hipFree(p); };
    if (hipMalloc(&d_x, m * es) != hipSuccess || hipMalloc(&d_y, m * es) != hipSuccess || hipMalloc(&d_one, m * es) != hipSuccess ||
        hipMalloc(&d_yb, m * es) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&d_bins), m * 2) != hipSuccess ||
        hipMalloc(&scratch, scratch_size(n_bins)) != hipSuccess) { cleanup(); return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed"); }
    (void)hipMemcpyAsync(d_x, xs.data(), m * es, hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(d_y, ys.data(), m * es, hipMemcpyHostToDevice, ctx->stream);
    unsigned char* base = static_cast<unsigned char*>(scratch);
    double* d_sums = reinterpret_cast<double*>(base + OFF_SUMS);
    (void)hipMemsetAsync(d_sums, 0, 16, ctx->stream);
    int rc;
    const xdemhip_allreduce_fn hook = ctx->allreduce;
    ctx->allreduce = nullptr;  // a purely local helper
    if (dtype == XDEMHIP_F32) {
        std::vector<float> e; make_edges<float>(smin, smax, n_bins, e);
        std::vector<float> ones((size_t)m, 1.0f);
        (void)hipMemcpyAsync(d_one, ones.data(), m * es, hipMemcpyHostToDevice, ctx->stream);
        (void)hipMemcpyAsync(base, e.data(), sizeof(float) * (n_bins + 1), hipMemcpyHostToDevice, ctx->stream);
        // reuse nk_y_kernel with vshift = 0 and slope_tan = 1: y passes through unchanged ((y - 0) / 1 is exact)
        hipLaunchKernelGGL((nk_y_kernel<float>), dim3(grid_for(ctx, m, 256, 16)), dim3(256), sizeof(float) * (n_bins + 1), ctx->stream,
                           static_cast<const float*>(d_y), static_cast<const float*>(d_one), static_cast<const float*>(d_x), m, 0.0f,
                           reinterpret_cast<const float*>(base), n_bins, static_cast<float*>(d_yb), d_bins, d_sums);
        std::vector<SelResult<uint32_t>> hs;
        rc = run_select<float>(ctx, static_cast<const float*>(d_yb), d_bins, m, n_bins, base, hs);
        if (rc == XDEMHIP_OK)
            for (int k = 0; k < n_bins; ++k) { counts[k] = (int64_t)hs[k].st.count; medians[k] = median_from<float>(hs[k]); }
        for (int k = 0; k <= n_bins; ++k) edges[k] = (double)e[k];
    } else {
        std::vector<double> e; make_edges<double>(smin, smax, n_bins, e);
        std::vector<double> ones((size_t)m, 1.0);
        (void)hipMemcpyAsync(d_one, ones.data(), m * es, hipMemcpyHostToDevice, ctx->stream);
        (void)hipMemcpyAsync(base, e.data(), sizeof(double) * (n_bins + 1), hipMemcpyHostToDevice, ctx->stream);
        hipLaunchKernelGGL((nk_y_kernel<double>), dim3(grid_for(ctx, m, 256, 16)), dim3(256), sizeof(double) * (n_bins + 1), ctx->stream,
                           static_cast<const double*>(d_y), static_cast<const double*>(d_one), static_cast<const double*>(d_x), m, 0.0,
                           reinterpret_cast<const double*>(base), n_bins, static_cast<double*>(d_yb), d_bins, d_sums);
        std::vector<SelResult<uint64_t>> hs;
        rc = run_select<double>(ctx, static_cast<const double*>(d_yb), d_bins, m, n_bins, base, hs);
        if (rc == XDEMHIP_OK)
            for (int k = 0; k < n_bins; ++k) { counts[k] = (int64_t)hs[k].st.count; medians[k] = median_from<double>(hs[k]); }
        for (int k = 0; k <= n_bins; ++k) edges[k] = (double)e[k];
    }
    ctx->allreduce = hook;
    (void)hipStreamSynchronize(ctx->stream);
    cleanup();
    return rc;
}

}  // extern "C"
