// nk_onepass.h -- DEVICE code of the one-pass Nuth-Kaab step (the host side is nk_step_onepass in nuthkaab.hip): the fused data
// pass, the per-bin candidate segments, the value-bucket selection of the median of dh, the predicted brackets, and the exchange
// kernels of partitioned plans.  Part of nuthkaab.hip's translation unit: included there once, after the definitions it builds on
// (NkGeom / bi_* of nk_geom.h, DhStats, NkRowTab, nk_nearest, nk_bin_t / BinCacheRec, on_last_edge, make_edges_into).
#pragma once

namespace xd {

// ================================================================================================================
// Round 4: the ONE-PASS step (option "nk_fused", default on; single GPU, median statistic, NaN rules 0 / 1, EXT route).
// The two data passes of rounds 2-3 -- dh (written) with the counting for its median, then y = (dh - vshift) / slope_tan
// binned by aspect with the counting for the bin medians (dh re-read) -- become one: 4 (masked reference) + 4 (tba) + 4
// (slope tangent) + 2 (cached aspect bin) = 14 B/pixel, no dh raster.  The obstacle is that y needs vshift = the exact
// median of dh, which exists only after the pass.  But the 1/64 sample brackets it BEFORE the pass: v_lo <= vshift <= v_hi
// (proved afterwards by the integer counts, as before).  With v^ the bracket's midpoint and delta >= max(v^ - v_lo, v_hi -
// v^), every pixel's y lies within m = delta / slope_tan (+ rounding slack) of y^ = (dh - v^) / slope_tan, so against the
// bracket [lo_b, hi_b] of its aspect bin (from the same sample, evaluated with v^) a pixel is
//     certainly below   if y^ + m < lo_b,        certainly above   if y^ - m > hi_b,        a candidate otherwise;
// the certain ones are counted, the candidates (a few percent) staged as (dh, slope_tan, bin) triples.  Once vshift is
// known exactly (selection among the dh candidates), the resolve step (nk_resolve_scatter_kernel) evaluates the candidates' y in the
// reference's arithmetic, counts those below / inside [lo_b, hi_b] and hands the inside ones to the same exact selection as
// before: rank (k - certainly below - candidates below) among them.  All counting is integer; every rank claim is checked by
// the counts (bracket_given_kernel); a miss or an overflow anywhere sends the step to the plain route.
// Monotonicity makes the classification safe, not a tolerance: y(v) = fl(fl(dh - v) / st) is non-increasing in v for st > 0,
// and m bounds |y(v) - y^| for every v in [v_lo, v_hi] including the roundings of both evaluations (slack terms below).
// nanmean / nanstd of y (the p0 of the 72-point curve fit only) come from sums of y^ and the first-order correction in
// (v^ - vshift): sum y = sum y^ + (v^ - v) sum r, sum y^2 = sum y^^2 + 2 (v^ - v) sum y^ r + (v^ - v)^2 sum r^2, r = 1 / st.
// np.linspace(smin, smax, nb + 1) in double (k * step + start, end point forced), cast to T -- SciPy's _bin_edges
template <typename T> __host__ __device__ inline void make_edges_into(double smin, double smax, int nb, T* e) {
    if (smin == smax) { smin -= 0.5; smax += 0.5; }
    const double step = (smax - smin) / nb;
    for (int k = 0; k <= nb; ++k) e[k] = (T)((double)k * step + smin);
    e[nb] = (T)smax;
}
// Lane masks as 64-bit scalars: compares that deliver the mask itself (a C++ `bool` that also feeds a ballot is legalised into a
// 0/1 register and compared again), and selects that take such a mask as their condition
__device__ __forceinline__ unsigned long long cm_nlt(float a, float b) { unsigned long long m; asm("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ unsigned long long cm_ngt(float a, float b) { unsigned long long m; asm("v_cmp_ngt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ unsigned long long cm_nlt(double a, double b) { unsigned long long m; asm("v_cmp_nlt_f64_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ unsigned long long cm_ngt(double a, double b) { unsigned long long m; asm("v_cmp_ngt_f64_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ uint32_t sel_mask(uint32_t a, uint32_t b, unsigned long long mask) {   // lane's mask bit ? b : a
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
__device__ __forceinline__ float sel_mask(float a, float b, unsigned long long mask) { return __uint_as_float(sel_mask(__float_as_uint(a), __float_as_uint(b), mask)); }
__device__ __forceinline__ int sel_mask(int a, int b, unsigned long long mask) { return (int)sel_mask((uint32_t)a, (uint32_t)b, mask); }
__device__ __forceinline__ uint32_t* sel_mask_ptr(uint32_t* a, uint32_t* b, unsigned long long mask) {   // LDS pointers: 32-bit addresses
    typedef __attribute__((address_space(3))) uint32_t* lp;
    const uint32_t r = sel_mask((uint32_t)(uintptr_t)(lp)a, (uint32_t)(uintptr_t)(lp)b, mask);
    return (uint32_t*)(lp)(uintptr_t)r;
}

// Loads through buffer descriptors: address = descriptor base + per-lane byte offset (a loop-invariant register) + a scalar byte
// offset -- the row of the chunk -- added by the load unit itself: no vector instruction forms an address inside the row loop
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fz_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xFFFFFFFF, 0x00020000);
}
__device__ __forceinline__ float fz_bufload(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 2 /* nt */));
}
__device__ __forceinline__ double fz_bufload(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, double) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 2));
}
__device__ __forceinline__ uint16_t fz_bufload16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)soff, 2);
}

template <typename T> struct FzEps;   // relative slack that covers the roundings of y^ (fast reciprocal) and of y itself
template <> struct FzEps<float> { static constexpr float rel = 4e-6f, grow = 1.00002f, tiny = 1e-37f; };
template <> struct FzEps<double> { static constexpr double rel = 1e-14, grow = 1.0000000001, tiny = 1e-300; };
__device__ __forceinline__ float fz_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double fz_rcp(double x) { return 1.0 / x; }

// edges from the min / max aspect of this step (EXT lists), freshness of the aspect-bin cache, EXT miss -> flag
template <typename T>
__global__ void nk_fz_prep_kernel(const DhStats* stats, const unsigned long long* ext_survivors, int nb, int custom_edges, T* edges,
                                  BinCacheRec* rec, int force, unsigned long long* ctr) {
    typedef typename KeyT<T>::type K;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ext_survivors[0] == 0 || ext_survivors[1] == 0) { ctr[3] = 1ull; ctr[6] = 1ull; }   // min / max aspect unknown on this route ([6]: the plan stops using its lists)
    if (!custom_edges) make_edges_into<T>((double)val_of((K)stats->asp_min), (double)val_of((K)stats->asp_max), nb, edges);
    const double e0 = (double)edges[0], eN = (double)edges[nb];
    rec->fresh = (!force && rec->nb == nb && rec->e0 == e0 && rec->eN == eN) ? 1 : 0;
}

template <typename T> __device__ __forceinline__ uint16_t fz_digitize(const T* e, double inv_width, int nb, T x, int last_decimal) {
    int idx = (int)(((double)x - (double)e[0]) * inv_width);  // (the digitize of nk_y_kernel / NkYSource)
    idx = idx < 0 ? 0 : (idx > nb ? nb : idx);
    while (idx > 0 && !(e[idx] <= x)) --idx;
    while (idx < nb && e[idx + 1] <= x) ++idx;
    if (!(e[0] <= x)) idx = -1;
    if (idx == nb && on_last_edge<T>(x, e[nb], last_decimal)) idx = nb - 1;
    return (idx >= 0 && idx < nb) ? (uint16_t)idx : (uint16_t)0xFFFF;
}
// (re)fill of the aspect-bin cache; leaves at once while the cache is fresh
template <typename T>
__global__ __launch_bounds__(256) void nk_bin_fill_kernel(const T* __restrict__ aspect, int64_t n, const T* __restrict__ edges, int nb,
                                                          int last_decimal, const BinCacheRec* rec, nk_bin_t* __restrict__ bcache) {
    if (rec->fresh == 1) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char fz_smem[];
    T* e = reinterpret_cast<T*>(fz_smem);
    for (int k = threadIdx.x; k <= nb; k += blockDim.x) e[k] = edges[k];
    __syncthreads();
    const double inv_width = (double)nb / ((double)e[nb] - (double)e[0]);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x)
        bcache[p] = nk_bin8(fz_digitize<T>(e, inv_width, nb, aspect[p], last_decimal));
}

// positional 1/64 line sample of dh at this step's shift: slot i <-> element (i mod 8) of sampled line (i / 8); NaN where the
// pixel has no dh (the digit passes skip NaN), so no compaction and no counter
template <typename T>
__global__ __launch_bounds__(256) void nk_sample_dh_kernel(const T* __restrict__ ref_m, const T* __restrict__ tba, NkGeom g, int64_t q0, int64_t n,
                                                           double invW, int64_t n_slots, T* __restrict__ s_d, SelReset reset) {
    // (on the side: the reset of the selection that runs on this sample next -- a launch less)
    if (reset.base) select_reset_slice(reset, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = (sel_sampled_line(i >> SEL_LINE_LOG2) << SEL_LINE_LOG2) + (i & (SEL_LINE - 1));
        T out = (T)NAN;
        if (p < n) {
            const int64_t q = q0 + p;
            int64_t li, j;
            row_col(q, g.W, invW, li, j);
            const BiTap t = bi_locate(g, li + g.roff, j);
            const BiVals<T> tv = bi_load<T>(tba, t);
            T val;
            const bool in = bi_value<T>(g, tba, t, tv.a00, tv.a01, tv.a10, tv.a11, val);
            const T d = t_sub(ref_m[q], val);
            if (in && t_finite(d)) out = d;
        }
        s_d[i] = out;
    }
}

// v^ and delta from the bracket keys of the dh sample (two selection states: low end, high end)
template <typename T>
__device__ __forceinline__ bool nk_vhat_of(uint64_t sample_count, typename KeyT<T>::type lo, typename KeyT<T>::type hi, T& vhat, T& delta) {
    typedef typename KeyT<T>::type K;
    if (sample_count == 0) { vhat = (T)0; delta = (T)0; return false; }
    const K mid = (K)(lo + (K)((K)(hi - lo) >> 1));
    const double vl = (double)val_of(lo), vh = (double)val_of(hi), vm = (double)val_of(mid);
    const double d = fmax(vm - vl, vh - vm);
    vhat = val_of(mid);
    // rounded up, with room for the roundings of (dh - v) in the value dtype
    T df = (T)(d * 1.000001 + 1e-300);
    if ((double)df < d * 1.0000005) df = (T)((double)df * 1.000001);
    delta = df;
    return (vl <= vm && vm <= vh) && t_finite((T)vl) && t_finite((T)vh);   // (a bracket that reaches +-Inf: not this route)
}

// sample of y^ = (dh - v^) / slope_tan with its aspect bin, in place over the dh sample
template <typename T>
__global__ __launch_bounds__(256) void nk_sample_y_kernel(T* __restrict__ s_v, uint16_t* __restrict__ s_b, const T* __restrict__ slope_tan,
                                                          const nk_bin_t* __restrict__ bcache, int64_t n, int64_t n_slots, const T* vhat_p,
                                                          const typename KeyT<T>::type* klo_d = nullptr, const typename KeyT<T>::type* khi_d = nullptr,
                                                          T* vhat_out = nullptr, T* delta_out = nullptr, unsigned long long* ctr = nullptr,
                                                          SelReset reset = SelReset()) {
    // round 5, on the side (two launches less): v^ and delta from the dh sample's bracket, which a kernel of their own used to derive -- every
    // thread forms them (a handful of scalar operations), the first one stores them for the data pass; and the reset of the selection
    // that runs on this sample next.  (An empty sample is told from the bracket itself -- bracket_finish_body leaves {0, all-ones} --
    // not from the selection's states, which that reset is clearing.)
    T vhat;
    if (klo_d) {
        typedef typename KeyT<T>::type K;
        T delta;
        const K lo = klo_d[0], hi = khi_d[0];
        const bool ok = nk_vhat_of<T>((lo == (K)0 && hi == (K)~(K)0) ? 0u : 1u, lo, hi, vhat, delta);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            *vhat_out = vhat;
            *delta_out = delta;
            if (!ok) ctr[3] = 1ull;
        }
    } else {
        vhat = *vhat_p;
    }
    if (reset.base) select_reset_slice(reset, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = (sel_sampled_line(i >> SEL_LINE_LOG2) << SEL_LINE_LOG2) + (i & (SEL_LINE - 1));
        const T d = s_v[i];
        T y = (T)NAN;
        uint16_t b = 0xFFFF;
        if (p < n && d == d) {
            b = nk_bin16(bcache[p]);
            y = t_div(t_sub(d, vhat), slope_tan[p]);
            if (b == 0xFFFF || !(y == y)) y = (T)NAN;
        }
        s_v[i] = y;
        s_b[i] = b;
    }
}

// Both samples in one kernel, for steps whose bracket of the median of dh is PREDICTED (nk_step_onepass: predict_d): v^ is known
// before any sample is taken, so the dh sample never has to exist in memory -- one launch, and 50 MB of sample traffic, less.
// Slot for slot what nk_sample_dh_kernel followed by nk_sample_y_kernel leave in s_v / s_b.
template <typename T>
__global__ __launch_bounds__(256) void nk_sample_dy_kernel(const T* __restrict__ ref_m, const T* __restrict__ tba, NkGeom g, int64_t q0, int64_t n,
                                                           double invW, int64_t n_slots, T* __restrict__ s_v, uint16_t* __restrict__ s_b,
                                                           const T* __restrict__ slope_tan /* + q0 */, const nk_bin_t* __restrict__ bcache /* + q0 */,
                                                           const typename KeyT<T>::type* klo_d, const typename KeyT<T>::type* khi_d, T* vhat_out,
                                                           T* delta_out, unsigned long long* ctr, SelReset reset) {
    typedef typename KeyT<T>::type K;
    T vhat, delta;
    const K lo = klo_d[0], hi = khi_d[0];
    const bool ok = nk_vhat_of<T>((lo == (K)0 && hi == (K)~(K)0) ? 0u : 1u, lo, hi, vhat, delta);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *vhat_out = vhat;
        *delta_out = delta;
        if (!ok) ctr[3] = 1ull;
    }
    if (reset.base) select_reset_slice(reset, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = (sel_sampled_line(i >> SEL_LINE_LOG2) << SEL_LINE_LOG2) + (i & (SEL_LINE - 1));
        T y = (T)NAN;
        uint16_t b = 0xFFFF;
        if (p < n) {
            const int64_t q = q0 + p;
            int64_t li, j;
            row_col(q, g.W, invW, li, j);
            const BiTap t = bi_locate(g, li + g.roff, j);
            const BiVals<T> tv = bi_load<T>(tba, t);
            T val;
            const bool in = bi_value<T>(g, tba, t, tv.a00, tv.a01, tv.a10, tv.a11, val);
            const T d = t_sub(ref_m[q], val);
            if (in && t_finite(d)) {
                b = nk_bin16(bcache[p]);
                y = t_div(t_sub(d, vhat), slope_tan[p]);
                if (b == 0xFFFF || !(y == y)) y = (T)NAN;
            }
        }
        s_v[i] = y;
        s_b[i] = b;
    }
}

// Round 6: the brackets of a step from the host's PREDICTION instead of from samples (nk_step_onepass).  Values in, what the sample
// selections would have left out: bracket keys of the median of dh and of the bins' medians, the rebase shifts of the candidate
// selections, v^ and delta.  An empty bin (lo > hi) gets the bracket {0, all-ones} of a bin without sample elements.
constexpr int NK_PREDICT_MAX_BINS = 128;
template <typename T> struct NkPredicted { T dlo, dhi; T lo[NK_PREDICT_MAX_BINS], hi[NK_PREDICT_MAX_BINS]; };
template <typename T>
__global__ __launch_bounds__(64) void nk_predict_kernel(const NkPredicted<T> pr, int nb, typename KeyT<T>::type* klo_d, typename KeyT<T>::type* khi_d,
                                                        uint32_t* rbs_d, T* vhat, T* delta, typename KeyT<T>::type* klo_y,
                                                        typename KeyT<T>::type* khi_y, uint32_t* rbs_y, unsigned long long* ctr) {
    typedef typename KeyT<T>::type K;
    const int lane = threadIdx.x;
    K r = 0;
    for (int b = lane; b < nb; b += 64) {
        K lo = (K)0, hi = (K)~(K)0;
        if (pr.lo[b] <= pr.hi[b]) { lo = key_of(pr.lo[b]); hi = key_of(pr.hi[b]); }
        klo_y[b] = lo;
        khi_y[b] = hi;
        const K d = hi >= lo ? (K)(hi - lo) : (K)0;
        r = d > r ? d : r;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const K o = k_shfl_down(r, off);
        r = o > r ? o : r;
    }
    if (lane == 0) {
        *rbs_y = rebase_shift_of(r);
        const K lo = key_of(pr.dlo), hi = key_of(pr.dhi);
        *klo_d = lo;
        *khi_d = hi;
        *rbs_d = rebase_shift_of(hi >= lo ? (K)(hi - lo) : (K)0);
        T v = (T)0, d = (T)0;   // (an empty / reversed bracket raises the miss flag: the values are then never used)
        if (!(pr.dlo <= pr.dhi) || !nk_vhat_of<T>(1u, lo, hi, v, d)) ctr[3] = 1ull;
        *vhat = v;
        *delta = d;
    }
}

// The five float64 sums of the pass from its workgroups' slots, in a FIXED order (the same bits in every run): the first 256 threads of
// the calling workgroup each add every 256th slot, then a tree over the 256 partial sums.  `s_part`: 256 x 5 doubles of LDS.
constexpr int NK_SUMS_THREADS = 256;
__device__ __forceinline__ void nk_sums_reduce(const double* __restrict__ wg_sums, int n_wg, double (*s_part)[5], double* out5) {
    const int t = threadIdx.x;
    if (t < NK_SUMS_THREADS) {
        double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        for (int w = t; w < n_wg; w += NK_SUMS_THREADS)
#pragma unroll
            for (int k = 0; k < 5; ++k) a[k] += wg_sums[(size_t)w * 5 + k];
#pragma unroll
        for (int k = 0; k < 5; ++k) s_part[t][k] = a[k];
    }
    __syncthreads();
    for (int half = NK_SUMS_THREADS / 2; half > 0; half >>= 1) {
        if (t < half)
#pragma unroll
            for (int k = 0; k < 5; ++k) s_part[t][k] += s_part[t + half][k];
        __syncthreads();
    }
    if (t < 5) out5[t] = s_part[0][t];
}

#ifndef XD_NKZ_ROWS      // (measurement builds override the two pipeline constants of the one-pass kernel)
#define XD_NKZ_ROWS 8
#endif
#ifndef XD_NKZ_PF
#define XD_NKZ_PF 4
#endif
#ifndef XD_NKZ_CAP      // staging slots of the bin candidates per workgroup (float32)
#define XD_NKZ_CAP 1024
#endif
#ifndef XD_NKZ_LB       // workgroups per CU the register allocation aims at
#define XD_NKZ_LB 7
#endif
#ifndef XD_NKZ_BUFFER   // 1: loads through buffer descriptors, 0: global loads from uniform row pointers + 32-bit offsets
#define XD_NKZ_BUFFER 0
#endif
constexpr int NKZ_ROWS = XD_NKZ_ROWS;      // rows between two looks at the staging buffers
constexpr int NKZ_PF = XD_NKZ_PF;          // rows of loads in flight per wave
// staging slots per workgroup and kind (flushed once fewer than 2 x NKZ_ROWS rows would still fit; float64: static LDS stays < 48 KiB)
template <typename T> struct NkzCap { static constexpr int v = sizeof(T) == 4 ? XD_NKZ_CAP : 1024; };
template <typename T> struct FzPair { T lo, hi; };

constexpr int NKZ_CHUNK_MAX = 256;   // rows of a workgroup's chunk (row-tap table in LDS)
template <typename T, int RULE>   // (RULE 2 = rules 2 / 3 through the bad-bit mask: six more registers -> one workgroup per CU fewer instead of spills; float64 rasters: 5 / 4 -- what the allocator reaches)
__global__ __launch_bounds__(256, (sizeof(T) == 8 ? (RULE == 2 ? 4 : 5) : (RULE == 2 ? XD_NKZ_LB - 1 : XD_NKZ_LB))) void nk_fused_kernel(const T* __restrict__ ref, const T* __restrict__ tba, const T* __restrict__ slope_tan,
                                                       const nk_bin_t* __restrict__ bcache, NkGeom g, int64_t row0, int64_t row1, int64_t nbuf,
                                                       int nb, int copies, const typename KeyT<T>::type* __restrict__ klo_p,
                                                       const typename KeyT<T>::type* __restrict__ khi_p, const T* vhat_p, const T* delta_p,
                                                       const typename KeyT<T>::type* __restrict__ klo_y,
                                                       const typename KeyT<T>::type* __restrict__ khi_y, uint64_t* cnt_d /* [3] */,
                                                       uint64_t* cls_y /* [3][nb]: above, below, candidates */, T* cd_vals, int64_t cd_cap,
                                                       T* cy_d, T* cy_st, uint16_t* cy_b, int64_t cy_cap,
                                                       unsigned long long* ctr /* [1] dh candidates, [5] y candidates, [2] overflow */,
                                                       double* wg_sums /* [workgroups][5]: nk_sums_reduce adds them up in a fixed order */,
                                                       const uint64_t* __restrict__ badbits = nullptr, int64_t bad_wpr = 0) {
    typedef typename KeyT<T>::type K;
    constexpr int NKZ_CAP = NkzCap<T>::v;
    constexpr int SEG = NKZ_CAP / 4;      // staging slots of ONE wave: waves reserve in their own segment with a scalar counter --
                                          // no LDS atomic with return and its round trip on the path of every row (bin candidates are
                                          // ~7 % of the pixels: practically every row of every wave holds some)
    // (a look at the staging buffers every NKZ_ROWS rows, a flush when a segment is half full; a wave that would overrun its segment
    // before the next look -- more than half of its pixels candidates over NKZ_ROWS rows: not a raster this route is for -- raises
    // the overflow flag and the step falls through to the plain route)
    static_assert(SEG >= 2 * 64, "a segment holds at least two rows of candidates");
    // candidates of the median of dh are ~0.7 % of the pixels (half a pixel per wave and row): their segments are small and a wave
    // that would overrun its segment between two looks -- more than half of its pixels inside the bracket of the median: a raster
    // of (nearly) one dh value -- raises the overflow flag, i.e. hands the step to the plain route, which is built for that
    constexpr int SEG_D = 128;
    __shared__ NkRowTab tab[NKZ_CHUNK_MAX + 1];
    __shared__ T stage_d[4 * SEG_D + 4];     // (+ one slot per wave that nobody reads: lanes without a candidate write there, so the
    __shared__ T sy_d[NKZ_CAP + 4];          //  staging stores need no exec mask)
    __shared__ T sy_st[NKZ_CAP + 4];
    __shared__ uint16_t sy_b[NKZ_CAP + 4];
    __shared__ int s_cnt[2][4];
    __shared__ unsigned long long s_base[2];
    __shared__ unsigned long long s_red[4][3];
    __shared__ double s_sum[4][5];
    extern __shared__ __attribute__((aligned(16))) unsigned char fz_smem[];
    FzPair<T>* lohi = reinterpret_cast<FzPair<T>*>(fz_smem);                       // [nb] bin brackets as values
    uint32_t* c = reinterpret_cast<uint32_t*>(lohi + nb);                          // [copies][cs]: 3 counters per bin
    const int cs = (3 * nb) | 1;  // odd copy stride: the copies of one counter fall into different LDS banks
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int k = threadIdx.x; k < nb; k += blockDim.x) { lohi[k].lo = val_of(klo_y[k]); lohi[k].hi = val_of(khi_y[k]); }
    for (int k = threadIdx.x; k < cs * copies; k += blockDim.x) c[k] = 0;
    const K klo = *klo_p, khi = *khi_p;
    const T vhat = *vhat_p;
    const T dgrow = (T)(*delta_p * FzEps<T>::grow);
    const int64_t chunk = (row1 - row0 + gridDim.y - 1) / gridDim.y;  // <= NKZ_CHUNK_MAX (launcher)
    const int64_t i0 = row0 + (int64_t)blockIdx.y * chunk;
    // (readfirstlane: the row count and everything derived from it -- loop counters, candidate counters, flush decisions -- are
    // wave-uniform and belong on the scalar unit; the compiler does not see that through the 64-bit arithmetic above)
    const int nrow = __builtin_amdgcn_readfirstlane((int)((i0 + chunk < row1 ? i0 + chunk : row1) - i0));
    for (int r = threadIdx.x; r <= nrow && r <= NKZ_CHUNK_MAX; r += blockDim.x) {
        const BiAxis a = bi_axis(i0 + (r < nrow ? r : nrow - 1), g.dr, g.H, RULE);
        int64_t kl = a.k0 - g.roff;
        kl = (a.in && kl >= 0 && kl + a.d1 < nbuf) ? kl : 0;
        NkRowTab e;
        e.fr = a.f; e.k0l = (int)kl; e.flags = (a.in ? 1 : 0) | (a.d1 ? 2 : 0);
        const int64_t rn = nk_nearest(a.pos);
        e.rnl = (rn >= 1 && rn + 1 < g.H && rn - g.roff >= 0 && rn - g.roff < nbuf) ? (int)(rn - g.roff) : -1;
        e.pad_ = 0;
        tab[r] = e;
    }
    __syncthreads();
    if (nrow <= 0) {  // (uniform over the workgroup)
        if (threadIdx.x < 5) wg_sums[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 5 + threadIdx.x] = 0.0;
        return;
    }
    uint32_t* cc = c + (threadIdx.x % (unsigned)copies) * cs;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool jin = j < g.W;
    const BiAxis col = bi_axis(j, g.dc, g.W, RULE);
    const bool cin = col.in & jin;
    // rules 2 / 3 (RULE == 2): word and bit of this lane's nearest column in a row of the bad-bit mask (nk_badbits_kernel)
    int64_t cnc = nk_nearest(col.pos);
    cnc = cnc < -1 ? -1 : (cnc > g.W ? g.W : cnc);
    const uint32_t bad_ob = (uint32_t)(2 + (cnc >> 5)) * 4u;   // byte offset in the row (32-bit halves of the words: one register per row in flight)
    const int bad_sh = (int)(cnc & 31);
    const char* const bad_base = reinterpret_cast<const char*>(badbits);
    const int64_t bad_rowb = bad_wpr * 8;
    const unsigned long long m_cin = __builtin_amdgcn_ballot_w64(cin);
    // byte offsets of this lane's columns: 32-bit, added to uniform row pointers by the load instruction itself (a chunk spans
    // at most NKZ_CHUNK_MAX rows: (NKZ_CHUNK_MAX + 1) * W * 4 < 2^32 is the launcher's condition for this route)
    const uint32_t c0b = (cin ? (uint32_t)col.k0 : 0u) * (uint32_t)sizeof(T);
    const uint32_t c1b = c0b + (cin ? (uint32_t)col.d1 : 0u) * (uint32_t)sizeof(T);
    const uint32_t jl = jin ? (uint32_t)j : 0u;
    const double fc = col.f;
    auto hlerp = [&](T a, T b) -> double {
        const double v0 = a, v1 = b;
        return t_add(v0, t_mul(fc, t_sub(v1, v0)));
    };
    uint32_t n_all = 0, n_below = 0, n_in = 0;  // wave-uniform
    int held_d = 0, held_y = 0;                  // wave-uniform: candidates staged in this wave's segments
    T* const seg_d = stage_d + wave * SEG_D;
    const int trash_d = 4 * SEG_D + wave - wave * SEG_D;
    T* const seg_yd = sy_d + wave * SEG;
    T* const seg_ys = sy_st + wave * SEG;
    uint16_t* const seg_yb = sy_b + wave * SEG;
    const int trash = NKZ_CAP + wave - wave * SEG;       // index of this wave's unread slot, relative to its segment
    typedef __attribute__((address_space(3))) uint32_t* lds_u32p;
    auto lds_addr = [](uint32_t* p) { return (uint32_t)(uintptr_t)(lds_u32p)p; };
    auto lds_u32 = [](uint32_t a) { return (uint32_t*)(lds_u32p)(uintptr_t)a; };
    const uint32_t dummy_a = lds_addr(c + cs * copies + lane);      // 64 words behind the counters: where "add 0" goes
    const uint32_t cls_a = lds_addr(cc), cls_b = lds_addr(cc + nb), cls_c = lds_addr(cc + 2 * nb);
    // staging: [0] candidates of the median of dh (values), [1] candidates of the bin medians (dh, slope_tan, bin); a look at the
    // buffers every NKZ_ROWS rows (one barrier), a flush -- one global atomic per kind and workgroup -- when some wave's segment
    // could not take NKZ_ROWS more rows
    auto block_flush = [&](int threshold) {  // every thread of the workgroup
        if (lane == 0) { s_cnt[0][wave] = held_d; s_cnt[1][wave] = held_y; }
        __syncthreads();
        int n0[4], n1[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) { n0[w] = __builtin_amdgcn_readfirstlane(s_cnt[0][w]); n1[w] = __builtin_amdgcn_readfirstlane(s_cnt[1][w]); }
        const int m0 = max(max(n0[0], n0[1]), max(n0[2], n0[3])), m1 = max(max(n1[0], n1[1]), max(n1[2], n1[3]));
        const bool f0 = m0 > (threshold < SEG_D / 2 ? threshold : SEG_D / 2), f1 = m1 > threshold;
        if (f0 || f1) {   // (uniform)
            if (threadIdx.x == 0 && f0) s_base[0] = atomicAdd(&ctr[1], (unsigned long long)(n0[0] + n0[1] + n0[2] + n0[3]));
            if (threadIdx.x == 64 && f1) s_base[1] = atomicAdd(&ctr[5], (unsigned long long)(n1[0] + n1[1] + n1[2] + n1[3]));
            __syncthreads();
            if (f0) {
                unsigned long long b0 = s_base[0];
#pragma unroll 1
                for (int w = 0; w < 4; ++w) {
                    const int nw = __builtin_amdgcn_readfirstlane(s_cnt[0][w]);
                    for (int k = threadIdx.x; k < nw; k += blockDim.x) {
                        if ((int64_t)(b0 + k) < cd_cap) cd_vals[b0 + k] = stage_d[w * SEG_D + k];
                        else ctr[2] = 1ull;
                    }
                    b0 += (unsigned long long)nw;
                }
                held_d = 0;
            }
            if (f1) {
                unsigned long long b1 = s_base[1];
#pragma unroll 1
                for (int w = 0; w < 4; ++w) {
                    const int nw = __builtin_amdgcn_readfirstlane(s_cnt[1][w]);
                    for (int k = threadIdx.x; k < nw; k += blockDim.x) {
                        if ((int64_t)(b1 + k) < cy_cap) { cy_d[b1 + k] = sy_d[w * SEG + k]; cy_st[b1 + k] = sy_st[w * SEG + k]; cy_b[b1 + k] = sy_b[w * SEG + k]; }
                        else ctr[2] = 1ull;
                    }
                    b1 += (unsigned long long)nw;
                }
                held_y = 0;
            }
            __syncthreads();   // (the segments are free again; s_cnt is rewritten only after every thread has read it)
        }
    };
    int have = -1;
    double hl = 0.0;
    struct Pre { T b0, b1, rv, st; nk_bin_t bin; uint32_t bw; };
    Pre pre[NKZ_PF];
    // (wave-uniform, and said so: the descriptors below must sit in scalar registers -- a descriptor the compiler takes for
    // lane-varying is read back lane by lane in a loop around every load)
    const uint64_t rb0_u = (uint64_t)((i0 - g.roff) * g.W);
    const int64_t rb0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rb0_u >> 32)) << 32) |
                                  (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rb0_u));
    // buffer descriptors: the three rasters indexed by the output pixel from the chunk's first row, tba from the chunk's first
    // tap row (tap rows ascend with the output row; rows whose taps leave the raster are discarded anyway and read row k_base)
    const int k_base = __builtin_amdgcn_readfirstlane((tab[0].flags & 1) ? tab[0].k0l : 0);
#if XD_NKZ_BUFFER
    const __amdgpu_buffer_rsrc_t r_tba = fz_rsrc(tba + (int64_t)k_base * g.W);
    const __amdgpu_buffer_rsrc_t r_ref = fz_rsrc(ref + rb0);
    const __amdgpu_buffer_rsrc_t r_st = fz_rsrc(slope_tan + rb0);
    const __amdgpu_buffer_rsrc_t r_bin = fz_rsrc(bcache + rb0);
#endif
    const uint32_t wbytes = (uint32_t)g.W * (uint32_t)sizeof(T), wbytes2 = (uint32_t)g.W * (uint32_t)sizeof(nk_bin_t);
    const uint32_t ob = jl * (uint32_t)sizeof(T), ob2 = jl * (uint32_t)sizeof(nk_bin_t);
    [[maybe_unused]] auto tap_row = [&](int k) -> uint32_t { return (uint32_t)((k > k_base ? k : k_base) - k_base) * wbytes; };   // (scalar)
    auto issue = [&](int rr, Pre& q) {  // rows past the chunk repeat its last row
        const int rc = rr < nrow ? rr : nrow - 1;
        const int tk = __builtin_amdgcn_readfirstlane(tab[rc].k0l), tf = __builtin_amdgcn_readfirstlane(tab[rc].flags);
        if (RULE == 2) {
            const int rnl = __builtin_amdgcn_readfirstlane(tab[rc].rnl);
            q.bw = rnl >= 0 ? *reinterpret_cast<const uint32_t*>(bad_base + (int64_t)rnl * bad_rowb + bad_ob) : ~0u;
        } else {
            q.bw = 0;
        }
#if XD_NKZ_BUFFER
        const uint32_t so_t = tap_row(tk + ((tf >> 1) & 1));
        q.b0 = fz_bufload(r_tba, c0b, so_t, T());
        q.b1 = fz_bufload(r_tba, c1b, so_t, T());
        const uint32_t so_r = (uint32_t)rc * wbytes;
        q.rv = fz_bufload(r_ref, ob, so_r, T());
        q.st = fz_bufload(r_st, ob, so_r, T());
        q.bin = (nk_bin_t)__builtin_amdgcn_raw_buffer_load_b8(r_bin, (int)ob2, (int)((uint32_t)rc * wbytes2), 2);
#else
        const char* rowp = reinterpret_cast<const char*>(tba + (int64_t)(tk + ((tf >> 1) & 1)) * g.W);
        q.b0 = __builtin_nontemporal_load(reinterpret_cast<const T*>(rowp + c0b));
        q.b1 = __builtin_nontemporal_load(reinterpret_cast<const T*>(rowp + c1b));
        const uint32_t o_b = (uint32_t)rc * wbytes + ob;
        q.rv = __builtin_nontemporal_load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(ref + rb0) + o_b));
        q.st = __builtin_nontemporal_load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(slope_tan + rb0) + o_b));
        q.bin = __builtin_nontemporal_load(reinterpret_cast<const nk_bin_t*>(reinterpret_cast<const char*>(bcache + rb0) + ((uint32_t)rc * wbytes2 + ob2)));
#endif
    };
    // sums of y^ and of the correction terms: float32 partial sums (y^, y^ y^ folded into float64 every NKZ_PF rows; the three
    // correction sums scale a term ~1e-3 of the total and stay float32 over the chunk)
    float p_y = 0.0f, p_yy = 0.0f, p_r = 0.0f, p_yr = 0.0f, p_rr = 0.0f;
    double a_y = 0.0, a_yy = 0.0;
#pragma unroll
    for (int u = 0; u < NKZ_PF; ++u) issue(u, pre[u]);
    for (int r0 = 0; r0 < nrow; r0 += NKZ_PF) {
#pragma unroll
        for (int u = 0; u < NKZ_PF; ++u) {
            const int r = r0 + u;
            if (r < nrow) {   // (uniform)
                const T b0v = pre[u].b0, b1v = pre[u].b1, rv = pre[u].rv, stv = pre[u].st;
                const nk_bin_t bin = pre[u].bin;
                const unsigned long long m_clean = RULE != 2 ? ~0ull : __builtin_amdgcn_ballot_w64(((pre[u].bw >> bad_sh) & 1u) == 0);
                issue(r + NKZ_PF, pre[u]);
                const int k0l = __builtin_amdgcn_readfirstlane(tab[r].k0l), fl = __builtin_amdgcn_readfirstlane(tab[r].flags);
                const double fr = tab[r].fr;
                double top;
                if (have == k0l) {
                    top = hl;
                } else {  // chunk start, or a step of the tap row other than +1: fetch the upper row
#if XD_NKZ_BUFFER
                    const uint32_t so_u = tap_row(k0l);
                    top = hlerp(fz_bufload(r_tba, c0b, so_u, T()), fz_bufload(r_tba, c1b, so_u, T()));
#else
                    const char* up = reinterpret_cast<const char*>(tba + (int64_t)k0l * g.W);
                    top = hlerp(*reinterpret_cast<const T*>(up + c0b), *reinterpret_cast<const T*>(up + c1b));
#endif
                }
                double bot = top;
                if (fl & 2) bot = hlerp(b0v, b1v);
                have = k0l + ((fl >> 1) & 1);
                hl = bot;
                const T val = (T)t_add(top, t_mul(fr, t_sub(bot, top)));
                const T out = t_sub(rv, val);
                // lane masks as 64-bit scalars; per-lane choices are selects on those masks, stores and the counter update run
                // unmasked (lanes that have nothing to say write to a slot / add 0 to a word nobody reads): one basic block per row
                const unsigned long long m_row = (fl & 1) ? (m_cin & m_clean) : 0ull;
                const unsigned long long m_ok = __builtin_amdgcn_ballot_w64(t_finite(out)) & m_row;
                const K key = key_of(out);
                const unsigned long long m_lt = __builtin_amdgcn_ballot_w64(key < klo), m_le = __builtin_amdgcn_ballot_w64(key <= khi);
                const unsigned long long mask = m_ok & ~m_lt & m_le;
                n_all += (uint32_t)__popcll(m_ok);
                n_below += (uint32_t)__popcll(m_ok & m_lt);
                if (mask) {   // (uniform)
                    const int cn = __popcll(mask);
                    n_in += (uint32_t)cn;
                    if (held_d + cn <= SEG_D) {   // (uniform)
                        const int pos = held_d + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                        seg_d[sel_mask(trash_d, pos, mask)] = out;
                        held_d += cn;
                    } else if (lane == 0) {
                        ctr[2] = 1ull;
                    }
                }
                // ---- the bin side: y^ with its margin against the bracket of the pixel's aspect bin
                const T rr = fz_rcp(stv);
                const T yh = (T)(out - vhat) * rr;
                const T m = (T)(dgrow * rr) + (T)(fabs(yh) * FzEps<T>::rel);   // (m = 0 only for y^ = 0 under an exact v^: then y = 0 too)
                const unsigned long long m_yb = m_ok & __builtin_amdgcn_ballot_w64(bin != NK_NOBIN) & __builtin_amdgcn_ballot_w64(yh == yh);
                const uint32_t binx = sel_mask(0u, (uint32_t)bin, m_yb);
                const FzPair<T> lh = lohi[binx];
                const unsigned long long m_nb = cm_nlt((T)(yh + m), lh.lo), m_na = cm_ngt((T)(yh - m), lh.hi);   // not certainly below / above
                // class row of this lane's counter copy: 0 above, 1 below, 2 candidate (the three row bases are loop-invariant)
                const uint32_t rowa = sel_mask(cls_b, sel_mask(cls_a, cls_c, m_na), m_nb);
                // (lanes without a bin add 0 to a word of their own: no exec mask, no same-address pile-up)
                atomicAdd(lds_u32(sel_mask(dummy_a, rowa + (binx << 2), m_yb)), sel_mask(0u, 1u, m_yb));
                const unsigned long long my = m_yb & m_nb & m_na;
                if (my) {   // (uniform)
                    const int cn = __popcll(my);
                    if (held_y + cn <= SEG) {   // (uniform)
                        const int pos = sel_mask(trash, held_y + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(my >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)my, 0u)), my);
                        seg_yd[pos] = out; seg_ys[pos] = stv; seg_yb[pos] = (uint16_t)bin;
                        held_y += cn;
                    } else if (lane == 0) {
                        ctr[2] = 1ull;
                    }
                }
                const float yf = sel_mask(0.0f, (float)yh, m_ok), rf = sel_mask(0.0f, (float)rr, m_ok);
                p_y += yf; p_yy = fmaf(yf, yf, p_yy);
                p_r += rf; p_yr = fmaf(yf, rf, p_yr); p_rr = fmaf(rf, rf, p_rr);
            }
            // (r is uniform over the workgroup: every wave walks the same rows) room for NKZ_ROWS more rows must remain
            if (((r + 1) % NKZ_ROWS) == 0 && r + 1 < nrow) block_flush(SEG / 2);
        }
        a_y += (double)p_y; a_yy += (double)p_yy;
        p_y = 0.0f; p_yy = 0.0f;
    }
    block_flush(0);
    double sv[5] = {a_y, a_yy, (double)p_r, (double)p_yr, (double)p_rr};
#pragma unroll
    for (int k = 0; k < 5; ++k)
        for (int off = 32; off > 0; off >>= 1) sv[k] += __shfl_down(sv[k], off);
    if (lane == 0) {
        s_red[wave][0] = n_all; s_red[wave][1] = n_below; s_red[wave][2] = n_in;
#pragma unroll
        for (int k = 0; k < 5; ++k) s_sum[wave][k] = sv[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long q0 = 0, q1 = 0, q2 = 0;
        for (int w = 0; w < 4; ++w) { q0 += s_red[w][0]; q1 += s_red[w][1]; q2 += s_red[w][2]; }
        if (q0) atomicAdd(reinterpret_cast<unsigned long long*>(&cnt_d[0]), q0);
        if (q1) atomicAdd(reinterpret_cast<unsigned long long*>(&cnt_d[1]), q1);
        if (q2) atomicAdd(reinterpret_cast<unsigned long long*>(&cnt_d[2]), q2);
    }
    // (a slot per workgroup, not an atomic: float64 adds in the order the workgroups happen to finish gave every run of the same step its
    //  own last bits of nanmean / nanstd of y -- the p0 of the curve fit -- and a whole fit its own last digits of the shift)
    if (threadIdx.x < 5)
        wg_sums[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 5 + threadIdx.x] = s_sum[0][threadIdx.x] + s_sum[1][threadIdx.x] + s_sum[2][threadIdx.x] + s_sum[3][threadIdx.x];
    for (int k = threadIdx.x; k < 3 * nb; k += blockDim.x) {
        unsigned long long t = 0;
        for (int q = 0; q < copies; ++q) t += c[q * cs + k];
        if (t) atomicAdd(reinterpret_cast<unsigned long long*>(&cls_y[k]), t);
    }
}




// ---- round 5: the bin candidates partitioned by bin; ONE workgroup per bin selects its exact median ----------------------------
// After round 4's resolve kernel (nk_resolve_kernel, gone since round 6) the exact selection among the candidates of the 72 bins used to run as select_enqueue: three more digit
// passes over ALL candidate slots (the resolved-away ones left as NaN) with the whole [72][256] LDS table zeroed and flushed by every
// workgroup of every pass, an advance kernel behind each, the successor pass -- 10 launches, ~150 us of the step.  The fused kernel
// already counts the candidates per bin (cls[2][b]), so the resolve kernel can write the kept y values INTO PER-BIN SEGMENTS
// (exclusive scan of those counts; a workgroup reserves its places with one global atomic per bin and round) and histogram them on
// the way into 256 VALUE buckets of the bin's bracket: bucket(y) = floor((y - lo_b) * 256 / (hi_b - lo_b)) in float64, a monotone
// map.  (Digits of the float KEYS would not do: the bin medians of y lie around zero, a bracket that straddles zero spans every
// binade of the key space, and most of a bin would sit in two or three leading-digit values.)  One workgroup per bin then reads its
// segment ONCE: the histogram names the bucket that holds the wanted rank, the few hundred values in it go to LDS as keys, and the
// exact order statistic and its successor are settled there by the usual digit passes.  Same integers and the same order of keys
// as on the other routes: results are identical bit for bit.
constexpr int BINSEG_U = 16;            // values per thread and round of the scatter (one reservation per bin, workgroup and round)
constexpr int BINSEG_CTR_STRIDE = 16;   // the per-bin place counters sit one 128-byte line apart (same-line atomics serialise)
struct BinsegMap { double lo, scale; };
template <typename T, typename K> __device__ __forceinline__ BinsegMap binseg_map(K klo, K khi) {
    BinsegMap m;
    m.lo = (double)val_of(klo);
    const double w = (double)val_of(khi) - m.lo;
    m.scale = w > 0.0 ? 256.0 / w : 0.0;
    return m;
}
template <typename T> __device__ __forceinline__ int binseg_bucket(T y, const BinsegMap& m) {   // y inside [lo, hi]
    const double t = ((double)y - m.lo) * m.scale;   // (monotone in y; >= 0)
    const int d = (int)t;
    return d > 255 ? 255 : (d < 0 ? 0 : d);
}
template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void nk_resolve_scatter_kernel(const T* __restrict__ c_v /* dh */, const T* __restrict__ c_st,
                                                                          const uint16_t* __restrict__ c_b, int64_t cap, const unsigned long long* n_dev,
                                                                          const T* vshift_p, int nb, const typename KeyT<T>::type* __restrict__ klo,
                                                                          const typename KeyT<T>::type* __restrict__ khi, const uint64_t* __restrict__ cls /* [3][nb] */,
                                                                          uint64_t* res /* [2][nb] */, unsigned long long* seg_ctr /* [nb], zeroed */,
                                                                          T* __restrict__ seg_v, int64_t seg_cap, unsigned long long* ctr /* [2] overflow */,
                                                                          const uint64_t* __restrict__ cls_lay = nullptr /* partitioned plans: THIS rank's classes (the
                                                                              segments hold this rank's candidates; `cls` is the sum over the ranks by then) */) {
    typedef typename KeyT<T>::type K;
    extern __shared__ __attribute__((aligned(16))) unsigned char fz_smem[];
    unsigned long long* off = reinterpret_cast<unsigned long long*>(fz_smem);   // [nb] first slot of the bin's segment
    unsigned long long* gbase = off + nb;                                       // [nb] this round's places inside the segment
    unsigned long long* room = gbase + nb;                                      // [nb] slots of the segment
    K* lo = reinterpret_cast<K*>(room + nb);
    K* hi = lo + nb;
    uint32_t* c = reinterpret_cast<uint32_t*>(hi + nb);   // [2][nb] below / inside (this workgroup)
    uint32_t* lcnt = c + 2 * nb;                           // [nb] kept values of the round
    const uint64_t* lay = cls_lay ? cls_lay : cls;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        lo[k] = klo[k]; hi[k] = khi[k]; lcnt[k] = 0u; room[k] = lay[2 * nb + k];
    }
    for (int k = threadIdx.x; k < 2 * nb; k += blockDim.x) c[k] = 0u;
    if (threadIdx.x == 0) {
        unsigned long long acc = 0;
        for (int k = 0; k < nb; ++k) { off[k] = acc; acc += lay[2 * nb + k]; }
    }
    __syncthreads();
    const unsigned long long m = *n_dev;
    const int64_t n = m < (unsigned long long)cap ? (int64_t)m : cap;
    const T vshift = *vshift_p;
    constexpr int U = BINSEG_U;
    const int64_t step = (int64_t)gridDim.x * blockDim.x * U;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x * U; base < n; base += step) {   // (uniform over the workgroup)
        T dv[U], sv[U];
        uint16_t bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {   // (all of a round's loads first)
            const int64_t p = base + (int64_t)u * blockDim.x + threadIdx.x;
            const bool have = p < n;
            dv[u] = have ? c_v[p] : (T)NAN;
            sv[u] = have ? c_st[p] : (T)1;
            bv[u] = have ? c_b[p] : (uint16_t)0xFFFF;
        }
        T yk[U];
        int bk[U];
        uint32_t li[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bk[u] = -1;
            li[u] = 0u;
            const int b = (int)bv[u];
            const T y = t_div(t_sub(dv[u], vshift), sv[u]);
            yk[u] = y;
            if (y == y && b < nb) {
                const K key = key_of(y);
                if (key < lo[b]) atomicAdd(&c[b], 1u);
                else if (key <= hi[b]) {
                    atomicAdd(&c[nb + b], 1u);
                    li[u] = atomicAdd(&lcnt[b], 1u);
                    bk[u] = b;
                }
            }
        }
        __syncthreads();
        for (int k = threadIdx.x; k < nb; k += blockDim.x) {
            const uint32_t cnt = lcnt[k];
            gbase[k] = cnt ? atomicAdd(&seg_ctr[(size_t)k * BINSEG_CTR_STRIDE], (unsigned long long)cnt) : 0ull;   // (one cache line per bin's counter)
            lcnt[k] = 0u;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (bk[u] >= 0) {
                const unsigned long long q = gbase[bk[u]] + li[u];
                const unsigned long long pos = off[bk[u]] + q;
                if (q < room[bk[u]] && (int64_t)pos < seg_cap) seg_v[pos] = yk[u];
                else ctr[2] = 1ull;   // (cannot happen for consistent counters: the step then takes the plain route)
            }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < 2 * nb; k += blockDim.x)
        if (c[k]) atomicAdd(reinterpret_cast<unsigned long long*>(&res[k]), (unsigned long long)c[k]);
}

constexpr int BINSEL_COPIES = 8;
constexpr int BINSEL_KEY_BYTES = 96 * 1024;   // keys of the leading digit's group, held in LDS
template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void nk_bin_select_kernel(const T* __restrict__ seg_v, const uint64_t* __restrict__ cls /* [3][nb] */,
                                                                     const uint64_t* __restrict__ res /* [2][nb] */, const unsigned long long* seg_ctr, int nb,
                                                                     const typename KeyT<T>::type* __restrict__ klo, const typename KeyT<T>::type* __restrict__ khi,
                                                                     const uint32_t* rbs_p,
                                                                     SelState<typename KeyT<T>::type>* st_out, uint64_t* succ_out, uint64_t* cnt_out /* [3][nb] */,
                                                                     unsigned long long* ctr /* [2] overflow, [3] miss */,
                                                                     int64_t seg_stride = 0 /* > 0: bin b's values start at b * seg_stride (partitioned plans) */) {
    typedef typename KeyT<T>::type K;
    constexpr int P = KeyT<T>::passes;
    constexpr int CAPK = BINSEL_KEY_BYTES / (int)sizeof(K);
    extern __shared__ __attribute__((aligned(16))) unsigned char fz_smem[];
    K* keys = reinterpret_cast<K*>(fz_smem);                                      // [CAPK]
    uint32_t* h = reinterpret_cast<uint32_t*>(keys + CAPK);                       // [BINSEL_COPIES][257]
    unsigned long long* s_tot = reinterpret_cast<unsigned long long*>(h + BINSEL_COPIES * (SEL_RADIX + 1) + ((BINSEL_COPIES * (SEL_RADIX + 1)) & 1));   // [256]
    unsigned long long* s_pick = s_tot + SEL_RADIX;                               // digit, elements below it, elements in it
    unsigned long long* s_off = s_pick + 4;
    K* s_min = reinterpret_cast<K*>(s_off + 1);
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_off + 2);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const uint64_t total = cls[b] + cls[nb + b] + cls[2 * nb + b], lt = cls[nb + b] + res[b], in = res[nb + b];
    if (tid == 0) {
        cnt_out[b] = total; cnt_out[nb + b] = lt; cnt_out[2 * nb + b] = in;
        unsigned long long acc = 0;
        for (int k = 0; k < b; ++k) acc += cls[2 * nb + k];
        *s_off = seg_stride > 0 ? (unsigned long long)b * (unsigned long long)seg_stride : acc;
        *s_min = ~(K)0;
        *s_cnt = 0u;
    }
    SelState<K> s;
    s.prefix = 0; s.rank = 0; s.count = 0; s.n_le = 0; s.group = 0;
    bool run = total != 0;
    if (run) {   // (the rule of bracket_given_kernel)
        const uint64_t k = (total - 1) / 2, need = (total & 1) ? k : k + 1;
        if (lt > k || need - lt >= in) { run = false; if (tid == 0) ctr[3] = 1ull; }
        else { s.rank = k - lt; s.count = in; s.group = in; }
        if (run && seg_ctr[(size_t)b * BINSEG_CTR_STRIDE] != in) { run = false; if (tid == 0) ctr[2] = 1ull; }   // (the segment does not hold what the counters say)
    }
    // the leading digit from the histogram the resolve kernel filled (one wave: lane l owns buckets 4 l .. 4 l + 3)
    auto pick = [&](auto count_of) {   // every thread of the FIRST wave calls it; the result lands in s_pick
        unsigned long long cq[4], mine = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { cq[q] = count_of(4 * lane + q); mine += cq[q]; }
        unsigned long long incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const unsigned long long excl = incl - mine;
        if (s.rank >= excl && s.rank < incl) {   // (exactly one lane for a consistent histogram)
            unsigned long long cum = excl;
            int q = 0;
            if (cum + cq[0] <= s.rank) { cum += cq[0]; q = 1;
                if (cum + cq[1] <= s.rank) { cum += cq[1]; q = 2;
                    if (cum + cq[2] <= s.rank) { cum += cq[2]; q = 3; } } }
            s_pick[0] = (unsigned long long)(4 * lane + q);
            s_pick[1] = cum;
            s_pick[2] = q == 0 ? cq[0] : (q == 1 ? cq[1] : (q == 2 ? cq[2] : cq[3]));
        }
    };
    __syncthreads();
    if (!run) {   // (uniform over the workgroup)
        if (tid == 0) { s.count = 0; st_out[b] = s; succ_out[b] = ~(uint64_t)0; }
        return;
    }
    const T* v = seg_v + *s_off;
    const int64_t n = (int64_t)in;
    const K lo = klo[b];
    const BinsegMap mp = binseg_map<T, K>(lo, khi[b]);   // 256 value buckets over the bin's bracket
    const int rbs = (int)*rbs_p;                         // the rebase the host expects the state in
    uint32_t* hc = h + (tid & (BINSEL_COPIES - 1)) * (SEL_RADIX + 1);
    constexpr int U = 8;
    // first read of the segment: how many values per bucket -> the bucket that holds the wanted rank
    for (int k = tid; k < BINSEL_COPIES * (SEL_RADIX + 1); k += blockDim.x) h[k] = 0u;
    __syncthreads();
    for (int64_t i0 = tid; i0 < n; i0 += (int64_t)blockDim.x * U) {
        T x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + (int64_t)u * blockDim.x;
            x[u] = i < n ? v[i] : (T)NAN;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (x[u] == x[u]) atomicAdd(&hc[binseg_bucket<T>(x[u], mp)], 1u);
    }
    __syncthreads();
    if (tid < SEL_RADIX) {
        unsigned long long t = 0;
#pragma unroll
        for (int q = 0; q < BINSEL_COPIES; ++q) t += h[q * (SEL_RADIX + 1) + tid];
        s_tot[tid] = t;
    }
    __syncthreads();
    if (tid < 64) pick([&](int d) { return s_tot[d]; });
    __syncthreads();
    if (s_pick[2] > (unsigned long long)CAPK) {   // (a bucket beyond the LDS buffer -- ties en masse: the plain route takes the step)
        if (tid == 0) { ctr[2] = 1ull; s.count = 0; st_out[b] = s; succ_out[b] = ~(uint64_t)0; }
        return;
    }
    const int d1 = (int)s_pick[0];
    s.prefix = 0;
    s.n_le = s_pick[1];
    s.rank -= s_pick[1];
    s.group = s_pick[2];
    // second read (from the L2): keys of the chosen bucket's values into LDS, the smallest key of the buckets above on the way
    K mn = ~(K)0;
    __syncthreads();
    for (int64_t i0 = tid; i0 < n; i0 += (int64_t)blockDim.x * U) {
        T x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + (int64_t)u * blockDim.x;
            x[u] = i < n ? v[i] : (T)NAN;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (x[u] != x[u]) continue;
            const K key = key_of(x[u]);
            const int d = binseg_bucket<T>(x[u], mp);
            if (d == d1) {
                const uint32_t pos = atomicAdd(s_cnt, 1u);
                if (pos < (uint32_t)CAPK) keys[pos] = key;
            } else if (d > d1 && key < mn) {
                mn = key;
            }
        }
    }
    __syncthreads();
    const int g = (int)(*s_cnt < (uint32_t)CAPK ? *s_cnt : (uint32_t)CAPK);   // (= s.group)
    for (int p = 0; p < P; ++p) {   // exact selection among the bucket's keys (a few hundred, in LDS)
        const int shift = 8 * (P - 1 - p);
        for (int k = tid; k < BINSEL_COPIES * (SEL_RADIX + 1); k += blockDim.x) h[k] = 0u;
        __syncthreads();
        const K himask = p == 0 ? (K)0 : (K)(~(K)0 << (shift + 8));
        for (int i = tid; i < g; i += blockDim.x) {
            const K key = keys[i];
            if ((key & himask) == s.prefix) atomicAdd(&hc[(int)((key >> shift) & 0xFF)], 1u);
        }
        __syncthreads();
        if (tid < SEL_RADIX) {
            unsigned long long t = 0;
#pragma unroll
            for (int q = 0; q < BINSEL_COPIES; ++q) t += h[q * (SEL_RADIX + 1) + tid];
            s_tot[tid] = t;
        }
        __syncthreads();
        if (tid < 64) pick([&](int d) { return s_tot[d]; });
        __syncthreads();
        s.prefix |= (K)s_pick[0] << shift;
        s.n_le += s_pick[1];
        s.rank -= s_pick[1];
        s.group = s_pick[2];
        __syncthreads();
    }
    s.n_le += s.group;   // every digit fixed: group = the selected key's duplicates
    // successor: the smallest key above the selected one -- inside its bucket (LDS) or, failing that, the smallest key of the buckets
    // above (collected while reading the segment)
    for (int i = tid; i < g; i += blockDim.x) {
        const K key = keys[i];
        if (key > s.prefix && key < mn) mn = key;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const K t = k_shfl_down(mn, o);
        mn = t < mn ? t : mn;
    }
    if (lane == 0 && mn != ~(K)0) k_atomic_min(s_min, mn);
    __syncthreads();
    if (tid == 0) {
        // (handed back in the rebase of the other routes: offset from the bin's low end << rbs)
        s.prefix = (K)((K)(s.prefix - lo) << rbs);
        st_out[b] = s;
        succ_out[b] = *s_min == ~(K)0 ? ~(uint64_t)0 : (uint64_t)(K)((K)(*s_min - lo) << rbs);
    }
}

// ---- round 5: the median of dh among its candidates in THREE launches -------------------------------------------------------------
// The candidates of the median of dh (every dh inside the sample bracket [lo, hi]: a few 1e5 values at 4e8 pixels) went through the
// generic selection: a reset, four digit passes with an advance kernel each, the successor pass, the vshift kernel -- eleven
// dependent launches for 2 MB of data.  Here, three: `nk_dhsel_hist_kernel` counts them into 4096 VALUE buckets of the bracket (the
// monotone map of the per-bin segments above, 16 x finer); `nk_dhsel_gather_kernel` -- every workgroup -- scans that histogram for
// the bucket that holds the wanted rank, appends the bucket's keys (a few hundred) to a small buffer and notes the smallest key of
// the buckets above; `nk_dhsel_final_kernel` (one workgroup) settles the exact order statistic and its successor among those keys
// in LDS and writes vshift (value, count, flags: the step's info block).  (Gather and final in one launch -- the last workgroup to finish, by a
// ticket, doing the final part -- needs the keys, plain stores of many workgroups, published by device-scope fences: on this
// multi-XCD part a fence writes back the issuing XCD's whole L2, 38-55 us for 64 workgroups, measured; a kernel boundary is cheaper.)
// Same integers, same keys: the result is the generic selection's bit for bit (GPU tests: every route agrees).
constexpr int DSEL_BUCKETS = 4096;
constexpr int DSEL_CAP = 8192;        // keys of the chosen bucket (more -- ties en masse -- send the step to the plain route)
constexpr int DSEL_HDR_WORDS = 4;     // 64-bit words: [0] ~(smallest key IN the bucket), [1] keys appended, [2] ~(smallest key above the bucket), [3] largest key in the bucket; then the histogram
// Two forms of the monotone bucket map.  VALUE buckets (floor((v - lo) * 4096 / (hi - lo)) in float64) for brackets that straddle
// zero or span many binades -- there the float KEYS are spread exponentially and most values would sit in a few key buckets.
// KEY buckets ((key - klo) >> s, s the smallest shift that brings the bracket under 4096 buckets) for brackets inside at most three
// binades of one sign -- there keys are evenly dense, and this is where ties concentrate: a well-aligned pair (the state every fit
// converges to) has dh = offset + small noise, a bracket a few hundred float32 values wide holding a million candidates.  With
// s = 0 a bucket IS a key: the histogram alone gives the selected key, its duplicates and its successor -- nothing is gathered,
// whatever the multiplicity (`exact`).  (Found by bench.py's whole-fit leg, whose later iterations fell to the plain route with
// value buckets only: 2.4e6 candidates on ~300 distinct keys overflow any per-bucket key buffer.)
template <typename K> struct DselMap { double lo, scale; K klo; int shift; int keyspace; };
template <typename T, typename K> __device__ __forceinline__ DselMap<K> dsel_map(K klo, K khi) {
    DselMap<K> m;
    m.klo = klo;
    m.lo = (double)val_of(klo);
    const double w = (double)val_of(khi) - m.lo;
    m.scale = w > 0.0 ? (double)DSEL_BUCKETS / w : 0.0;
    constexpr int MANT = sizeof(K) == 4 ? 23 : 52;
    const K top = (K)1 << (sizeof(K) * 8 - 1);
    const K span = khi >= klo ? (K)(khi - klo) : (K)0;
    m.keyspace = (int)(((klo ^ khi) & top) == 0 && span < (K)3 << MANT);
    int bits = 0;
    for (K r = span; r; r >>= 1) ++bits;   // (bits needed for span)
    m.shift = bits > 12 ? bits - 12 : 0;
    return m;
}
template <typename T, typename K> __device__ __forceinline__ int dsel_bucket(T v, const DselMap<K>& m) {   // v inside [lo, hi]; monotone in v
    int d;
    if (m.keyspace) d = (int)((K)(key_of(v) - m.klo) >> m.shift);
    else d = (int)(((double)v - m.lo) * m.scale);
    return d > DSEL_BUCKETS - 1 ? DSEL_BUCKETS - 1 : (d < 0 ? 0 : d);
}
template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void nk_dhsel_hist_kernel(const T* __restrict__ cd, int64_t cap, const unsigned long long* n_dev,
                                                                     const typename KeyT<T>::type* klo, const typename KeyT<T>::type* khi,
                                                                     uint32_t* hist /* [DSEL_BUCKETS], zeroed */,
                                                                     const double* __restrict__ wg_sums = nullptr, int n_wg = 0, double* sums_out = nullptr) {
    typedef typename KeyT<T>::type K;
    __shared__ uint32_t h[DSEL_BUCKETS];
    __shared__ double s_part[NK_SUMS_THREADS][5];
    // (on the side, first launch behind the pass: its float64 sums in a fixed order -- one workgroup's job, the last one's)
    if (wg_sums && blockIdx.x == gridDim.x - 1) nk_sums_reduce(wg_sums, n_wg, s_part, sums_out);
    for (int k = threadIdx.x; k < DSEL_BUCKETS; k += blockDim.x) h[k] = 0u;
    __syncthreads();
    const unsigned long long m = *n_dev;
    const int64_t n = m < (unsigned long long)cap ? (int64_t)m : cap;
    const DselMap<K> mp = dsel_map<T, K>(*klo, *khi);
    constexpr int U = 8;
    const int64_t step = (int64_t)blockDim.x * U;
    for (int64_t base = (int64_t)blockIdx.x * step; base < n; base += (int64_t)gridDim.x * step) {
        T x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + (int64_t)u * blockDim.x + threadIdx.x;
            x[u] = i < n ? cd[i] : (T)NAN;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (x[u] == x[u]) atomicAdd(&h[dsel_bucket<T, K>(x[u], mp)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < DSEL_BUCKETS; k += blockDim.x)
        if (h[k]) atomicAdd(&hist[k], h[k]);
}

// wanted rank among the candidates and the bucket that holds it: every workgroup of both kernels below comes to the same conclusions
// from the same counters and the same (complete) histogram
struct DselWhere { bool run; int bucket; unsigned long long rank, below, group; };
template <typename T>
__device__ __forceinline__ DselWhere dsel_locate(const uint64_t* cnt, int64_t n, const uint32_t* hist, unsigned long long* s_pick /* [3] */,
                                                 unsigned long long* s_wsum /* [16] */, unsigned long long* ctr, bool report, bool exact,
                                                 bool check_n = true /* false on partitioned plans: `n` is this rank's share of the candidates */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    DselWhere w;
    w.run = false; w.bucket = 0; w.rank = 0; w.below = 0; w.group = 0;
    const uint64_t total = cnt[0], lt = cnt[1], in = cnt[2];
    if (total == 0) return w;
    {   // (the rule of bracket_given_kernel)
        const uint64_t k = (total - 1) / 2, need = (total & 1) ? k : k + 1;
        if (lt > k || need - lt >= in) { if (report && tid == 0) ctr[3] = 1ull; return w; }
        w.rank = k - lt;
        if (check_n && (uint64_t)n != in) { if (report && tid == 0) ctr[2] = 1ull; return w; }   // (the buffer does not hold what the counters say)
    }
    // block scan of the 4096 counts (thread t owns buckets 4 t .. 4 t + 3)
    unsigned long long cq[4], mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { cq[q] = hist[4 * tid + q]; mine += cq[q]; }
    unsigned long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wsum[wave] = incl;
    if (tid == 0) s_pick[2] = ~0ull;
    __syncthreads();
    unsigned long long before = 0;
    for (int v = 0; v < wave; ++v) before += s_wsum[v];
    incl += before;
    {
        const unsigned long long excl = incl - mine;
        if (w.rank >= excl && w.rank < incl) {   // (exactly one thread for a consistent histogram)
            unsigned long long cum = excl;
            int q = 0;
            if (cum + cq[0] <= w.rank) { cum += cq[0]; q = 1;
                if (cum + cq[1] <= w.rank) { cum += cq[1]; q = 2;
                    if (cum + cq[2] <= w.rank) { cum += cq[2]; q = 3; } } }
            s_pick[0] = (unsigned long long)(4 * tid + q);
            s_pick[1] = cum;
            s_pick[2] = q == 0 ? cq[0] : (q == 1 ? cq[1] : (q == 2 ? cq[2] : cq[3]));
        }
    }
    __syncthreads();
    if (s_pick[2] == ~0ull) {   // no bucket found (inconsistent histogram): the plain route
        if (report && tid == 0) ctr[2] = 1ull;
        return w;
    }
    w.bucket = (int)s_pick[0];
    w.below = s_pick[1];
    w.group = s_pick[2];
    w.run = true;
    return w;
}

// keys of the chosen bucket -> small buffer; the smallest key of the buckets above
template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void nk_dhsel_gather_kernel(const T* __restrict__ cd, int64_t cap, const unsigned long long* n_dev,
                                                                       const uint64_t* __restrict__ cnt /* total, below, inside */,
                                                                       const typename KeyT<T>::type* klo, const typename KeyT<T>::type* khi,
                                                                       uint32_t* hdr /* DSEL_HDR_WORDS 64-bit words, then the histogram */,
                                                                       typename KeyT<T>::type* gkeys /* [DSEL_CAP] */, unsigned long long* ctr,
                                                                       int check_n = 1) {
    typedef typename KeyT<T>::type K;
    __shared__ unsigned long long s_pick[4], s_wsum[16];
    __shared__ K s_min;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long m = *n_dev;
    const int64_t n = m < (unsigned long long)cap ? (int64_t)m : cap;
    if (tid == 0) s_min = ~(K)0;
    const DselMap<K> mp = dsel_map<T, K>(*klo, *khi);
    const bool exact = mp.keyspace && mp.shift == 0;
    if (exact) return;    // (a bucket is a key: the histogram says everything -- nk_dhsel_final_kernel)
    const DselWhere w = dsel_locate<T>(cnt, n, hdr + 2 * DSEL_HDR_WORDS, s_pick, s_wsum, ctr, blockIdx.x == 0, false, check_n != 0);
    if (!w.run) return;   // (uniform over the grid; the final kernel hands back NaN)
    // A bucket with more values than the key buffer takes: ties.  (dh of a well-aligned float32 pair is a difference of elevations of
    // ~1e3 m: a multiple of their ulp, 1.2e-4 m -- a dozen distinct values carry millions of candidates.)  Its keys are not gathered;
    // its smallest and largest key are: if they agree the bucket IS that key and the histogram says the rest (nk_dhsel_final_kernel).
    const bool big = w.group > (unsigned long long)DSEL_CAP;
    __shared__ K s_kmin, s_kmax;
    // (the bucket's keys are collected per workgroup first and appended with ONE global atomic: a returning atomic per key on one
    //  address cost ~15 us for the few hundred keys)
    constexpr int LOCAL_KEYS = 512;
    __shared__ K s_keys[LOCAL_KEYS];
    __shared__ uint32_t s_nkeys, s_base;
    if (tid == 0) { s_kmin = ~(K)0; s_kmax = (K)0; s_nkeys = 0u; }
    __syncthreads();
    K bmin = ~(K)0, bmax = (K)0;
    K mn = ~(K)0;
    constexpr int U = 8;
    const int64_t step = (int64_t)blockDim.x * U;
    for (int64_t base = (int64_t)blockIdx.x * step; base < n; base += (int64_t)gridDim.x * step) {
        T x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + (int64_t)u * blockDim.x + tid;
            x[u] = i < n ? cd[i] : (T)NAN;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (x[u] != x[u]) continue;
            const K key = key_of(x[u]);
            const int d = dsel_bucket<T, K>(x[u], mp);
            if (d == w.bucket) {
                if (big) {
                    bmin = key < bmin ? key : bmin;
                    bmax = key > bmax ? key : bmax;
                } else {
                    const uint32_t lp = atomicAdd(&s_nkeys, 1u);
                    if (lp < (uint32_t)LOCAL_KEYS) s_keys[lp] = key;
                    else {   // (more than the local buffer takes: straight to the global one)
                        const uint32_t pos = atomicAdd(&hdr[2], 1u);
                        if (pos < (uint32_t)DSEL_CAP) gkeys[pos] = key;
                    }
                }
            } else if (d > w.bucket && key < mn) {
                mn = key;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const K t = k_shfl_down(mn, o);
        mn = t < mn ? t : mn;
        const K t1 = k_shfl_down(bmin, o), t2 = k_shfl_down(bmax, o);
        bmin = t1 < bmin ? t1 : bmin;
        bmax = t2 > bmax ? t2 : bmax;
    }
    __syncthreads();
    {
        const uint32_t nk = s_nkeys < (uint32_t)LOCAL_KEYS ? s_nkeys : (uint32_t)LOCAL_KEYS;
        if (tid == 0 && nk) s_base = atomicAdd(&hdr[2], nk);
        __syncthreads();
        for (uint32_t i = tid; i < nk; i += blockDim.x)
            if (s_base + i < (uint32_t)DSEL_CAP) gkeys[s_base + i] = s_keys[i];
    }
    if (lane == 0 && mn != ~(K)0) k_atomic_min(&s_min, mn);   // (one global atomic per workgroup: they all land on one address)
    if (lane == 0 && big && bmin != ~(K)0) { k_atomic_min(&s_kmin, bmin); k_atomic_max(&s_kmax, bmax); }
    __syncthreads();
    unsigned long long* hdr64 = reinterpret_cast<unsigned long long*>(hdr);
    if (tid == 0 && s_min != ~(K)0) k_atomic_max(reinterpret_cast<K*>(hdr64 + 2), (K)~s_min);
    if (tid == 0 && big && s_kmin != ~(K)0) {
        k_atomic_max(reinterpret_cast<K*>(hdr64 + 0), (K)~s_kmin);
        k_atomic_max(reinterpret_cast<K*>(hdr64 + 3), s_kmax);
    }
}

// one workgroup: the exact order statistic and its successor among the bucket's keys, vshift
template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void nk_dhsel_final_kernel(int64_t cap, const unsigned long long* n_dev, const uint64_t* __restrict__ cnt,
                                                                      const typename KeyT<T>::type* klo, const typename KeyT<T>::type* khi,
                                                                      const uint32_t* hdr, const typename KeyT<T>::type* gkeys, unsigned long long* ctr,
                                                                      unsigned char* info, int check_n = 1) {
    typedef typename KeyT<T>::type K;
    constexpr int P = KeyT<T>::passes;
    extern __shared__ __attribute__((aligned(16))) unsigned char fz_smem[];
    K* keys = reinterpret_cast<K*>(fz_smem);                                      // [DSEL_CAP]
    uint32_t* h = reinterpret_cast<uint32_t*>(keys + DSEL_CAP);                   // [BINSEL_COPIES][257]
    unsigned long long* s_tot = reinterpret_cast<unsigned long long*>(h + BINSEL_COPIES * (SEL_RADIX + 1) + ((BINSEL_COPIES * (SEL_RADIX + 1)) & 1));   // [256]
    unsigned long long* s_pick = s_tot + SEL_RADIX;                               // bucket / digit, elements below it, elements in it
    unsigned long long* s_wsum = s_pick + 4;                                      // [16] wave totals of the bucket scan
    K* s_min = reinterpret_cast<K*>(s_wsum + 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const uint64_t total = cnt[0], lt = cnt[1], in = cnt[2];
    const unsigned long long m = *n_dev;
    const int64_t n = m < (unsigned long long)cap ? (int64_t)m : cap;
    auto hand_back = [&](T vs) {   // the step's info block: vshift in the DEM dtype and in float64, the valid count, the flags
        *reinterpret_cast<T*>(info) = vs;
        *reinterpret_cast<uint64_t*>(info + 8) = total;
        *reinterpret_cast<uint64_t*>(info + 16) = (uint64_t)((ctr[2] != 0) | ((ctr[3] != 0) << 1));
        *reinterpret_cast<double*>(info + 24) = (double)vs;
    };
    if (tid == 0) *s_min = ~(K)0;
    const DselMap<K> mp = dsel_map<T, K>(*klo, *khi);
    const bool exact = mp.keyspace && mp.shift == 0;
    const DselWhere w = dsel_locate<T>(cnt, n, hdr + 2 * DSEL_HDR_WORDS, s_pick, s_wsum, ctr, true, exact, check_n != 0);
    const unsigned long long* hdr64 = reinterpret_cast<const unsigned long long*>(hdr);
    const bool big = w.run && !exact && w.group > (unsigned long long)DSEL_CAP;
    K one_key = (K)0;
    if (big) {   // more values than the key buffer takes: one key many times over, or the plain route
        const K kmin = (K)~*reinterpret_cast<const K*>(hdr64 + 0), kmax = *reinterpret_cast<const K*>(hdr64 + 3);
        if (kmin != kmax) {
            __syncthreads();
            if (tid == 0) { ctr[2] = 1ull; hand_back((T)NAN); }
            return;
        }
        one_key = kmin;
    }
    if (w.run && (exact || big)) {
        // the bucket is ONE key (key buckets with shift 0: klo + bucket; or smallest = largest key of a bucket of ties): its duplicates = the
        // bucket's count, its successor = the smallest key of the buckets above (the next bucket that holds anything / what the gather noted)
        if (exact) {
            const uint32_t* hist = hdr + 2 * DSEL_HDR_WORDS;
            int nxt = DSEL_BUCKETS;
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                const int d = 4 * tid + q;
                if (d > w.bucket && hist[d] != 0u) nxt = d;
            }
            for (int o = 32; o > 0; o >>= 1) {
                const int t = __shfl_down(nxt, o);
                nxt = t < nxt ? t : nxt;
            }
            if (lane == 0 && nxt < DSEL_BUCKETS) k_atomic_min(s_min, (K)(mp.klo + (K)nxt));
        } else if (tid == 0) {
            *s_min = (K)~*reinterpret_cast<const K*>(hdr64 + 2);   // (zero-initialised: ~0 = none)
        }
        __syncthreads();
        if (tid == 0) {
            const K sel = exact ? (K)(mp.klo + (K)w.bucket) : one_key;
            const uint64_t n_le = w.below + w.group + lt;
            const T lo = val_of(sel);
            T vs = lo;
            if (!(total & 1)) {
                const uint64_t k2 = total / 2;
                T hi = lo;
                if (!(n_le > k2)) hi = val_of(*s_min);
                vs = (T)((T)(lo + hi) / (T)2);
            }
            hand_back(vs);
        }
        return;
    }
    const uint32_t got = hdr[2];
    if (!w.run || (unsigned long long)got != w.group) {   // (the second: cannot happen -- histogram and gather saw the same values)
        __syncthreads();
        if (tid == 0) {
            if (w.run) ctr[2] = 1ull;
            hand_back((T)NAN);
        }
        return;
    }
    const int g = (int)got;
    for (int i = tid; i < g; i += blockDim.x) keys[i] = gkeys[i];
    const K above = (K)~*reinterpret_cast<const K*>(reinterpret_cast<const unsigned long long*>(hdr) + 2);   // (zero-initialised: ~0 = none)
    SelState<K> s;
    s.prefix = 0; s.rank = w.rank - w.below; s.count = in; s.n_le = w.below; s.group = w.group;
    uint32_t* hc = h + (tid & (BINSEL_COPIES - 1)) * (SEL_RADIX + 1);
    auto pick = [&](auto count_of) {   // every thread of the FIRST wave calls it; the result lands in s_pick
        unsigned long long c4[4], mine4 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c4[q] = count_of(4 * lane + q); mine4 += c4[q]; }
        unsigned long long inc = mine4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        const unsigned long long exc = inc - mine4;
        if (s.rank >= exc && s.rank < inc) {
            unsigned long long cum = exc;
            int q = 0;
            if (cum + c4[0] <= s.rank) { cum += c4[0]; q = 1;
                if (cum + c4[1] <= s.rank) { cum += c4[1]; q = 2;
                    if (cum + c4[2] <= s.rank) { cum += c4[2]; q = 3; } } }
            s_pick[0] = (unsigned long long)(4 * lane + q);
            s_pick[1] = cum;
            s_pick[2] = q == 0 ? c4[0] : (q == 1 ? c4[1] : (q == 2 ? c4[2] : c4[3]));
        }
    };
    __syncthreads();
    for (int p = 0; p < P; ++p) {   // exact selection among the bucket's keys
        const int shift = 8 * (P - 1 - p);
        for (int k = tid; k < BINSEL_COPIES * (SEL_RADIX + 1); k += blockDim.x) h[k] = 0u;
        __syncthreads();
        const K himask = p == 0 ? (K)0 : (K)(~(K)0 << (shift + 8));
        for (int i = tid; i < g; i += blockDim.x) {
            const K key = keys[i];
            if ((key & himask) == s.prefix) atomicAdd(&hc[(int)((key >> shift) & 0xFF)], 1u);
        }
        __syncthreads();
        if (tid < SEL_RADIX) {
            unsigned long long t = 0;
#pragma unroll
            for (int q = 0; q < BINSEL_COPIES; ++q) t += h[q * (SEL_RADIX + 1) + tid];
            s_tot[tid] = t;
        }
        __syncthreads();
        if (tid < 64) pick([&](int d) { return s_tot[d]; });
        __syncthreads();
        s.prefix |= (K)s_pick[0] << shift;
        s.n_le += s_pick[1];
        s.rank -= s_pick[1];
        s.group = s_pick[2];
        __syncthreads();
    }
    s.n_le += s.group;   // every digit fixed: group = the selected key's duplicates
    K mn2 = above;
    for (int i = tid; i < g; i += blockDim.x) {
        const K key = keys[i];
        if (key > s.prefix && key < mn2) mn2 = key;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const K t = k_shfl_down(mn2, o);
        mn2 = t < mn2 ? t : mn2;
    }
    if (lane == 0 && mn2 != ~(K)0) k_atomic_min(s_min, mn2);
    __syncthreads();
    if (tid == 0) {   // (np.nanmedian's arithmetic: the middle value, or the mean of the two middle values formed in the DEM dtype)
        const uint64_t n_le = s.n_le + lt;
        const T lo = val_of(s.prefix);
        T vs = lo;
        if (!(total & 1)) {
            const uint64_t k2 = total / 2;
            T hi = lo;
            if (!(n_le > k2)) hi = val_of(*s_min);
            vs = (T)((T)(lo + hi) / (T)2);
        }
        hand_back(vs);
    }
}

// ---- round 5 (second half): the ONE-PASS step on PARTITIONED plans ---------------------------------------------------------------
// With a reduction hook (row blocks over ranks, one process per GPU) the step used to take a two-pass route (rounds 3-5; retired in round 6): ~25 small all-reduces and
// two data passes.  The one-pass step needs, besides its data pass over THIS rank's rows, global counters and global order statistics
// among candidates that are spread over the ranks.  Everything that crosses ranks is an all-reduce of 8-byte words through the hook
// (sums), TEN per step:
//   1-3   the dh sample's dual bracket selection: one histogram per digit        (select_enqueue); the first
//         of them also carries min / max aspect and the survivors of every rank's EXT lists, each rank in ITS OWN slot (all other
//         ranks add zeros there): a sum all-reduce used as an all-gather, folded by every rank afterwards
//   4-6   the y^ sample's dual bracket selection of the 72 bins
//   7     the pass's counters (cnt_d[3], cls_y[3][nb]) + every rank's five float64 sums in per-rank slots (added in rank order on
//         every rank: the same bits everywhere, whatever the reduction tree) + the 4096-bucket histogram of the dh candidates, ONE ROW PER
//         RANK (same trick): every rank then knows the global histogram -- hence the bucket that holds the wanted rank -- and how
//         many of that bucket's keys each rank holds, i.e. where its own keys go in the bucket's global key list
//   8     that key list (each rank writes its keys at its offset, zeros elsewhere) + the gather's header words per rank; then the
//         single-GPU final kernel runs on it unchanged, on every rank: same vshift everywhere
//   9     per bin: the 256-value-bucket histogram of this rank's segment of kept y, one row per rank, + the resolved counters
//   10    per bin: the chosen bucket's values of all ranks + the smallest value above the bucket of every rank that has one, at
//         offsets known from 9, in a fixed stride of MR_GSEG values per bin; the single-GPU `nk_bin_select_kernel` then runs on that
//         small array with counters rewritten so that the wanted rank, the count below and the successor rule come out as on the
//         whole set (elements below the bucket are counted into "below", nothing above the first value above the bucket matters)
// All integers, the same keys, the same selection code: medians, counts, vshift are the single-GPU fit's bit for bit (GPU test:
// 2 ranks == 1 process).  Failure flags that only one rank can see (a buffer overflow) travel with 9; flags raised later derive
// from reduced data and are identical on every rank, so all ranks fall through to the plain route together or not at all.
constexpr int MR_WORLD_MAX = 16;
constexpr int MR_GSEG = 2048;   // values per bin in exchange 10 (the chosen bucket of a bin holds a few hundred)

// exchange 1 (rides on the first histogram all-reduce of the dh sample's selection): min / max aspect key and survivors of this
// rank's EXT lists in its slot of [world][4], zeros in the others'; afterwards the fold over the slots
static __global__ void nk_mr_ext_pack_kernel(const DhStats* s, const unsigned long long* surv, int rank, int world, uint64_t* slots) {
    const int k = threadIdx.x;
    if (blockIdx.x != 0 || k >= 4 * world) return;
    uint64_t v = 0;
    if (k / 4 == rank) v = (k & 3) == 0 ? s->asp_min : ((k & 3) == 1 ? s->asp_max : (uint64_t)surv[(k & 3) - 2]);
    slots[k] = v;
}
static __global__ void nk_mr_ext_unpack_kernel(const uint64_t* slots, int world, DhStats* s, unsigned long long* surv) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t mn = ~(uint64_t)0, mx = 0, s0 = 0, s1 = 0;
    for (int r = 0; r < world; ++r) {
        mn = slots[4 * r] < mn ? slots[4 * r] : mn;
        mx = slots[4 * r + 1] > mx ? slots[4 * r + 1] : mx;
        s0 += slots[4 * r + 2];
        s1 += slots[4 * r + 3];
    }
    s->asp_min = mn; s->asp_max = mx; surv[0] = s0; surv[1] = s1;
}
// exchange 7: [0, 3) cnt_d | [3, 3 + 3 nb) cls_y | [world][5] float64 sums, slot of this rank only | (then the histogram rows)
static __global__ __launch_bounds__(256) void nk_mr_counts_pack_kernel(const uint64_t* cnt_d, const uint64_t* cls_y, const double* wg_sums, int n_wg, int nb, int rank,
                                                                        int world, uint64_t* red, uint64_t* cls_loc) {
    static_assert(NK_SUMS_THREADS == 256, "this kernel's block is the reduction's");
    __shared__ double s_part[NK_SUMS_THREADS][5];
    __shared__ double sums[5];
    nk_sums_reduce(wg_sums, n_wg, s_part, sums);   // (this rank's sums of the pass, fixed order)
    __syncthreads();
    const int nsum = 3 + 3 * nb, total = nsum + 5 * world;
    for (int k = threadIdx.x; k < total; k += blockDim.x) {
        uint64_t v = 0;
        if (k < 3) v = cnt_d[k];
        else if (k < nsum) { v = cls_y[k - 3]; cls_loc[k - 3] = v; }
        else if ((k - nsum) / 5 == rank) v = (uint64_t)__double_as_longlong(sums[(k - nsum) % 5]);
        red[k] = v;
    }
}
static __global__ __launch_bounds__(256) void nk_mr_counts_unpack_kernel(const uint64_t* red, int nb, int world, uint64_t* cnt_d, uint64_t* cls_y, double* sums) {
    const int nsum = 3 + 3 * nb;
    for (int k = threadIdx.x; k < nsum; k += blockDim.x) {
        if (k < 3) cnt_d[k] = red[k];
        else cls_y[k - 3] = red[k];
    }
    if (threadIdx.x < 5) {
        double a = 0.0;
        for (int r = 0; r < world; ++r) a += __longlong_as_double((long long)red[nsum + 5 * r + threadIdx.x]);   // (rank order: the same bits on every rank)
        sums[threadIdx.x] = a;
    }
}
// after exchange 7: the global histogram (sum of the rows) into the selection's own place, the bucket of the wanted rank, and where
// this rank's keys of that bucket go in the global key list: hdr[2] (the gather's append counter) starts there, hdr[3] remembers it
template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void nk_mr_dh_base_kernel(const uint32_t* __restrict__ rows /* [world][DSEL_BUCKETS] */, int world, int rank,
                                                                     const uint64_t* __restrict__ cnt, const typename KeyT<T>::type* klo,
                                                                     const typename KeyT<T>::type* khi, uint32_t* hdr, unsigned long long* ctr) {
    typedef typename KeyT<T>::type K;
    __shared__ unsigned long long s_pick[4], s_wsum[16];
    uint32_t* hist = hdr + 2 * DSEL_HDR_WORDS;
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int d = 4 * tid + q;
        uint32_t t = 0;
        for (int r = 0; r < world; ++r) t += rows[(size_t)r * DSEL_BUCKETS + d];
        hist[d] = t;   // (read back below by the thread that wrote it)
    }
    const DselMap<K> mp = dsel_map<T, K>(*klo, *khi);
    if (mp.keyspace && mp.shift == 0) return;   // a bucket is a key: nothing is gathered
    const DselWhere w = dsel_locate<T>(cnt, 0, hist, s_pick, s_wsum, ctr, false, false, false);
    if (!w.run) return;
    if (tid == 0) {
        uint32_t base = 0;
        for (int r = 0; r < rank; ++r) base += rows[(size_t)r * DSEL_BUCKETS + w.bucket];
        hdr[2] = base;
        hdr[3] = base;
    }
}
// exchange 8, before: this rank's header words into its slot; after: the headers of all ranks folded (maxima; keys appended = sum of
// what every rank appended behind its base)
static __global__ void nk_mr_dh_hdr_pack_kernel(const uint32_t* hdr, int rank, uint64_t* slots /* [world][DSEL_HDR_WORDS] */) {
    if (threadIdx.x < DSEL_HDR_WORDS && blockIdx.x == 0)
        slots[(size_t)rank * DSEL_HDR_WORDS + threadIdx.x] = reinterpret_cast<const uint64_t*>(hdr)[threadIdx.x];
}
static __global__ void nk_mr_dh_hdr_merge_kernel(const uint64_t* slots, int world, uint32_t* hdr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t m0 = 0, m2 = 0, m3 = 0, got = 0;
    for (int r = 0; r < world; ++r) {
        const uint64_t* sl = slots + (size_t)r * DSEL_HDR_WORDS;
        m0 = sl[0] > m0 ? sl[0] : m0;
        m2 = sl[2] > m2 ? sl[2] : m2;
        m3 = sl[3] > m3 ? sl[3] : m3;
        got += (uint64_t)(uint32_t)sl[1] - (uint64_t)(uint32_t)(sl[1] >> 32);
    }
    uint64_t* h64 = reinterpret_cast<uint64_t*>(hdr);
    h64[0] = m0; h64[1] = got; h64[2] = m2; h64[3] = m3;
}
// exchange 9, before: one workgroup per bin histograms THIS rank's segment of kept y into the 256 value buckets of the bin's bracket
// (nk_bin_select_kernel's map), its row of red; the resolved counters of the rank; flags only this rank may know
// red: [0] overflow, [1] miss | res [2][nb] | rows [world][nb][256] (uint32)
template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void nk_mr_bin_hist_kernel(const T* __restrict__ seg_v, const uint64_t* __restrict__ cls_loc /* [3][nb] */,
                                                                      const uint64_t* __restrict__ res /* [2][nb], this rank */, const unsigned long long* seg_ctr,
                                                                      int nb, const typename KeyT<T>::type* __restrict__ klo, const typename KeyT<T>::type* __restrict__ khi,
                                                                      int rank, int world, uint64_t* red, const unsigned long long* ctr) {
    // (one workgroup per bin, like nk_bin_select_kernel's first read: 1024 threads, eight values per thread and trip with all loads
    //  issued first, eight copies of the table -- 256 threads with one value in flight took 190 us for the 72 segments of C3)
    typedef typename KeyT<T>::type K;
    __shared__ uint32_t h[BINSEL_COPIES * (SEL_RADIX + 1)];
    __shared__ unsigned long long s_off;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < BINSEL_COPIES * (SEL_RADIX + 1); k += blockDim.x) h[k] = 0u;
    const unsigned long long n = seg_ctr[(size_t)b * BINSEG_CTR_STRIDE];
    if (tid == 0) {
        unsigned long long acc = 0;
        for (int k = 0; k < b; ++k) acc += cls_loc[2 * nb + k];
        s_off = acc;
        red[2 + b] = res[b];
        red[2 + nb + b] = res[nb + b];
        if (n != res[nb + b]) atomicAdd(reinterpret_cast<unsigned long long*>(&red[0]), 1ull);   // (the segment does not hold what the counters say)
        if (b == 0) {
            if (ctr[2] != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&red[0]), 1ull);
            if (ctr[3] != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&red[1]), 1ull);
        }
    }
    __syncthreads();
    const T* v = seg_v + s_off;
    const BinsegMap mp = binseg_map<T, K>(klo[b], khi[b]);
    uint32_t* hc = h + (tid & (BINSEL_COPIES - 1)) * (SEL_RADIX + 1);
    constexpr int U = 8;
    for (unsigned long long i0 = tid; i0 < n; i0 += (unsigned long long)blockDim.x * U) {
        T x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long i = i0 + (unsigned long long)u * blockDim.x;
            x[u] = i < n ? v[i] : (T)NAN;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (x[u] == x[u]) atomicAdd(&hc[binseg_bucket<T>(x[u], mp)], 1u);
    }
    __syncthreads();
    if (tid < SEL_RADIX) {
        uint32_t t = 0;
#pragma unroll
        for (int q = 0; q < BINSEL_COPIES; ++q) t += h[q * (SEL_RADIX + 1) + tid];
        uint32_t* rows = reinterpret_cast<uint32_t*>(red + 2 + 2 * nb);
        rows[((size_t)rank * nb + b) * SEL_RADIX + tid] = t;
    }
}
// exchange 10, before: per bin the bucket that holds the wanted rank (from the summed rows), this rank's values of that bucket and its
// smallest value above it into the bin's stride of `gseg` at the offsets the rows give; the counters nk_bin_select_kernel will read,
// rewritten for the small array (see the block comment); the true (total, below, inside) for the host's bracket statistics
template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void nk_mr_bin_gather_kernel(const T* __restrict__ seg_v, const uint64_t* __restrict__ cls_loc, const uint64_t* __restrict__ cls /* summed */,
                                                               const uint64_t* __restrict__ red /* exchange 9, summed */, const unsigned long long* seg_ctr, int nb,
                                                               const typename KeyT<T>::type* __restrict__ klo, const typename KeyT<T>::type* __restrict__ khi,
                                                               int rank, int world, T* __restrict__ gseg /* [nb][MR_GSEG], zeroed */, uint64_t* cls_f /* [3][nb] */,
                                                               uint64_t* res_f /* [2][nb] */, unsigned long long* segf_ctr, uint64_t* cnt_true /* [3][nb] */,
                                                               unsigned long long* ctr) {
    typedef typename KeyT<T>::type K;
    __shared__ unsigned long long s_g[SEL_RADIX];
    __shared__ unsigned long long s_pick[3];
    __shared__ uint32_t s_mloc[MR_WORLD_MAX], s_above[MR_WORLD_MAX];
    __shared__ uint32_t s_cnt;
    __shared__ K s_min;
    __shared__ unsigned long long s_off;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const uint64_t* res_g = red + 2;
    const uint32_t* rows = reinterpret_cast<const uint32_t*>(red + 2 + 2 * nb);
    if (b == 0 && tid == 0) {   // what some rank flagged before the exchange holds for every rank
        if (red[0] != 0) ctr[2] = 1ull;
        if (red[1] != 0) ctr[3] = 1ull;
    }
    const uint64_t total = cls[b] + cls[nb + b] + cls[2 * nb + b], lt = cls[nb + b] + res_g[b], in = res_g[nb + b];
    auto hand_over = [&](uint64_t below, uint64_t g) {   // (thread 0) the counters of the small array: total kept, `below` certainly below it
        cls_f[b] = g ? total - below - g : 0; cls_f[nb + b] = g ? below : 0; cls_f[2 * nb + b] = g;
        res_f[b] = 0; res_f[nb + b] = g;
        segf_ctr[(size_t)b * BINSEG_CTR_STRIDE] = g;
        cnt_true[b] = total; cnt_true[nb + b] = lt; cnt_true[2 * nb + b] = in;
    };
    bool run = total != 0;
    uint64_t rk = 0;
    if (run) {   // (the rule of bracket_given_kernel, as in nk_bin_select_kernel)
        const uint64_t k = (total - 1) / 2, need = (total & 1) ? k : k + 1;
        if (lt > k || need - lt >= in) { run = false; if (tid == 0) ctr[3] = 1ull; }
        else rk = k - lt;
    }
    if (tid == 0) {
        s_cnt = 0u; s_min = ~(K)0; s_pick[2] = ~0ull;
        unsigned long long acc = 0;
        for (int k = 0; k < b; ++k) acc += cls_loc[2 * nb + k];
        s_off = acc;
    }
    if (!run) {   // (uniform) an empty bin, or a bracket that missed: the selection kernel sees an empty bin
        if (tid == 0) hand_over(0, 0);
        return;
    }
    if (tid < SEL_RADIX) {
        unsigned long long t = 0;
        for (int r = 0; r < world; ++r) t += rows[((size_t)r * nb + b) * SEL_RADIX + tid];
        s_g[tid] = t;
    }
    __syncthreads();
    if (tid < 64) {   // one wave: lane l owns buckets 4 l .. 4 l + 3 (nk_bin_select_kernel's pick)
        unsigned long long cq[4], mine = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { cq[q] = s_g[4 * lane + q]; mine += cq[q]; }
        unsigned long long incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const unsigned long long excl = incl - mine;
        if (rk >= excl && rk < incl) {
            unsigned long long cum = excl;
            int q = 0;
            if (cum + cq[0] <= rk) { cum += cq[0]; q = 1;
                if (cum + cq[1] <= rk) { cum += cq[1]; q = 2;
                    if (cum + cq[2] <= rk) { cum += cq[2]; q = 3; } } }
            s_pick[0] = (unsigned long long)(4 * lane + q);
            s_pick[1] = cum;
            s_pick[2] = q == 0 ? cq[0] : (q == 1 ? cq[1] : (q == 2 ? cq[2] : cq[3]));
        }
    }
    __syncthreads();
    if (s_pick[2] == ~0ull) {   // (inconsistent rows: cannot happen -- the same on every rank, they are summed data)
        if (tid == 0) { ctr[2] = 1ull; hand_over(0, 0); }
        return;
    }
    const int d1 = (int)s_pick[0];
    const unsigned long long below_d1 = s_pick[1], m = s_pick[2];
    if (tid < world) {
        const uint32_t* row = rows + ((size_t)tid * nb + b) * SEL_RADIX;
        uint32_t above = 0;
        for (int d = d1 + 1; d < SEL_RADIX; ++d) above |= row[d];
        s_mloc[tid] = row[d1];
        s_above[tid] = above ? 1u : 0u;
    }
    __syncthreads();
    unsigned long long g = m, base = 0;
    for (int r = 0; r < world; ++r) {
        g += s_above[r];
        if (r < rank) base += (unsigned long long)s_mloc[r] + s_above[r];
    }
    if (g > (unsigned long long)MR_GSEG) {   // (uniform, and the same on every rank)
        if (tid == 0) { ctr[2] = 1ull; hand_over(0, 0); }
        return;
    }
    const T* v = seg_v + s_off;
    const unsigned long long n = seg_ctr[(size_t)b * BINSEG_CTR_STRIDE];
    const BinsegMap mp = binseg_map<T, K>(klo[b], khi[b]);
    T* out = gseg + (size_t)b * MR_GSEG + base;
    const uint32_t mine = s_mloc[rank];
    K mn = ~(K)0;
    constexpr int U = 8;   // (1024 threads, eight values per thread and trip, all loads first: nk_bin_select_kernel's second read)
    for (unsigned long long i0 = tid; i0 < n; i0 += (unsigned long long)blockDim.x * U) {
        T x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned long long i = i0 + (unsigned long long)u * blockDim.x;
            x[u] = i < n ? v[i] : (T)NAN;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (x[u] != x[u]) continue;
            const int d = binseg_bucket<T>(x[u], mp);
            if (d == d1) {
                const uint32_t pos = atomicAdd(&s_cnt, 1u);
                if (pos < mine) out[pos] = x[u];
            } else if (d > d1) {
                const K key = key_of(x[u]);
                mn = key < mn ? key : mn;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const K t = k_shfl_down(mn, o);
        mn = t < mn ? t : mn;
    }
    if (lane == 0 && mn != ~(K)0) k_atomic_min(&s_min, mn);
    __syncthreads();
    if (tid == 0) {
        if (s_cnt != mine) ctr[2] = 1ull;   // (the segment changed between the two reads: cannot happen)
        if (s_above[rank]) {
            if (s_min != ~(K)0) out[mine] = val_of(s_min);
            else ctr[2] = 1ull;
        }
        hand_over(lt + below_d1, g);
    }
}

// every small result of a step gathered into one block (one device-to-host copy instead of ten)
struct FzPack { const unsigned char* src[16]; uint32_t bytes[16]; uint32_t off[16]; int n; unsigned char* dst; };
static __global__ __launch_bounds__(256) void nk_fz_pack_kernel(FzPack a) {
    for (int k = 0; k < a.n; ++k)
        for (uint32_t i = threadIdx.x; i < a.bytes[k]; i += blockDim.x) a.dst[a.off[k] + i] = a.src[k][i];
}

}  // namespace xd
