// select_run.h -- device passes and host driver of the exact per-bin selection (select.h), shared by the Nuth-Kaab
// step (nuthkaab.hip) and the N-D binned statistics (binstats.hip): LDS-privatised digit histograms over
// (values, bin ids), the successor pass for even counts, and run_select() which sequences them (all-reducing the
// integer tables through the context hook when the data are sharded over GPUs).
#pragma once
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "select.h"

namespace xd {

// ---- dtype-exact arithmetic ---------------------------------------------------------------------------------
// Plain operators under `fp contract(off)`: hipcc then emits IEEE-correctly-rounded add / mul / div / sqrt (its
// default -fhip-fp32-correctly-rounded-divide-sqrt) and never fuses a*b+c.  (The __f*_rn intrinsics are NOT strict
// in HIP: __fsqrt_rn is the 1-ulp native square root and __fmul_rn / __fadd_rn may be contracted.)
#pragma clang fp contract(off)
template <typename T> __device__ __forceinline__ T t_sub(T a, T b) { return a - b; }
template <typename T> __device__ __forceinline__ T t_add(T a, T b) { return a + b; }
template <typename T> __device__ __forceinline__ T t_mul(T a, T b) { return a * b; }
template <typename T> __device__ __forceinline__ T t_div(T a, T b) { return a / b; }
__device__ __forceinline__ float t_sqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ double t_sqrt(double a) { return sqrt(a); }
template <typename T> __device__ __forceinline__ bool t_finite(T v) { return fabs((double)v) <= 1.79769313486231570e308 && v == v; }
template <> __device__ __forceinline__ bool t_finite<float>(float v) { return fabsf(v) <= 3.402823466e38f; }

// 64-bit keys are `unsigned long` on Linux; HIP's atomics / shuffles want `unsigned long long`
__device__ __forceinline__ void k_atomic_min(uint32_t* p, uint32_t v) { atomicMin(p, v); }
__device__ __forceinline__ void k_atomic_max(uint32_t* p, uint32_t v) { atomicMax(p, v); }
__device__ __forceinline__ void k_atomic_min(uint64_t* p, uint64_t v) { atomicMin(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); }
__device__ __forceinline__ void k_atomic_max(uint64_t* p, uint64_t v) { atomicMax(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); }
__device__ __forceinline__ uint32_t k_shfl_down(uint32_t v, int off) { return __shfl_down(v, off); }
__device__ __forceinline__ uint64_t k_shfl_down(uint64_t v, int off) { return (uint64_t)__shfl_down((unsigned long long)v, off); }

// ---- generic histogram / successor passes over (values, bin ids) ----------------------------------------------
// bins == nullptr: single bin (global median).  LDS: `copies` privatised tables of nb * 256 uint32 counters
// (copy = lane % copies): with few bins every lane of a wave would otherwise hit the same counter of the
// low-entropy leading digit and serialise 64-way.  1024-thread workgroups so that even the 72 KB table of the
// 72 aspect bins runs at full occupancy (2 workgroups = 32 waves per CU).
constexpr int HIST_THREADS = 1024;

template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void hist_pass_kernel(const T* __restrict__ vals, const uint16_t* __restrict__ bins,
                                                                 int64_t n, int nb, int bin0, int copies,
                                                                 const SelState<typename KeyT<T>::type>* st, int shift, int first,
                                                                 uint64_t* hist) {
    typedef typename KeyT<T>::type K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* h = reinterpret_cast<uint32_t*>(smem);
    const int table = nb * SEL_RADIX;
    for (int k = threadIdx.x; k < table * copies; k += blockDim.x) h[k] = 0;
    __syncthreads();
    uint32_t* hc = h + (threadIdx.x % copies) * table;
    const K himask = first ? (K)0 : (K)(~(K)0 << (shift + 8));
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const T v = vals[p];
        if (v != v) continue;
        int b = bins ? (int)bins[p] - bin0 : 0;
        if (b < 0 || b >= nb) continue;
        const K key = key_of(v);
        if (!first && (key & himask) != st[bin0 + b].prefix) continue;
        atomicAdd(&hc[b * SEL_RADIX + (int)((key >> shift) & 0xFF)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < table; k += blockDim.x) {
        unsigned long long c = 0;
        for (int q = 0; q < copies; ++q) c += h[q * table + k];
        if (c) atomicAdd(reinterpret_cast<unsigned long long*>(&hist[(size_t)bin0 * SEL_RADIX + k]), c);
    }
}

template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void succ_pass_kernel(const T* __restrict__ vals, const uint16_t* __restrict__ bins,
                                                                 int64_t n, int nb, const SelState<typename KeyT<T>::type>* st,
                                                                 uint64_t* succ /* [nb], all-ones = none */) {
    typedef typename KeyT<T>::type K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    K* m = reinterpret_cast<K*>(smem);
    K* pref = m + nb;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) { m[k] = ~(K)0; pref[k] = st[k].prefix; }
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const T v = vals[p];
        if (v != v) continue;
        const int b = bins ? (int)bins[p] : 0;
        if (b < 0 || b >= nb) continue;
        const K key = key_of(v);
        if (key > pref[b] && key < m[b]) k_atomic_min(&m[b], key);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nb; k += blockDim.x)
        if (m[k] != ~(K)0) k_atomic_min(&succ[k], (uint64_t)m[k]);
}


constexpr int MAX_BINS_PER_SWEEP = 128;  // 128 * 256 * 4 B = 128 KiB of LDS histograms per workgroup

template <typename F> int set_big_lds(xdemhip_ctx* ctx, F func, size_t bytes) {
    if (bytes > 48 * 1024) XD_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(func), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return XDEMHIP_OK;
}

inline int grid_for(const xdemhip_ctx* ctx, int64_t n, int block, int per_cu) {
    int64_t g = (n + block - 1) / block;
    const int64_t cap = (int64_t)ctx->num_cu * per_cu;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}


// scratch layout (bytes): [0, 16384) bin edges | stats | sums | selection states | successor keys | histograms
constexpr size_t OFF_STATS = 16384, OFF_SUMS = OFF_STATS + 64, OFF_STATE = OFF_STATS + 128;
inline int nb1(int nb) { return nb > 1 ? nb : 1; }
inline size_t off_succ(int nb) { return OFF_STATE + (size_t)nb1(nb) * 64; }
inline size_t off_hist(int nb) { return off_succ(nb) + (size_t)nb1(nb) * 8; }
inline size_t scratch_size(int nb) { return off_hist(nb) + (size_t)nb1(nb) * SEL_RADIX * 8 + 256; }

template <typename K> struct SelResult {
    SelState<K> st;
    uint64_t succ;  // smallest key above the selected one, all-ones if none
};

// Exact lower/upper medians of vals[0..n) per bin (bins == nullptr: one bin).  With an all-reduce hook installed the
// integer histograms / successor keys are combined over the ranks after every pass, so every rank selects the same
// global order statistics from its own share of the data.
template <typename T>
inline int run_select(xdemhip_ctx* ctx, const T* vals, const uint16_t* bins, int64_t n, int nb, unsigned char* scratch,
               std::vector<SelResult<typename KeyT<T>::type>>& out) {
    typedef typename KeyT<T>::type K;
    SelState<K>* st = reinterpret_cast<SelState<K>*>(scratch + OFF_STATE);
    uint64_t* d_succ = reinterpret_cast<uint64_t*>(scratch + off_succ(nb));
    uint64_t* d_hist = reinterpret_cast<uint64_t*>(scratch + off_hist(nb));
    XD_HIP_CHECK(ctx, hipMemsetAsync(st, 0, sizeof(SelState<K>) * nb, ctx->stream));
    XD_HIP_CHECK(ctx, hipMemsetAsync(d_succ, 0xFF, 8 * (size_t)nb, ctx->stream));
    XD_HIP_CHECK(ctx, hipMemsetAsync(d_hist, 0, sizeof(uint64_t) * (size_t)nb * SEL_RADIX, ctx->stream));
    const int passes = KeyT<T>::passes;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * (passes - 1 - p);
        if (n > 0)
            for (int b0 = 0; b0 < nb; b0 += MAX_BINS_PER_SWEEP) {
                const int nbs = (nb - b0) < MAX_BINS_PER_SWEEP ? (nb - b0) : MAX_BINS_PER_SWEEP;
                // privatise the table as often as fits in ~64 KB (32 copies for the single-bin global median)
                int copies = (64 * 1024) / (nbs * SEL_RADIX * (int)sizeof(uint32_t));
                copies = copies < 1 ? 1 : (copies > 32 ? 32 : copies);
                const size_t lds = (size_t)nbs * SEL_RADIX * sizeof(uint32_t) * copies;
                int rc = set_big_lds(ctx, hist_pass_kernel<T>, lds);
                if (rc) return rc;
                hipLaunchKernelGGL((hist_pass_kernel<T>), dim3(grid_for(ctx, n, HIST_THREADS * 4, 2)), dim3(HIST_THREADS), lds,
                                   ctx->stream, vals, bins, n, nbs, b0, copies, st, shift, (int)(p == 0), d_hist);
                XD_HIP_CHECK(ctx, hipGetLastError());
            }
        int rc = xd_allreduce_device(ctx, d_hist, (int64_t)nb * SEL_RADIX, XDEMHIP_RED_SUM_U64);
        if (rc) return rc;
        hipLaunchKernelGGL((select_advance_kernel<K>), dim3(nb), dim3(64), 0, ctx->stream, st, d_hist, nb, shift,
                           (int)(p == 0), (int)(p == passes - 1));
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    if (n > 0) {
        hipLaunchKernelGGL((succ_pass_kernel<T>), dim3(grid_for(ctx, n, HIST_THREADS * 4, 2)), dim3(HIST_THREADS), 2 * sizeof(K) * nb,
                           ctx->stream, vals, bins, n, nb, st, d_succ);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    int rc = xd_allreduce_device(ctx, d_succ, nb, XDEMHIP_RED_MIN_U64);
    if (rc) return rc;
    std::vector<SelState<K>> hs(nb);
    std::vector<uint64_t> hsucc(nb);
    XD_HIP_CHECK(ctx, hipMemcpyAsync(hs.data(), st, sizeof(SelState<K>) * nb, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipMemcpyAsync(hsucc.data(), d_succ, 8 * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    out.resize(nb);
    for (int k = 0; k < nb; ++k) { out[k].st = hs[k]; out[k].succ = hsucc[k]; }
    return XDEMHIP_OK;
}

// np.nanmedian of a bin from its selection state: odd count -> the middle value; even -> mean of the two middle
// values in the value dtype (np.mean of a 2-element array).
template <typename T> double median_from(const SelResult<typename KeyT<T>::type>& r) {
    typedef typename KeyT<T>::type K;
    const SelState<K>& s = r.st;
    if (s.count == 0) return NAN;
    const T lo = val_of(s.prefix);
    if (s.count & 1) return (double)lo;
    const uint64_t k2 = s.count / 2;  // 0-based rank of the upper median
    const T hi = (s.n_le > k2) ? lo : val_of((K)r.succ);
    return (double)(T)((T)(lo + hi) / (T)2);
}


}  // namespace xd
