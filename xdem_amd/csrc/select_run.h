// select_run.h -- device passes and host driver of the exact per-bin selection (select.h), shared by the Nuth-Kaab
// step (nuthkaab.hip) and the N-D binned statistics (binstats.hip): LDS-privatised digit histograms over
// (values, bin ids), the successor pass for even counts, and run_select() which sequences them (all-reducing the
// integer tables through the context hook when the data are sharded over GPUs).
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <utility>
#include <functional>
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
constexpr int SEL_UNROLL = 4;   // elements per thread and step, all loads issued before the first use (8 measured slower: r04e)

__host__ __device__ inline uint32_t rebase_shift_of(uint32_t range) { return range ? (uint32_t)__builtin_clz(range) : 0u; }
__host__ __device__ inline uint32_t rebase_shift_of(uint64_t range) { return range ? (uint32_t)__builtin_clzll((unsigned long long)range) : 0u; }
// bracket ends of a dual selection (states [0, nb) the low ends, [nb, 2 nb) the high ends) + the rebase shift of the widest
// bracket: one wave (the body of bracket_finish_kernel below)
template <typename K>
__device__ __forceinline__ void bracket_finish_body(const int lane, const SelState<K>* st, int nb, int degenerate, K low_mask, K* klo, K* khi, uint32_t* shift) {
    K r = 0;
    for (int b = lane; b < nb; b += 64) {
        const K lo = st[b].count > 0 ? st[b].prefix : (K)0;
        const K hi = st[nb + b].count > 0 ? (degenerate ? lo : (K)(st[nb + b].prefix | low_mask)) : (K)~(K)0;
        klo[b] = lo;
        khi[b] = hi;
        const K d = hi >= lo ? (K)(hi - lo) : (K)0;
        r = d > r ? d : r;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const K o = k_shfl_down(r, off);
        r = o > r ? o : r;
    }
    if (lane == 0) *shift = rebase_shift_of(r);
}

// hist_pass_kernel<T, FUSE = true> (round 5): the LAST workgroup of the pass to flush its table -- a ticket counter behind a
// device-scope fence -- advances the selection states itself (select_advance_body, one wave per state) and, after the last digit
// of a bracket selection, writes the bracket ends (bracket_finish_body): the Nuth-Kaab step's two sample selections go from
// 3 x (pass + advance) + finish = 7 dependent launches to 3.  Dual bracket selections without a reduction hook only.
// LEGAL ONLY FOR BRACKETS THAT A COUNTING PASS VERIFIES: the hand-over to the last workgroup goes through relaxed device-scope atomics,
// counted waits and a workgroup-scope fence -- no agent-scope release / acquire pair (a device-scope fence writes back and invalidates the
// XCD's whole L2: 38-60 us per pass, measured) -- which the HIP memory model does not order formally; the states it produces are SAMPLE
// brackets whose every rank claim the following exact counting pass checks (a wrong bracket costs a fall-back, never a wrong order
// statistic).  select_enqueue asserts the use: `fuse` requires the dual bracket mode with caller-supplied bracket outputs.
template <typename K> struct HistFuse {
    uint32_t* ticket;      // zero before the pass; the last workgroup puts it back
    SelState<K>* st;       // the states (writable view of `st`)
    int n_states;          // 2 x data bins
    int dual_nb;           // data bins
    int last;              // this is the key's last digit
    uint32_t narrow;
    int finish;            // also write the bracket ends ...
    K low_mask;
    K* klo;
    K* khi;
    uint32_t* rb_shift_out;
};

template <typename T, bool FUSE = false>
__global__ __launch_bounds__(HIST_THREADS) void hist_pass_kernel(const T* __restrict__ vals, const uint16_t* __restrict__ bins,
                                                                 int64_t n, int nb, int bin0, int copies,
                                                                 const SelState<typename KeyT<T>::type>* st, int shift, int first,
                                                                 uint64_t* hist, const unsigned long long* n_dev = nullptr,
                                                                 const typename KeyT<T>::type* rb_lo = nullptr,
                                                                 const uint32_t* rb_shift = nullptr, int dual_total = 0,
                                                                 HistFuse<typename KeyT<T>::type> fa = HistFuse<typename KeyT<T>::type>()) {
    // dual (dual_total = number of data bins of the whole selection, 0 = off): TWO selection states per data bin -- states
    // [0, dual_total) the low ends, [dual_total, 2 dual_total) the high ends of the brackets -- advance in the same passes over
    // the sample; every element is offered to both states of its bin.  `nb` stays the number of DATA bins of this sweep, the
    // LDS table holds 2 nb rows (low-end rows first)
    typedef typename KeyT<T>::type K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* h = reinterpret_cast<uint32_t*>(smem);
    if (n_dev) {  // element count produced on the device (sample / candidate buffers): `n` is then the buffer capacity
        const unsigned long long m = *n_dev;
        n = m < (unsigned long long)n ? (int64_t)m : n;
    }
    if (!first && rb_shift && shift + 8 <= (int)*rb_shift) return;  // all-zero digit of rebased keys: see select_advance_kernel
    const int rows = dual_total ? 2 * nb : nb;
    const int table = rows * SEL_RADIX;
    // privatised copies sit an ODD number of words apart: a stride that is a multiple of 32 would put the same counter of every
    // copy into one LDS bank, and lanes that agree on (bin, digit) -- the common case -- would serialise on it
    const int cstride = copies > 1 ? table + 1 : table;
    for (int k = threadIdx.x; k < cstride * copies; k += blockDim.x) h[k] = 0;
    // per-bin rebase offsets and prefixes: LDS copies (two dependent global loads per element otherwise)
    K* s_lo = reinterpret_cast<K*>(h + (size_t)cstride * copies + (((size_t)cstride * copies) & 1));
    K* s_pref = s_lo + nb;   // [rows]
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        s_lo[k] = rb_lo ? rb_lo[bin0 + k] : (K)0;
        s_pref[k] = st[bin0 + k].prefix;
        if (dual_total) s_pref[nb + k] = st[dual_total + bin0 + k].prefix;
    }
    __syncthreads();
    uint32_t* hc = h + (threadIdx.x % copies) * cstride;
    const K himask = first ? (K)0 : (K)(~(K)0 << (shift + 8));
    // Rebased keys (candidates of a bracketed selection): every candidate of a bin lies in [lo, hi] and shares the leading
    // digits of lo -- all lanes would hammer one LDS counter per bin.  (key - lo[bin]) << s, with s the same for all bins,
    // keeps the order inside a bin and spreads the leading digit.
    const int rbs = rb_shift ? (int)*rb_shift : 0;
    // four elements per thread and step, all loads issued before the first use (memory-level parallelism)
    const int64_t step = (int64_t)blockDim.x * SEL_UNROLL;
    // (the loads of the NEXT step are issued before this step's elements are counted: with one workgroup per CU, or fewer, a thread
    //  walks 6-18 steps and used to sit out a full memory latency in each)
    T vn[SEL_UNROLL];
    uint16_t bn[SEL_UNROLL];
    auto fetch = [&](int64_t base) {
#pragma unroll
        for (int q = 0; q < SEL_UNROLL; ++q) {
            const int64_t p = base + (int64_t)q * blockDim.x + threadIdx.x;
            vn[q] = (p < n) ? vals[p] : (T)NAN;
            bn[q] = (p < n && bins) ? bins[p] : (uint16_t)0;
        }
    };
    fetch((int64_t)blockIdx.x * step);
    for (int64_t base = (int64_t)blockIdx.x * step; base < n; base += (int64_t)gridDim.x * step) {
        T v[SEL_UNROLL];
        uint16_t bb[SEL_UNROLL];
#pragma unroll
        for (int q = 0; q < SEL_UNROLL; ++q) { v[q] = vn[q]; bb[q] = bn[q]; }
        fetch(base + (int64_t)gridDim.x * step);
#pragma unroll
        for (int q = 0; q < SEL_UNROLL; ++q) {
            if (v[q] != v[q]) continue;
            const int b = bins ? (int)bb[q] - bin0 : 0;
            if (b < 0 || b >= nb) continue;
            if (dual_total) {
                const K key = key_of(v[q]);
                const int digit = (int)((key >> shift) & 0xFF);
                // (first digit: one histogram serves both ends -- select_advance_kernel sets both states up from the low end's row)
                if (first || (key & himask) == s_pref[b]) atomicAdd(&hc[b * SEL_RADIX + digit], 1u);
                if (!first && (key & himask) == s_pref[nb + b]) atomicAdd(&hc[(nb + b) * SEL_RADIX + digit], 1u);
                continue;
            }
            K key = key_of(v[q]);
            if (rb_lo) key = (K)((K)(key - s_lo[b]) << rbs);
            if (!first && (key & himask) != s_pref[b]) continue;
            atomicAdd(&hc[b * SEL_RADIX + (int)((key >> shift) & 0xFF)], 1u);
        }
    }
    __syncthreads();
    // (every workgroup finishes its share at about the same time and would walk the table in the same order -- the same few cache
    //  lines of `hist` under all of them at once: each starts at its own row)
    const int flush_n = (first && dual_total) ? nb * SEL_RADIX : table;
    const int flush_0 = (int)(((unsigned)blockIdx.x * 37u) % (unsigned)(flush_n / SEL_RADIX)) * SEL_RADIX;
    for (int kk = threadIdx.x; kk < flush_n; kk += blockDim.x) {
        const int k = kk + flush_0 < flush_n ? kk + flush_0 : kk + flush_0 - flush_n;
        unsigned long long c = 0;
        for (int q = 0; q < copies; ++q) c += h[q * cstride + k];
        // (dual: the high-end rows of this sweep belong to the states behind all the low-end ones)
        const size_t row0 = (dual_total && k >= nb * SEL_RADIX) ? (size_t)(dual_total - nb) + (size_t)bin0 : (size_t)bin0;
        if (c) atomicAdd(reinterpret_cast<unsigned long long*>(&hist[row0 * SEL_RADIX + k]), c);
    }
    if (FUSE) {
        // Hand-over to the last workgroup WITHOUT a device-scope fence (a fence writes back and invalidates the XCD's whole L2: with
        // 256 workgroups that cost 60 us per pass, measured).  Everything that crosses workgroups here is a device-scope atomic --
        // the histogram adds above, the ticket, the last workgroup's loads of the histogram (select_advance_body<K, true>) -- performed
        // at the device's coherence point: a workgroup only has to wait until its own adds have been acknowledged before it takes
        // its ticket.  (Should a bracket ever come out of a stale count, the integer counts of the data pass tell: brackets from
        // samples are proved afterwards, never trusted.)
        __shared__ uint32_t s_last;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) s_last = (atomicAdd(fa.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
        __syncthreads();
        if (s_last == 0u) return;
        if (threadIdx.x == 0) __hip_atomic_store(fa.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int lane = threadIdx.x & 63;
        SelState<K>* lds_st = reinterpret_cast<SelState<K>*>(smem);   // (the table has been flushed: every thread is past the barrier above)
        for (int b = threadIdx.x >> 6; b < fa.n_states; b += HIST_THREADS / 64)
            select_advance_body<K, true>(b, lane, fa.st, hist, fa.n_states, shift, first, fa.last, SEL_BRACKET_DUAL, nullptr, nullptr, PAIR_DEFF_WIDE,
                                         fa.dual_nb, nullptr, nullptr, fa.narrow, fa.finish ? lds_st : nullptr);
        if (fa.finish) {
            __syncthreads();
            if (threadIdx.x < 64) bracket_finish_body<K>(lane, lds_st, fa.dual_nb, 0, fa.low_mask, fa.klo, fa.khi, fa.rb_shift_out);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void succ_pass_kernel(const T* __restrict__ vals, const uint16_t* __restrict__ bins,
                                                                 int64_t n, int nb, const SelState<typename KeyT<T>::type>* st,
                                                                 uint64_t* succ /* [nb], all-ones = none */,
                                                                 const unsigned long long* n_dev = nullptr,
                                                                 const typename KeyT<T>::type* rb_lo = nullptr,
                                                                 const uint32_t* rb_shift = nullptr, const uint32_t* need = nullptr) {
    typedef typename KeyT<T>::type K;
    if (need && *need == 0u) return;   // every bin's successor came out of the last histogram (select_advance_kernel)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    K* m = reinterpret_cast<K*>(smem);
    if (n_dev) {
        const unsigned long long c = *n_dev;
        n = c < (unsigned long long)n ? (int64_t)c : n;
    }
    K* pref = m + nb;
    K* s_lo = pref + nb;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) { m[k] = ~(K)0; pref[k] = st[k].prefix; s_lo[k] = rb_lo ? rb_lo[k] : (K)0; }
    const int rbs = rb_shift ? (int)*rb_shift : 0;
    __syncthreads();
    const int64_t step = (int64_t)blockDim.x * SEL_UNROLL;
    for (int64_t base = (int64_t)blockIdx.x * step; base < n; base += (int64_t)gridDim.x * step) {
        T v[SEL_UNROLL];
        uint16_t bb[SEL_UNROLL];
#pragma unroll
        for (int q = 0; q < SEL_UNROLL; ++q) {
            const int64_t p = base + (int64_t)q * blockDim.x + threadIdx.x;
            v[q] = (p < n) ? vals[p] : (T)NAN;
            bb[q] = (p < n && bins) ? bins[p] : (uint16_t)0;
        }
#pragma unroll
        for (int q = 0; q < SEL_UNROLL; ++q) {
            if (v[q] != v[q]) continue;
            const int b = (int)bb[q];
            if (b >= nb) continue;
            K key = key_of(v[q]);
            if (rb_lo) key = (K)((K)(key - s_lo[b]) << rbs);
            if (key > pref[b] && key < m[b]) k_atomic_min(&m[b], key);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nb; k += blockDim.x)
        if (m[k] != ~(K)0) k_atomic_min(&succ[k], (uint64_t)m[k]);
}


constexpr int MAX_BINS_PER_SWEEP = 128;  // 128 * 256 * 4 B = 128 KiB of LDS histograms per workgroup
constexpr int MAX_ROWS_PER_SWEEP_DUAL = 152;  // dual selections: two rows per data bin; 152 KiB + prefixes < 160 KiB (the 72 aspect bins in one sweep)

// Opt a kernel into more than 48 KiB of dynamic LDS.  hipFuncSetAttribute is a driver call (tens of microseconds, and the
// selection launches this kernel dozens of times per step): repeated only when a launch needs more than was granted before.
inline int set_big_lds_ptr(xdemhip_ctx* ctx, const void* func, size_t bytes) {
    if (bytes <= 48 * 1024) return XDEMHIP_OK;
    struct Entry { const void* f; int dev; size_t bytes; };
    static std::mutex mu;
    static std::vector<Entry> done;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& d : done)
        if (d.f == func && d.dev == ctx->device) {
            if (d.bytes >= bytes) return XDEMHIP_OK;
            XD_HIP_CHECK(ctx, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            d.bytes = bytes;
            return XDEMHIP_OK;
        }
    XD_HIP_CHECK(ctx, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done.push_back({func, ctx->device, bytes});
    return XDEMHIP_OK;
}
template <typename F> int set_big_lds(xdemhip_ctx* ctx, F func, size_t bytes) {
    return set_big_lds_ptr(ctx, reinterpret_cast<const void*>(func), bytes);
}

inline int grid_for(const xdemhip_ctx* ctx, int64_t n, int block, int per_cu) {
    int64_t g = (n + block - 1) / block;
    const int64_t cap = (int64_t)ctx->num_cu * per_cu;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}


// scratch layout (bytes): [0, 16384) bin edges | stats | sums | selection states | successor keys | histograms
constexpr size_t OFF_STATS = 16384, OFF_SUMS = OFF_STATS + 64, OFF_NEED = OFF_STATS + 96 /* uint32: successor pass needed */, OFF_STATE = OFF_STATS + 128;
inline int nb1(int nb) { return nb > 1 ? nb : 1; }
inline size_t off_succ(int nb) { return OFF_STATE + (size_t)nb1(nb) * 64; }
inline size_t off_hist(int nb) { return off_succ(nb) + (size_t)nb1(nb) * 8; }
// (sized for 2 nb states: the dual bracket selection keeps a low-end and a high-end state per bin)
inline size_t scratch_size(int nb) { return off_hist(2 * nb1(nb)) + (size_t)2 * nb1(nb) * SEL_RADIX * 8 + 256; }

// states | successor keys | histograms of a selection, contiguous in `scratch`: zeros, all-ones, zeros.  A device function so that a
// kernel that runs before the selection anyway (the sample kernels of the Nuth-Kaab step) can do it on the side: SelReset says where.
struct SelReset { uint64_t* base; int64_t w_state, w_succ, words; uint32_t* need; };
__device__ __forceinline__ void select_reset_slice(const SelReset& r, int64_t gtid, int64_t gthreads) {
    for (int64_t w = gtid; w < r.words; w += gthreads) r.base[w] = (w >= r.w_state && w < r.w_state + r.w_succ) ? ~(uint64_t)0 : (uint64_t)0;
    if (r.need && gtid == 0) { r.need[0] = 0u; r.need[2] = 0u; }   // ([2]: the ticket of hist_pass_kernel<T, true>)
}
static __global__ void select_reset_kernel(uint64_t* base, int64_t w_state, int64_t w_succ, int64_t words, uint32_t* need = nullptr) {
    const SelReset r = {base, w_state, w_succ, words, need};
    select_reset_slice(r, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}
// what select_enqueue's own reset launch would do for a selection over `nb` data bins in `mode` (for callers that pass reset_done)
template <typename K> inline SelReset select_reset_plan(unsigned char* scratch, int nb, int mode) {
    static_assert(sizeof(SelState<K>) <= 64, "state records live in 64-byte slots");
    if (mode == SEL_BRACKET_DUAL) nb *= 2;
    SelReset r;
    r.base = reinterpret_cast<uint64_t*>(scratch + OFF_STATE);
    r.w_state = (int64_t)nb1(nb) * 8;
    r.w_succ = (int64_t)nb1(nb);
    r.words = r.w_state + r.w_succ + (int64_t)nb * SEL_RADIX;
    r.need = reinterpret_cast<uint32_t*>(scratch + OFF_NEED);
    return r;
}

template <typename K> struct SelResult {
    SelState<K> st;
    uint64_t succ;  // smallest key above the selected one, all-ones if none
};

// Exact lower/upper medians of vals[0..n) per bin (bins == nullptr: one bin).  With an all-reduce hook installed the
// integer histograms / successor keys are combined over the ranks after every pass, so every rank selects the same
// global order statistics from its own share of the data.
//
// select_enqueue only queues the passes on the context's stream (no host synchronisation unless the hook is installed):
// the states and successor keys stay in `scratch` until select_fetch copies them out.  `d_n` (optional) is a device-side
// element count -- `n` is then the capacity of the buffer and `n_grid` the size the launch grids are dimensioned for.
template <typename T>
int select_enqueue(xdemhip_ctx* ctx, const T* vals, const uint16_t* bins, int64_t n, int64_t n_grid, const unsigned long long* d_n, int nb,
                   unsigned char* scratch, int mode, const uint64_t* d_given, int n_passes = 0, bool want_succ = true,
                   const typename KeyT<T>::type* rb_lo = nullptr, const uint32_t* rb_shift = nullptr,
                   bool first_hist_done = false /* the caller ran select_reset and filled the first digit's histogram itself */,
                   int narrow = 0 /* bracket modes: half width >> narrow (select_advance_kernel) */,
                   typename KeyT<T>::type* fuse_klo = nullptr /* SEL_BRACKET_DUAL: the passes advance their own states and the last one writes the */,
                   typename KeyT<T>::type* fuse_khi = nullptr /* bracket ends + rebase shift here (hist_pass_kernel<T, true>); *fused tells whether */,
                   uint32_t* fuse_rbs = nullptr, typename KeyT<T>::type fuse_low_mask = 0, bool* fused = nullptr,
                   bool reset_done = false /* the caller's own kernel did select_reset_slice(select_reset_plan(...)) */,
                   int extra_pass = -1, int64_t extra_words = 0 /* with a reduction hook: the all-reduce of pass `extra_pass` also carries the
                                                                    `extra_words` 8-byte words that follow the histograms in `scratch`
                                                                    (per-rank slots of the caller: a sum all-reduce used as an all-gather) */) {
    typedef typename KeyT<T>::type K;
    if (fused) *fused = false;
    // SEL_BRACKET_DUAL: two selection states per data bin (low ends in states [0, nb), high ends in [nb, 2 nb)); `scratch` holds
    // 2 nb states (scratch_size provides for that)
    const int dual = mode == SEL_BRACKET_DUAL ? 1 : 0;
    if (dual && want_succ) return xd_fail(ctx, XDEMHIP_EINVAL, "dual bracket selection: no successor pass");
    const int nb_data = nb;
    if (dual) { nb = 2 * nb_data; if (nb_data == 1) bins = nullptr; }
    SelState<K>* st = reinterpret_cast<SelState<K>*>(scratch + OFF_STATE);
    uint64_t* d_succ = reinterpret_cast<uint64_t*>(scratch + off_succ(nb));
    uint64_t* d_hist = reinterpret_cast<uint64_t*>(scratch + off_hist(nb));
    uint32_t* d_need = reinterpret_cast<uint32_t*>(scratch + OFF_NEED);
    if (!first_hist_done && !reset_done) {
        // states | successor keys | histograms are contiguous in `scratch`: one launch resets all three (zeros, all-ones, zeros)
        const SelReset r = select_reset_plan<K>(scratch, nb_data, mode);
        const int blocks = (int)((r.words + 1023) / 1024 < 256 ? (r.words + 1023) / 1024 : 256);
        hipLaunchKernelGGL(select_reset_kernel, dim3(blocks), dim3(256), 0, ctx->stream, r.base, r.w_state, r.w_succ, r.words, r.need);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    const int passes = KeyT<T>::passes;
    const int run = (n_passes > 0 && n_passes < passes) ? n_passes : passes;  // leading digits only: bracket ends need no more
    // (every workgroup zeroes and flushes a whole [bins][256] table: for small inputs -- the 1/64 samples -- one workgroup per CU
    // halves that fixed cost, which dominates their passes)
    // (measured on the Nuth-Kaab step, A/B in one session: 2 -> 1 workgroups per CU 3.00 / 3.13 -> 2.92 / 2.98 ms; half a
    // workgroup per CU 3.10: too few)
    const int grid = grid_for(ctx, n_grid, HIST_THREADS * SEL_UNROLL, n_grid < ((int64_t)1 << 24) ? 1 : 2);
    const bool fuse = fuse_klo && fuse_khi && fuse_rbs && dual && !ctx->allreduce && n > 0 && !first_hist_done && !rb_lo && !rb_shift &&
                      nb <= HIST_THREADS / 64;                          // (one state per wave of the last workgroup: with the 144 states of the 72
                                                                         //  aspect bins that workgroup took 9 rounds of device-scope loads, 20-40 us
                                                                         //  against the 5 us of a launch of 144 one-wave workgroups -- measured)
    if (fused) *fused = fuse;
    for (int p = 0; p < run; ++p) {
        const int shift = 8 * (passes - 1 - p);
        if (n > 0 && !(p == 0 && first_hist_done)) {
            const int per_sweep = dual ? MAX_ROWS_PER_SWEEP_DUAL / 2 : MAX_BINS_PER_SWEEP;
            for (int b0 = 0; b0 < nb_data; b0 += per_sweep) {
                const int nbs = (nb_data - b0) < per_sweep ? (nb_data - b0) : per_sweep;
                const int rows = dual ? 2 * nbs : nbs;
                // privatise the table as often as fits in ~64 KB (32 copies for the single-bin global median)
                int copies = (64 * 1024) / (rows * SEL_RADIX * (int)sizeof(uint32_t));
                copies = copies < 1 ? 1 : (copies > 32 ? 32 : copies);
                const size_t lds = ((size_t)rows * SEL_RADIX + 1) * sizeof(uint32_t) * copies + 8 + 2 * sizeof(K) * (size_t)rows;
                int rc = set_big_lds(ctx, hist_pass_kernel<T>, lds);
                if (rc) return rc;
                // every workgroup flushes a whole [rows][256] table with global atomics: with many rows and few elements (the 72-bin
                // samples: 36 k counters against 24 k elements per workgroup) the flush IS the pass -- no more workgroups than give
                // each of them two tables' worth of elements (at least 32)
                // (the SECOND digit's pass: the first digit of float keys -- sign and exponent -- leaves most counters empty and from
                // the third on few elements still match their prefix; measured on the 72-bin samples of the Nuth-Kaab step:
                // 87 -> 55 us for that pass, the others lose when they get fewer workgroups)
                int64_t g2 = n_grid / ((int64_t)2 * rows * SEL_RADIX);
                g2 = g2 < 32 ? 32 : g2;
                const int grid_p = (p == 1 && g2 < grid) ? (int)g2 : grid;
                if (fuse) {
                    HistFuse<K> fa;
                    fa.ticket = d_need + 2; fa.st = st; fa.n_states = nb; fa.dual_nb = nb_data; fa.last = (int)(p == passes - 1); fa.narrow = (uint32_t)narrow;
                    fa.finish = (int)(p == run - 1); fa.low_mask = fuse_low_mask; fa.klo = fuse_klo; fa.khi = fuse_khi; fa.rb_shift_out = fuse_rbs;
                    rc = set_big_lds(ctx, hist_pass_kernel<T, true>, lds);
                    if (rc) return rc;
                    hipLaunchKernelGGL((hist_pass_kernel<T, true>), dim3(grid_p), dim3(HIST_THREADS), lds, ctx->stream, vals, bins, n, nbs, b0, copies,
                                       st, shift, (int)(p == 0), d_hist, d_n, rb_lo, rb_shift, nb_data, fa);
                    XD_HIP_CHECK(ctx, hipGetLastError());
                    continue;
                }
                hipLaunchKernelGGL((hist_pass_kernel<T>), dim3(grid_p), dim3(HIST_THREADS), lds, ctx->stream, vals, bins, n, nbs, b0, copies,
                                   st, shift, (int)(p == 0), d_hist, d_n, rb_lo, rb_shift, dual ? nb_data : 0);
                XD_HIP_CHECK(ctx, hipGetLastError());
            }
        }
        if (fuse) continue;   // (states advanced by the pass itself)
        int rc = xd_allreduce_device(ctx, d_hist, (int64_t)nb * SEL_RADIX + (p == extra_pass ? extra_words : 0), XDEMHIP_RED_SUM_U64);
        if (rc) return rc;
        hipLaunchKernelGGL((select_advance_kernel<K>), dim3(nb), dim3(64), 0, ctx->stream, st, d_hist, nb, shift,
                           (int)(p == 0), (int)(p == passes - 1), mode, d_given, rb_shift, PAIR_DEFF_WIDE, nb_data,
                           want_succ ? d_succ : nullptr, want_succ ? d_need : nullptr, (uint32_t)narrow);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    if (!want_succ) return XDEMHIP_OK;
    if (n > 0) {
        // (leaves at once unless some bin's selected key is the largest of its leading-digit group: see select_advance_kernel)
        hipLaunchKernelGGL((succ_pass_kernel<T>), dim3(grid), dim3(HIST_THREADS), 3 * sizeof(K) * nb, ctx->stream, vals, bins, n, nb, st,
                           d_succ, d_n, rb_lo, rb_shift, d_need);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    return xd_allreduce_device(ctx, d_succ, nb, XDEMHIP_RED_MIN_U64);
}

// The reset of select_enqueue on its own, for callers that fill the first digit's histogram themselves (first_hist_done): returns
// the histogram table [nb][256] they add to.
template <typename K> uint64_t* select_reset(xdemhip_ctx* ctx, unsigned char* scratch, int nb) {
    SelState<K>* st = reinterpret_cast<SelState<K>*>(scratch + OFF_STATE);
    const int64_t w_state = (int64_t)nb1(nb) * 8, w_succ = (int64_t)nb1(nb), w_hist = (int64_t)nb * SEL_RADIX;
    const int64_t words = w_state + w_succ + w_hist;
    const int blocks = (int)((words + 1023) / 1024 < 256 ? (words + 1023) / 1024 : 256);
    hipLaunchKernelGGL(select_reset_kernel, dim3(blocks), dim3(256), 0, ctx->stream, reinterpret_cast<uint64_t*>(st), w_state, w_succ, words,
                       reinterpret_cast<uint32_t*>(scratch + OFF_NEED));
    return reinterpret_cast<uint64_t*>(scratch + off_hist(nb));
}

template <typename T>
int select_fetch(xdemhip_ctx* ctx, unsigned char* scratch, int nb, std::vector<SelResult<typename KeyT<T>::type>>& out) {
    typedef typename KeyT<T>::type K;
    std::vector<SelState<K>> hs(nb);
    std::vector<uint64_t> hsucc(nb);
    { const int rc_ = xd_d2h(ctx, hs.data(), scratch + OFF_STATE, sizeof(SelState<K>) * nb); if (rc_) return rc_; }
    { const int rc_ = xd_d2h(ctx, hsucc.data(), scratch + off_succ(nb), 8 * (size_t)nb); if (rc_) return rc_; }
    { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
    out.resize(nb);
    for (int k = 0; k < nb; ++k) { out[k].st = hs[k]; out[k].succ = hsucc[k]; }
    return XDEMHIP_OK;
}

template <typename T>
int run_select_core(xdemhip_ctx* ctx, const T* vals, const uint16_t* bins, int64_t n, int nb, unsigned char* scratch,
                    std::vector<SelResult<typename KeyT<T>::type>>& out, int mode, const uint64_t* d_given) {
    const int rc = select_enqueue<T>(ctx, vals, bins, n, n, nullptr, nb, scratch, mode, d_given);
    return rc ? rc : select_fetch<T>(ctx, scratch, nb, out);
}

// ---- bracketed selection ------------------------------------------------------------------------------------------
// The plain selection above reads the data once per key digit (4 passes for float32, 8 for float64) plus once for the
// successor.  For large inputs the medians are first bracketed from a ~1/64 sample made of whole 32-element lines
// (pseudo-randomly chosen, so only those lines are fetched), then ONE pass over the data counts, per bin, the elements
// below the bracket and compacts the few inside it; the exact order statistics are finally selected among those
// candidates.  Everything stays integer and exact: the counts prove that the wanted ranks lie inside the brackets; if a
// bracket misses (or the candidate buffer overflows) the plain selection runs instead.  Counts, sample and candidate
// histograms go through the same all-reduce hook, so the result is identical on every rank.
struct SelWorkspace {
    void* s_vals = nullptr; uint16_t* s_bins = nullptr; int64_t s_cap = 0;   // sample
    void* c_vals = nullptr; uint16_t* c_bins = nullptr; int64_t c_cap = 0;   // candidates
    uint64_t* d_small = nullptr;  // [0] sample count, [1] candidate count, [2] overflow | klo | khi | given | counters[3 nb]
    int nb_max = 0;
    size_t es = 4;
};
constexpr int64_t SEL_BRACKET_MIN_N = (int64_t)1 << 22;
// The bracket of a bin spans 6 sqrt(32 / m) of its elements (m = its share of the sample, n / (64 nb) on average): below
// ~7000 sampled elements per bin more than 40 % of the data would be candidates and the plain passes are the better route.
constexpr int64_t SEL_BRACKET_MIN_PER_BIN = 460000;

inline void sel_ws_free(SelWorkspace& w) {
    void* b[] = {w.s_vals, w.s_bins, w.c_vals, w.c_bins, w.d_small};
    for (void* p : b)
        if (p) (void)hipFree(p);
    w = SelWorkspace();
}
inline int sel_ws_create(xdemhip_ctx* ctx, int64_t n, size_t es, int nb_max, SelWorkspace& w) {
    w = SelWorkspace();
    w.es = es; w.nb_max = nb_max;
    w.s_cap = n / 24 + 4096;   // expected n / 64
    w.c_cap = n / 2 + 4096;
    if (hipMalloc(&w.s_vals, (size_t)w.s_cap * es) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&w.s_bins), (size_t)w.s_cap * 2) != hipSuccess ||
        hipMalloc(&w.c_vals, (size_t)w.c_cap * es) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&w.c_bins), (size_t)w.c_cap * 2) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&w.d_small), (size_t)(8 + 6 * nb_max) * 8) != hipSuccess) {
        sel_ws_free(w);
        return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc(selection workspace) failed");
    }
    return XDEMHIP_OK;
}

// Stratified 1/64 sample of SEL_LINE-element lines: group g covers lines [64 g, 64 g + 64) and contributes the one line picked
// by the top 6 bits of a multiplicative hash of g (no aliasing with the raster's row period, and sampled lines can be
// enumerated directly instead of testing every line).  Lines are 8 elements (one 32-byte sector of a float32 array): the
// bracket half width is sized for FULLY correlated lines (select.h: sel_bracket_halfwidth), so for the same sample volume
// 8-element lines give brackets half as wide as the 32-element lines of round 1 -- half the candidates in the pass over all
// elements and in the digit passes over the candidates -- for a sample pass that fetches 32-byte instead of 128-byte pieces.
constexpr int SEL_LINE_LOG2 = 3, SEL_LINE = 1 << SEL_LINE_LOG2;
__device__ __forceinline__ int64_t sel_sampled_line(int64_t group) {
    return group * 64 + (int64_t)((uint64_t)group * 0x9E3779B97F4A7C15ull >> 58);
}

// Block-level compaction: selected (value, bin) pairs collect in an LDS staging buffer (one LDS atomic per wave and step)
// and leave in coalesced bursts, ONE global atomic per burst of thousands.  (A global atomic per wave would serialise
// on a single address: ~1e8 updates/s, far below the data rate.)
constexpr int SEL_TILE = 4;                        // elements per thread and step
constexpr int SEL_STAGE_CAP = 8192;                // staging slots per workgroup
constexpr int SEL_FLUSH_EVERY = 4;                 // bracket pass: steps (of SEL_TILE x 1024 elements) between two flushes
template <typename T, int CAP = 8192 /* = SEL_STAGE_CAP: staging slots of the workgroup */> struct BlockStage {
    T* v;
    uint16_t* b;
    int* held;                   // LDS
    unsigned long long* base;    // LDS
    __device__ __forceinline__ void append(bool keep, T val, uint16_t bin) {
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
        if (!mask) return;
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)mask) - 1;
        int pos0 = 0;
        if (lane == leader) pos0 = atomicAdd(held, __popcll(mask));
        pos0 = __shfl(pos0, leader);
        if (keep) {
            const int pos = pos0 + __popcll(mask & ((1ull << lane) - 1ull));
            v[pos] = val;
            b[pos] = bin;
        }
    }
    // same, for producers without a bound on the elements per step: a full buffer raises the overflow flag instead of writing
    __device__ __forceinline__ void append_bounded(bool keep, T val, uint16_t bin, unsigned long long* overflow) {
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
        if (!mask) return;
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)mask) - 1;
        int pos0 = 0;
        if (lane == leader) pos0 = atomicAdd(held, __popcll(mask));
        pos0 = __shfl(pos0, leader);
        if (keep) {
            const int pos = pos0 + __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < CAP) { v[pos] = val; b[pos] = bin; }
            else *overflow = 1ull;
        }
    }
    // called by every thread of the workgroup after each step (and with force at the end)
    __device__ __forceinline__ void sync_and_flush(bool force, T* out_v, uint16_t* out_b, unsigned long long* counter, int64_t cap,
                                                   unsigned long long* overflow) {
        sync_and_flush_at(force ? 0 : CAP - SEL_TILE * (int)blockDim.x, out_v, out_b, counter, cap, overflow);
    }
    // barrier, then flush if more than `threshold` elements are staged
    __device__ __forceinline__ void sync_and_flush_at(int threshold, T* out_v, uint16_t* out_b, unsigned long long* counter, int64_t cap,
                                                      unsigned long long* overflow) {
        __syncthreads();
        int h = *held;
        h = h > CAP ? CAP : h;  // (bounded appends may have counted past the end)
        if (h > threshold) {
            if (threadIdx.x == 0) *base = atomicAdd(counter, (unsigned long long)h);
            __syncthreads();
            const unsigned long long b0 = *base;
            for (int i = threadIdx.x; i < h; i += blockDim.x) {
                const unsigned long long pos = b0 + (unsigned long long)i;
                if ((int64_t)pos < cap) { out_v[pos] = v[i]; out_b[pos] = b[i]; }
                else *overflow = 1ull;
            }
            __syncthreads();
            if (threadIdx.x == 0) *held = 0;
            __syncthreads();
        }
    }
};

// Element sources of the sample / bracket passes.  A source hands out element p in two steps -- fetch (global loads only,
// so that a step's loads can all be issued first) and eval (value, bin id, usable?) -- and may keep a table in LDS.
// ArraySource: plain (values, bin ids) arrays.  Other sources compute the values on the fly from the arrays they derive
// from (nuthkaab.hip: y = (dh - vshift) / slope_tan binned by aspect), which saves writing and re-reading them.
template <typename T> struct ArraySource {
    const T* vals;
    const uint16_t* bins;  // nullptr: single bin
    struct Raw { T v; uint16_t b; };
    struct Acc {};
    static size_t lds_bytes(int) { return 0; }
    __device__ __forceinline__ void setup(unsigned char*, int) {}
    __device__ __forceinline__ void fetch(int64_t p, Raw& r) const { r.v = vals[p]; r.b = bins ? bins[p] : (uint16_t)0; }
    __device__ __forceinline__ void blank(Raw& r) const { r.v = (T)NAN; r.b = 0; }
    template <bool ACC> __device__ __forceinline__ bool eval(const Raw& r, int nb, T& v, uint16_t& b, Acc&) const {
        v = r.v; b = r.b;
        return (v == v) && (int)b < nb;
    }
    __device__ __forceinline__ void finish(Acc&) const {}
    static constexpr bool HAS_LEAN = false;  // (sources with a dedicated counting kernel: see NkYSource)
};

template <typename T, typename Src>
__global__ __launch_bounds__(HIST_THREADS) void sample_lines_kernel(Src src, int64_t n, int nb, T* out_v, uint16_t* out_b,
                                                                    unsigned long long* ctr, int64_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    BlockStage<T> st;
    st.v = reinterpret_cast<T*>(smem);
    st.b = reinterpret_cast<uint16_t*>(st.v + SEL_STAGE_CAP);
    st.base = reinterpret_cast<unsigned long long*>(st.b + SEL_STAGE_CAP);
    st.held = reinterpret_cast<int*>(st.base + 1);
    src.setup(reinterpret_cast<unsigned char*>(st.base + 2), nb);
    if (threadIdx.x == 0) *st.held = 0;
    __syncthreads();
    typename Src::Acc acc;
    // a step covers one sampled line per SEL_LINE lanes (128 line groups = 65536 elements per 1024-thread workgroup)
    const int64_t n_groups = (((n + SEL_LINE - 1) >> SEL_LINE_LOG2) + 63) >> 6;
    const int halves = (int)blockDim.x >> SEL_LINE_LOG2;
    const int half = (int)threadIdx.x >> SEL_LINE_LOG2, l32 = (int)threadIdx.x & (SEL_LINE - 1);
    // SEL_TILE line groups per thread and step: all of a step's (scattered, 32-byte) fetches are issued before the first is used,
    // and the workgroup synchronises once per step -- with one group per step the kernel was a chain of a dozen dependent memory
    // round trips per workgroup (81 us for the 1/64 sample of a 20000^2 raster, as long as a pass over 8 % of the data)
    for (int64_t g0 = (int64_t)blockIdx.x * halves * SEL_TILE; g0 < n_groups; g0 += (int64_t)gridDim.x * halves * SEL_TILE) {
        typename Src::Raw r[SEL_TILE];
        bool have[SEL_TILE];
#pragma unroll
        for (int q = 0; q < SEL_TILE; ++q) {
            const int64_t g = g0 + (int64_t)q * halves + half;
            const int64_t p = (sel_sampled_line(g) << SEL_LINE_LOG2) + l32;
            have[q] = g < n_groups && p < n;
            if (have[q]) src.fetch(p, r[q]);
            else src.blank(r[q]);
        }
#pragma unroll
        for (int q = 0; q < SEL_TILE; ++q) {
            T v = (T)0;
            uint16_t b = 0;
            bool keep = false;
            if (have[q]) keep = src.template eval<false>(r[q], nb, v, b, acc);
            st.append(keep, v, b);
        }
        st.sync_and_flush(false, out_v, out_b, &ctr[0], cap, &ctr[2]);
    }
    st.sync_and_flush(true, out_v, out_b, &ctr[0], cap, &ctr[2]);
}

// Brackets from the two sample selections: which = 0 stores the low keys after the SEL_BRACKET_LO selection, which = 1 the
// high keys after SEL_BRACKET_HI (`degenerate`: high = low, a bracket that almost surely misses -- test mode of the fallback).
// (The sample selections fix only the leading digits: `low_mask` = the digits left open, all ones at the high end.)
template <typename K>
__global__ void bracket_keys_kernel(const SelState<K>* st, int nb, int which, int degenerate, K low_mask, K* klo, K* khi) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const bool have = st[b].count > 0;
    if (which == 0) klo[b] = have ? st[b].prefix : (K)0;
    else khi[b] = have ? (degenerate ? klo[b] : (K)(st[b].prefix | low_mask)) : (K)~(K)0;
}

// Left shift that brings the widest bracket's (hi - lo) up to the top key bit (rebased candidate keys, hist_pass_kernel).
template <typename K>
__global__ __launch_bounds__(64) void rebase_shift_kernel(const K* klo, const K* khi, int nb, uint32_t* shift) {
    K r = 0;
    for (int b = threadIdx.x; b < nb; b += 64) {
        const K d = khi[b] >= klo[b] ? (K)(khi[b] - klo[b]) : (K)0;
        r = d > r ? d : r;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const K o = k_shfl_down(r, off);
        r = o > r ? o : r;
    }
    if (threadIdx.x == 0) *shift = rebase_shift_of(r);
}

// bracket_keys_kernel (both ends) + rebase_shift_kernel in one launch: states [0, nb) hold the low ends, [nb, 2 nb) the high ends
template <typename K>
__global__ __launch_bounds__(64) void bracket_finish_kernel(const SelState<K>* st, int nb, int degenerate, K low_mask, K* klo, K* khi, uint32_t* shift) {
    bracket_finish_body<K>((int)threadIdx.x, st, nb, degenerate, low_mask, klo, khi, shift);
}

// Rank of the wanted order statistic among the candidates of every bin (all-ones: empty bin); raises flags[3] when a
// bracket does not hold the lower (and, for an even count, the upper) median.
static __global__ void bracket_given_kernel(const uint64_t* cnt /* [3][nb]: total, below, inside */, int nb, uint64_t* given,
                                     unsigned long long* flags) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint64_t total = cnt[b], lt = cnt[nb + b], in = cnt[2 * nb + b];
    uint64_t g = ~(uint64_t)0;
    if (total) {
        const uint64_t k = (total - 1) / 2;
        const uint64_t need = (total & 1) ? k : k + 1;
        if (lt > k || need - lt >= in) flags[3] = 1ull;
        else g = k - lt;
    }
    given[b] = g;
}

// One pass over the data: per bin the number of (non-NaN) elements, of elements below the bracket and inside it; elements
// inside [klo, khi] are compacted through the staging buffer.  LDS: staging | klo / khi per bin | `copies` privatised sets of
// 3 counters per bin.
template <typename T, typename Src>
__global__ __launch_bounds__(HIST_THREADS) void bracket_pass_kernel(Src src, int64_t n,
                                                                    int nb, int copies, const typename KeyT<T>::type* __restrict__ klo,
                                                                    const typename KeyT<T>::type* __restrict__ khi, uint64_t* counters /* [3][nb] */,
                                                                    T* out_v, uint16_t* out_b, unsigned long long* ctr, int64_t cap) {
    typedef typename KeyT<T>::type K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    BlockStage<T> st;
    st.v = reinterpret_cast<T*>(smem);
    st.b = reinterpret_cast<uint16_t*>(st.v + SEL_STAGE_CAP);
    st.base = reinterpret_cast<unsigned long long*>(st.b + SEL_STAGE_CAP);
    st.held = reinterpret_cast<int*>(st.base + 1);
    K* lo = reinterpret_cast<K*>(st.base + 2);
    K* hi = lo + nb;
    uint32_t* c = reinterpret_cast<uint32_t*>(hi + nb);
    for (int k = threadIdx.x; k < nb; k += blockDim.x) { lo[k] = klo[k]; hi[k] = khi[k]; }
    const int cs = (3 * nb) | 1;  // odd copy stride: the copies of one counter fall into different LDS banks
    for (int k = threadIdx.x; k < cs * copies; k += blockDim.x) c[k] = 0;
    src.setup(reinterpret_cast<unsigned char*>(c + cs * copies), nb);
    if (threadIdx.x == 0) *st.held = 0;
    __syncthreads();
    if constexpr (Src::HAS_LEAN) {
        if (src.skip_counting_pass()) return;  // (uniform) the source's own kernel, queued right behind, does this step
    }
    uint32_t* cc = c + (threadIdx.x % copies) * cs;
    typename Src::Acc acc;
    int it = 0;
    const int64_t step = (int64_t)blockDim.x * SEL_TILE;
    for (int64_t base = (int64_t)blockIdx.x * step; base < n; base += (int64_t)gridDim.x * step) {
        typename Src::Raw raw[SEL_TILE];
#pragma unroll
        for (int q = 0; q < SEL_TILE; ++q) {  // all loads of the step first
            const int64_t p = base + (int64_t)q * blockDim.x + threadIdx.x;
            if (p < n) src.fetch(p, raw[q]);
            else src.blank(raw[q]);
        }
#pragma unroll
        for (int q = 0; q < SEL_TILE; ++q) {
            bool cand = false;
            T v;
            uint16_t b;
            if (src.template eval<true>(raw[q], nb, v, b, acc)) {
                // exactly one LDS atomic per element: class 0 above the bracket, 1 below, 2 inside (the totals are summed at the end)
                const K key = key_of(v);
                const K l = lo[b], h = hi[b];
                cand = (key >= l) & (key <= h);
                const int cls = key < l ? 1 : (cand ? 2 : 0);
                atomicAdd(&cc[cls * nb + b], 1u);
            }
            st.append_bounded(cand, v, b, &ctr[2]);
        }
        // The staging buffer is emptied (one barrier pair + a coalesced burst) every SEL_FLUSH_EVERY steps only: it holds half
        // of the elements of that many steps, and the bracketed route is not taken when more than ~40 % of a bin would be
        // candidates.  A burst of candidates beyond that raises the overflow flag (-> plain selection), never a bad write.
        if ((++it % SEL_FLUSH_EVERY) == 0) st.sync_and_flush_at(0, out_v, out_b, &ctr[1], cap, &ctr[2]);
    }
    st.sync_and_flush_at(0, out_v, out_b, &ctr[1], cap, &ctr[2]);
    src.finish(acc);
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        unsigned long long above = 0, below = 0, inside = 0;
        for (int q = 0; q < copies; ++q) {
            above += c[q * cs + k]; below += c[q * cs + nb + k]; inside += c[q * cs + 2 * nb + k];
        }
        if (above + below + inside) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[k]), above + below + inside);
        if (below) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[nb + k]), below);
        if (inside) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[2 * nb + k]), inside);
    }
}

// Per-bin sums and counts of a source's elements in one pass (mean-type bin statistics: sum-reducible, so the tables
// all-reduce directly): LDS-privatised float64 sums (ds_add_f64) and uint32 counts, flushed with one global atomic per
// non-empty (bin, workgroup).
template <typename T, typename Src>
__global__ __launch_bounds__(HIST_THREADS) void bin_sums_kernel(Src src, int64_t n, int nb, int copies, double* sums /* [nb] */,
                                                                unsigned long long* counts /* [nb] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* s = reinterpret_cast<double*>(smem);
    uint32_t* c = reinterpret_cast<uint32_t*>(s + (size_t)nb * copies);
    for (int k = threadIdx.x; k < nb * copies; k += blockDim.x) { s[k] = 0.0; c[k] = 0; }
    src.setup(reinterpret_cast<unsigned char*>(c + (size_t)nb * copies), nb);
    __syncthreads();
    double* sc = s + (threadIdx.x % copies) * nb;
    uint32_t* cc = c + (threadIdx.x % copies) * nb;
    typename Src::Acc acc;
    const int64_t step = (int64_t)blockDim.x * SEL_TILE;
    for (int64_t base = (int64_t)blockIdx.x * step; base < n; base += (int64_t)gridDim.x * step) {
        typename Src::Raw raw[SEL_TILE];
#pragma unroll
        for (int q = 0; q < SEL_TILE; ++q) {
            const int64_t p = base + (int64_t)q * blockDim.x + threadIdx.x;
            if (p < n) src.fetch(p, raw[q]);
            else src.blank(raw[q]);
        }
#pragma unroll
        for (int q = 0; q < SEL_TILE; ++q) {
            T v;
            uint16_t b;
            if (src.template eval<true>(raw[q], nb, v, b, acc)) {
                unsafeAtomicAdd(&sc[b], (double)v);
                atomicAdd(&cc[b], 1u);
            }
        }
    }
    __syncthreads();
    src.finish(acc);
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        double t = 0.0;
        unsigned long long m = 0;
        for (int q = 0; q < copies; ++q) { t += s[q * nb + k]; m += c[q * nb + k]; }
        if (m) { unsafeAtomicAdd(&sums[k], t); atomicAdd(&counts[k], m); }
    }
}

template <typename T, typename Src>
int run_bin_sums(xdemhip_ctx* ctx, const Src& src, int64_t n, int nb, double* d_sums, unsigned long long* d_counts) {
    XD_HIP_CHECK(ctx, hipMemsetAsync(d_sums, 0, 8 * (size_t)nb, ctx->stream));
    XD_HIP_CHECK(ctx, hipMemsetAsync(d_counts, 0, 8 * (size_t)nb, ctx->stream));
    if (n > 0) {
        int copies = (48 * 1024) / (nb * 12);
        copies = copies < 1 ? 1 : (copies > 32 ? 32 : copies);
        const size_t lds = (size_t)nb * copies * 12 + Src::lds_bytes(nb);
        int rc = set_big_lds(ctx, bin_sums_kernel<T, Src>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((bin_sums_kernel<T, Src>), dim3(grid_for(ctx, n, HIST_THREADS * SEL_TILE, 2)), dim3(HIST_THREADS), lds, ctx->stream, src, n,
                           nb, copies, d_sums, d_counts);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    int rc = xd_allreduce_device(ctx, d_sums, nb, XDEMHIP_RED_SUM_F64);
    if (rc) return rc;
    return xd_allreduce_device(ctx, d_counts, nb, XDEMHIP_RED_SUM_U64);
}

// Bracketed selection over an element source.  *done = false (and `out` meaningless) when the route is not available --
// small input, no workspace, plain mode, some rank cannot -- or when a bracket missed / a buffer overflowed: the caller
// then runs the plain selection on materialised arrays.
template <typename T, typename Src>
int run_select_bracketed(xdemhip_ctx* ctx, const Src& src, int64_t n, int nb, unsigned char* scratch,
                         std::vector<SelResult<typename KeyT<T>::type>>& out, SelWorkspace* ws, bool* done, bool* passed = nullptr,
                         const std::function<int()>* before_sync = nullptr /* queues the caller's own result copies behind the route */) {
    typedef typename KeyT<T>::type K;
    *done = false;
    if (passed) *passed = false;  // set once the pass over all elements has been queued (sources with side effects rely on it)
    static const bool disabled = getenv("XDEMHIP_NO_BRACKET") != nullptr;  // (A/B timing knob)
    const bool plain = disabled || ctx->selection_mode == 1 || !ws || !ws->d_small || nb > ws->nb_max || ws->es != sizeof(T) || nb > MAX_BINS_PER_SWEEP ||
                       n < SEL_BRACKET_MIN_N || (ctx->selection_mode == 0 && n < SEL_BRACKET_MIN_PER_BIN * nb) || (n / 24 + 4096) > ws->s_cap;
    if (ctx->allreduce) {  // sharded data: every rank must take the same route (local sizes / allocations may differ)
        uint64_t can = plain ? 0 : 1;
        if (ctx->allreduce(&can, 1, XDEMHIP_RED_MIN_U64, ctx->allreduce_user) != 0) return xd_fail(ctx, XDEMHIP_EHIP, "all-reduce hook failed");
        if (!can) return XDEMHIP_OK;
    } else if (plain) {
        return XDEMHIP_OK;
    }
    uint64_t* d_ctr = ws->d_small;  // [0] sample count, [1] candidate count, [2] overflow, [3] bracket missed
    unsigned long long* d_flags = reinterpret_cast<unsigned long long*>(d_ctr);
    K* d_klo = reinterpret_cast<K*>(ws->d_small + 8);
    K* d_khi = reinterpret_cast<K*>(ws->d_small + 8 + ws->nb_max);
    uint64_t* d_given = ws->d_small + 8 + 2 * ws->nb_max;
    uint64_t* d_cnt = ws->d_small + 8 + 3 * ws->nb_max;
    const SelState<K>* d_st = reinterpret_cast<const SelState<K>*>(scratch + OFF_STATE);
    XD_HIP_CHECK(ctx, hipMemsetAsync(ws->d_small, 0, (size_t)(8 + 6 * ws->nb_max) * 8, ctx->stream));
    // Everything below is queued back to back: sample and candidate counts, brackets and ranks stay on the device, and the
    // host synchronises once at the end (with the all-reduce hook every reduction synchronises anyway).
    // 1. sample
    const size_t lds_stage = (size_t)SEL_STAGE_CAP * (sizeof(T) + 2) + 16;
    const size_t lds_src = Src::lds_bytes(nb);
    int rc = set_big_lds(ctx, sample_lines_kernel<T, Src>, lds_stage + lds_src);
    if (rc) return rc;
    hipLaunchKernelGGL((sample_lines_kernel<T, Src>), dim3(grid_for(ctx, n / 64 + 1, HIST_THREADS, 2)), dim3(HIST_THREADS), lds_stage + lds_src,
                       ctx->stream, src, n, nb, static_cast<T*>(ws->s_vals), ws->s_bins, d_flags, ws->s_cap);
    XD_HIP_CHECK(ctx, hipGetLastError());
    // 2. brackets from the sample (global over the ranks through the hook)
    const int nbb = (nb + 63) / 64;
    const int64_t m_est = n / 48 + 1;  // (expected n / 64)
    constexpr int BR_PASSES = 3;  // 24 leading key bits place the bracket ends finely enough (2^-15 / 2^-12 relative)
    const K low_mask = (K)(((K)1 << (8 * (KeyT<T>::passes - BR_PASSES))) - 1);
    uint32_t* d_rbs = reinterpret_cast<uint32_t*>(ws->d_small + 4);
    // both ends of every bin's bracket in ONE selection over the sample (SEL_BRACKET_DUAL: states [0, nb) low, [nb, 2 nb) high)
    rc = select_enqueue<T>(ctx, static_cast<const T*>(ws->s_vals), nb == 1 ? nullptr : ws->s_bins, ws->s_cap, m_est, d_flags + 0, nb, scratch,
                           SEL_BRACKET_DUAL, nullptr, BR_PASSES, false);
    if (rc) return rc;
    hipLaunchKernelGGL((bracket_finish_kernel<K>), dim3(1), dim3(64), 0, ctx->stream, d_st, nb, (int)(ctx->selection_mode == 2), low_mask, d_klo, d_khi, d_rbs);
    XD_HIP_CHECK(ctx, hipGetLastError());
    // 3. the one pass over the data
    int copies = (32 * 1024) / (nb * 12);
    copies = copies < 1 ? 1 : (copies > 32 ? 32 : copies);
    const size_t lds = lds_stage + (size_t)nb * 2 * sizeof(K) + (size_t)((3 * nb) | 1) * 4 * (size_t)copies + 8 + lds_src;
    rc = set_big_lds(ctx, bracket_pass_kernel<T, Src>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((bracket_pass_kernel<T, Src>), dim3(grid_for(ctx, n, HIST_THREADS * SEL_TILE, 2)), dim3(HIST_THREADS), lds, ctx->stream, src,
                       n, nb, copies, d_klo, d_khi, d_cnt, static_cast<T*>(ws->c_vals), ws->c_bins, d_flags, ws->c_cap);
    XD_HIP_CHECK(ctx, hipGetLastError());
    if constexpr (Src::HAS_LEAN) {
        rc = src.launch_lean(ctx, n, nb, d_klo, d_khi, d_cnt, static_cast<T*>(ws->c_vals), ws->c_bins, d_flags, ws->c_cap);
        if (rc) return rc;
    }
    if (passed) *passed = true;
    rc = xd_allreduce_device(ctx, d_cnt, 3 * (int64_t)nb, XDEMHIP_RED_SUM_U64);
    if (rc) return rc;
    rc = xd_allreduce_device(ctx, d_ctr + 2, 1, XDEMHIP_RED_SUM_U64);  // overflow anywhere -> everybody falls back
    if (rc) return rc;
    hipLaunchKernelGGL(bracket_given_kernel, dim3(nbb), dim3(64), 0, ctx->stream, d_cnt, nb, d_given, d_flags);
    XD_HIP_CHECK(ctx, hipGetLastError());
    // 4. exact selection among the candidates (a few percent of the data)
    rc = select_enqueue<T>(ctx, static_cast<const T*>(ws->c_vals), ws->c_bins, ws->c_cap, n / 32 + 1, d_flags + 1, nb, scratch, SEL_GIVEN, d_given,
                           0, true, d_klo, d_rbs);
    if (rc) return rc;
    std::vector<uint64_t> cnt(3 * nb);
    std::vector<K> klo(nb);
    uint64_t h_ctr[5];
    { const int rc_ = xd_d2h(ctx, cnt.data(), d_cnt, 8 * 3 * (size_t)nb); if (rc_) return rc_; }
    { const int rc_ = xd_d2h(ctx, klo.data(), d_klo, sizeof(K) * nb); if (rc_) return rc_; }
    { const int rc_ = xd_d2h(ctx, h_ctr, d_ctr, 40); if (rc_) return rc_; }
    if (before_sync) { rc = (*before_sync)(); if (rc) return rc; }
    rc = select_fetch<T>(ctx, scratch, nb, out);  // (synchronises the stream)
    if (rc) return rc;
    if (h_ctr[2] != 0 || h_ctr[3] != 0) return XDEMHIP_OK;  // buffer overflow or a bracket missed (*done stays false)
    const int rbs = (int)(uint32_t)h_ctr[4];
    for (int b = 0; b < nb; ++b) {
        const uint64_t total = cnt[b], lt = cnt[nb + b];
        if (total == 0) { out[b].st.count = 0; continue; }
        out[b].st.count = total;
        out[b].st.n_le += lt;
        out[b].st.prefix = (K)((K)(out[b].st.prefix >> rbs) + klo[b]);  // back from the rebased candidate keys
        if (out[b].succ != ~(uint64_t)0) out[b].succ = (uint64_t)(K)((K)((K)out[b].succ >> rbs) + klo[b]);
    }
    *done = true;
    return XDEMHIP_OK;
}

// Exact medians of (values, bin ids) arrays: bracketed route when it applies, plain digit passes otherwise.
template <typename T>
int run_select(xdemhip_ctx* ctx, const T* vals, const uint16_t* bins, int64_t n, int nb, unsigned char* scratch,
               std::vector<SelResult<typename KeyT<T>::type>>& out, SelWorkspace* ws = nullptr) {
    bool done = false;
    ArraySource<T> src{vals, bins};
    const int rc = run_select_bracketed<T, ArraySource<T>>(ctx, src, n, nb, scratch, out, ws, &done);
    if (rc || done) return rc;
    return run_select_core<T>(ctx, vals, bins, n, nb, scratch, out, SEL_MEDIAN, nullptr);
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
