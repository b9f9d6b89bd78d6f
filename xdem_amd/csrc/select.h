// select.h -- exact per-bin order statistics (medians) on the GPU by MSD radix selection.
//
// Medians are not sum-reducible, so the Nuth-Kaab aspect bins (np.nanmedian per bin,
// xdem/coreg/affine.py:2404, xdem/spatialstats.py:143-157), the global vertical shift (np.nanmedian,
// affine.py:504) and Dowd's variogram estimator (median of |differences| per lag) are all computed the same
// way: the values are mapped to order-preserving unsigned keys and the k-th smallest key of every bin is
// found digit by digit (8 bits per pass, most significant first).  One pass = every element whose key still
// matches its bin's prefix increments an LDS histogram [bin][256] (ds_add_u32), the block flushes non-zero
// counters to a global uint64 table, and a one-workgroup kernel advances each bin's (prefix, rank).  All
// counting is integer, hence exact, order-independent and all-reducible across GPUs.  For an even count the
// upper median is either the same value (duplicates) or the smallest key above it, found in one extra pass.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace xd {

constexpr int SEL_RADIX = 256;

// order-preserving key of an IEEE float / double (NaN never gets here)
__device__ __forceinline__ uint32_t key_of(float v) {
    const uint32_t b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ uint64_t key_of(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return b ^ ((b >> 63) ? 0xFFFFFFFFFFFFFFFFull : 0x8000000000000000ull);
}
__host__ __device__ inline float val_of(uint32_t k) {
    const uint32_t b = (k >> 31) ? (k ^ 0x80000000u) : ~k;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}
__host__ __device__ inline double val_of(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k ^ 0x8000000000000000ull) : ~k;
    double f;
    __builtin_memcpy(&f, &b, 8);
    return f;
}
template <typename T> struct KeyT;
template <> struct KeyT<float> { typedef uint32_t type; static constexpr int passes = 4; };
template <> struct KeyT<double> { typedef uint64_t type; static constexpr int passes = 8; };

// Per-bin selection state kept on the device.
template <typename K> struct SelState {
    K prefix;          // key bits fixed so far (high digits)
    uint64_t rank;     // remaining 0-based rank inside the current prefix group
    uint64_t count;    // elements in the bin
    uint64_t n_le;     // elements <= selected key (valid after the last pass)
    uint64_t group;    // elements that share the digits fixed so far with the selected key
};

// Advance every bin by one digit: hist[bin][256] holds the counts of the current digit among the elements that
// match the bin's prefix.  First pass (digit == top) also fixes count and the target rank (lower median).
// One wave64 per bin: lane l owns buckets 4l .. 4l+3; wave prefix sums locate the bucket that holds the rank.
// Target rank fixed at the first pass: SEL_MEDIAN = lower median (count-1)/2; SEL_BRACKET_LO / _HI = the median rank of a
// SAMPLE moved down / up by sel_bracket_halfwidth(count) (select_run.h: bracketed selection); SEL_GIVEN = given[bin]
// (all-ones: skip the bin).
enum { SEL_MEDIAN = 0, SEL_BRACKET_LO = 1, SEL_BRACKET_HI = 2, SEL_GIVEN = 3, SEL_BRACKET_LO_WIDE = 4, SEL_BRACKET_HI_WIDE = 5,
       SEL_BRACKET_DUAL = 6 /* states [0, nb) = low ends, [nb, 2 nb) = high ends of the bins' brackets, selected together */ };

// Half width (in sample ranks) of the bracket around the sample median that holds the population median with
// overwhelming probability: 6 standard deviations of the rank (0.5 sqrt(m_eff)) for an effective sample size of
// m / 8 -- the sample is made of whole 8-element lines (select_run.h: SEL_LINE), fully correlated lines being the worst
// case -- plus slack.
__host__ __device__ inline uint64_t sel_bracket_halfwidth(uint64_t m) {
    return (uint64_t)(3.0 * sqrt(8.0 * (double)m)) + 32;
}
// Narrowed half widths (callers that MEASURE how well their samples centre: the one-pass Nuth-Kaab step).  `narrow` is a code: below
// 16 a right shift of the rule's width (0 = the full rule, 1 = half, 2 = a quarter), 16 + q = q sixteenths of it (q = 1 .. 16).
constexpr uint32_t SEL_NARROW_16THS = 16;
__host__ __device__ inline uint64_t sel_narrowed(uint64_t h_full, uint32_t narrow) {
    const uint64_t w = h_full - 32;
    return (narrow < SEL_NARROW_16THS ? (w >> narrow) : ((w * (uint64_t)(narrow - SEL_NARROW_16THS)) >> 4)) + 32;
}
__host__ __device__ inline double sel_narrow_unit(uint32_t narrow) {   // the fraction of the full rule a code stands for
    return narrow < SEL_NARROW_16THS ? 1.0 / (double)(1u << narrow) : (double)(narrow - SEL_NARROW_16THS) / 16.0;
}
// Same for samples of point PAIRS (variogram.hip).  The sample is a 1/64 subsample of the B points against ALL A points of a block
// (variogram.hip: unit_sample_slot): pairs that share a point are strongly dependent (values of a spatially correlated field:
// |v_a - v_b| moves with v_b for all ~9000 A points at once), so the effective sample size is about the number of distinct
// sampled B points in the class, not the number of pairs -- taken as m / 4096 (measured on SURVEY 8d's C5 input: a class of
// 8.5e7 sampled pairs behaves like ~2e4 independent draws).
// `deff` = the design effect assumed for the sample (pairs per independent draw): 4096 for the samples that pair a few B points
// with ALL A points of a tile (i < j blocks), PAIR_DEFF_SPREAD for the samples in which every point of a unit takes part in a
// few pairs only (variogram.hip: unit_sample_slot).
// PAIR_DEFF_SPREAD measured on SURVEY 8d's C5 input (tools/vario_c5_probe.py): with 16 the wanted rank lay 0.08 half widths off
// the bracket centre at worst (rms 0.026 over the 50 classes) -- the spread sample behaves like independent draws or better (every
// unit is represented: design effect < 1).  Round 4 takes 4 (half the width of 16: half the candidates the counting pass stages and
// the final selection reads, 0.44 -> 0.22 % of the pairs): a factor ~10 in variance remains for fields and geometries that correlate
// more strongly, and a miss costs one more counting pass with 16 x the design effect, then the wide rule, not the plain route
// (variogram.hip: pairs_medians_typed).
constexpr uint32_t PAIR_DEFF_WIDE = 4096, PAIR_DEFF_SPREAD = 4;
__host__ __device__ inline uint64_t sel_bracket_halfwidth_wide(uint64_t m, uint32_t deff = PAIR_DEFF_WIDE) {
    return (uint64_t)(3.0 * sqrt((double)deff * (double)m)) + 64;
}

// One wave advances state `b` by one digit (the body of select_advance_kernel; also run by the last workgroup of a histogram pass
// that advances its own states, select_run.h: hist_pass_kernel<T, true>).
// DEV: the histogram was filled by device-scope atomics of THIS launch (other workgroups): it is read with device-scope loads, which
// are performed where those atomics were; `lds_states` (optional) also receives the advanced state -- readers in the same workgroup.
template <typename K, bool DEV = false>
__device__ __forceinline__ void select_advance_body(const int b, const int lane, SelState<K>* st, uint64_t* hist, int nb, int shift, int first, int last,
                                                    int mode, const uint64_t* given, const uint32_t* rb_shift, uint32_t wide_deff, int dual_nb,
                                                    uint64_t* succ, uint32_t* need_succ, uint32_t narrow, SelState<K>* lds_states = nullptr) {
    if (b >= nb) return;
    // SEL_BRACKET_DUAL, first digit: both ends of a bin's bracket start from the same histogram -- hist_pass_kernel fills only the
    // low end's row and the low end's block sets up both states
    const bool dual_first = first && mode == SEL_BRACKET_DUAL;
    if (dual_first && b >= dual_nb) return;
    // Rebased keys ((key - lo) << s, select_run.h) have s zero bits at the bottom: a digit that lies entirely inside them is 0
    // for every element, its pass is skipped (hist_pass_kernel leaves at once) and the state moves on without a histogram.
    if (!first && rb_shift && shift + 8 <= (int)*rb_shift) {
        if (last && lane == 0 && st[b].count) st[b].n_le += st[b].group;
        return;
    }
    // the last digit that is not entirely inside the zero bits of rebased keys: every key's bits below it are zero
    const bool eff_last = last || (rb_shift && shift <= (int)*rb_shift);
    uint64_t* h = hist + (size_t)b * SEL_RADIX;
    unsigned long long c[4], mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        c[q] = DEV ? (unsigned long long)__hip_atomic_load(&h[4 * lane + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned long long)h[4 * lane + q];
        mine += c[q];
        h[4 * lane + q] = 0;   // (zeroed for the next pass)
    }
    // inclusive scan of the per-lane totals
    unsigned long long incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    const unsigned long long total = __shfl(incl, 63);
    for (int half = 0; half < (dual_first ? 2 : 1); ++half) {
    const int bs = b + half * dual_nb;   // the state this trip advances
    SelState<K> s = st[bs];
    if (first) {
        s.count = total;
        uint64_t r = total ? (total - 1) / 2 : 0;  // lower median
        const bool lo_end = mode == SEL_BRACKET_LO || (mode == SEL_BRACKET_DUAL && bs < dual_nb);
        const bool hi_end = mode == SEL_BRACKET_HI || (mode == SEL_BRACKET_DUAL && bs >= dual_nb);
        if (lo_end && total) { const uint64_t h = sel_narrowed(sel_bracket_halfwidth(total), narrow); r = r > h ? r - h : 0; }
        if (hi_end && total) { const uint64_t h = sel_narrowed(sel_bracket_halfwidth(total), narrow); r = (r + h < total) ? r + h : total - 1; }
        if (mode == SEL_BRACKET_LO_WIDE && total) { const uint64_t h = sel_bracket_halfwidth_wide(total, wide_deff); r = r > h ? r - h : 0; }
        if (mode == SEL_BRACKET_HI_WIDE && total) { const uint64_t h = sel_bracket_halfwidth_wide(total, wide_deff); r = (r + h < total) ? r + h : total - 1; }
        if (mode == SEL_GIVEN) {
            r = given[bs];
            if (r == ~(uint64_t)0 || r >= total) { s.count = 0; r = 0; }
        }
        s.rank = r;
        s.prefix = 0;
        s.n_le = 0;
        s.group = total;
    }
    if (s.count) {
        // the lane whose bucket range contains the rank (the last non-empty lane if the rank lies beyond: cannot happen
        // for a consistent histogram, kept for robustness)
        const unsigned long long excl = incl - mine;
        const bool here = (s.rank >= excl) && (s.rank < incl);
        const unsigned long long vote = __ballot(here);
        const int src = vote ? (int)__ffsll((long long)vote) - 1 : 63;
        unsigned long long cum = excl;
        int q = 0;
        if (cum + c[0] <= s.rank) { cum += c[0]; q = 1;
            if (cum + c[1] <= s.rank) { cum += c[1]; q = 2;
                if (cum + c[2] <= s.rank) { cum += c[2]; q = 3; } } }
        const unsigned long long hq = q == 0 ? c[0] : (q == 1 ? c[1] : (q == 2 ? c[2] : c[3]));
        const int dsel = __shfl(4 * lane + q, src);
        const unsigned long long cumsel = __shfl(cum, src);
        const unsigned long long hsel = __shfl(hq, src);
        s.prefix |= (K)dsel << shift;
        s.n_le += cumsel;  // elements strictly below the chosen digit group
        s.rank -= cumsel;
        s.group = hsel;
        if (last) s.n_le += hsel;  // all digits fixed: group == the selected key's duplicates
        if (eff_last && succ) {
            // Round 4: the smallest key above the selected one usually shares its leading digits -- then it is the next non-empty
            // bucket of THIS histogram (the last effective digit completes the key) and the successor pass over the elements has nothing
            // to do; only a selected key that is the largest of its group leaves the question to that pass (need_succ).
            int cand = 0x7fffffff;
#pragma unroll
            for (int q = 3; q >= 0; --q)
                if (c[q] > 0 && 4 * lane + q > dsel) cand = 4 * lane + q;
            const unsigned long long have = __ballot(cand != 0x7fffffff);
            if (have) {
                const int nxt = __shfl(cand, (int)__ffsll((long long)have) - 1);   // buckets ascend with the lane: the first lane that has one holds the smallest
                if (lane == 0) succ[bs] = (uint64_t)(K)((s.prefix & ~((K)0xFF << shift)) | ((K)nxt << shift));
            } else if (lane == 0 && need_succ) {
                *need_succ = 1u;
            }
        }
    }
    if (lane == 0) {
        st[bs] = s;
        if (lds_states) lds_states[bs] = s;
    }
    }
}

template <typename K>
__global__ __launch_bounds__(64) void select_advance_kernel(SelState<K>* st, uint64_t* hist, int nb, int shift, int first, int last,
                                                            int mode = SEL_MEDIAN, const uint64_t* given = nullptr,
                                                            const uint32_t* rb_shift = nullptr, uint32_t wide_deff = PAIR_DEFF_WIDE,
                                                            int dual_nb = 1 /* SEL_BRACKET_DUAL: states below this are low ends */,
                                                            uint64_t* succ = nullptr /* [nb]: successor keys from the last histogram */,
                                                            uint32_t* need_succ = nullptr /* raised when some bin's successor needs the scan */,
                                                            uint32_t narrow = 0 /* SEL_BRACKET_LO / _HI / _DUAL: half width >> narrow (callers that
                                                                                   measured how centred their brackets are, nuthkaab.hip) */) {
    select_advance_body<K>((int)blockIdx.x, (int)threadIdx.x, st, hist, nb, shift, first, last, mode, given, rb_shift, wide_deff, dual_nb, succ, need_succ,
                           narrow);
}

}  // namespace xd
