// variogram.hip -- pairwise lag binning of the empirical variogram on gfx950.
//
// Replaces the pairwise work the reference hands to scikit-gstat from sample_empirical_variogram
// (xdem/spatialstats.py:1091 pdist, 1247-1255 cdist): for every pair of a "pair block" (every point of set A with
// every point of set B, or all i < j inside A) take the Euclidean distance d and the absolute value difference,
// assign the lag class k with e_{k-1} <= d < e_k from explicit right edges, and reduce per class:
//   sums   -> count and sum(diff^2) (Matheron) or sum(sqrt diff) (Cressie-Hawkins)
//   hist   -> one 8-bit digit histogram of the diff keys per class (exact median for Dowd, select.h)
//   succ   -> smallest key above the selected one (upper median of an even-sized class)
// This is O(N^2) arithmetic over O(N) bytes: points are staged in LDS and re-used 256x, HBM traffic is negligible
// and the bound is VALU + LDS-atomic throughput (no MFMA: there is no contraction, every pair ends in a scatter).
//
// One workgroup = 256 points of A (one per lane, in registers) x a chunk of B streamed through LDS in tiles of 256
// (all lanes read the same B element: LDS broadcast, conflict-free).  Class lookup is a binary search over
// thresholds on the SQUARED distance that are exact images of the edges under sqrt (computed on the host), so the
// class is bit-identical to comparing the rounded float64 distance with the edges.  Accumulators live in LDS
// (ds_add_u32 / ds_add_f64 / ds_min) and are flushed once per workgroup with global atomics; counts and histograms
// are integers, hence exact and all-reducible across GPUs.
#include <math.h>
#include <type_traits>
#include <string.h>

#include <vector>
#include <thread>
#include <atomic>
#include <algorithm>
#include <sched.h>

#include <stdio.h>
#include <time.h>
#include <stdlib.h>

#include "common.h"
#include "select.h"
#include "select_run.h"

namespace xd {

#pragma clang fp contract(off)

// spread pair sample (every-a-with-every-b blocks): B slots per lane and tile -> SPREAD_SLOTS / 256 of the pairs of every unit.
// (4 = 1/64 was the first setting; the brackets scale with 1 / sqrt(sample), the three sampled passes with the sample: 2 slots
// cost 0.1 % more candidates and save a third of the sampled passes' time on SURVEY 8d's C5)
#ifndef XD_SPREAD_SLOTS
#define XD_SPREAD_SLOTS 2
#endif
constexpr int SPREAD_SLOTS = XD_SPREAD_SLOTS;
constexpr int PT = 256;      // B points per LDS tile; A points per workgroup = NT (256 for sums / succ, 1024 for histograms)
// (a lane's SPREAD_SLOTS slots lie PT / SPREAD_SLOTS apart, so one wave covers SPREAD_SLOTS of the tile's four 64-slot groups:
// consecutive waves start in different groups and the waves of a workgroup together cover every B point of the tile -- with one
// common start the 2-slot sample used half of the B points of a unit and its ranks scattered three times as far)
static_assert(SPREAD_SLOTS == 1 || SPREAD_SLOTS == 2 || SPREAD_SLOTS == 4, "64-slot groups");
__device__ __forceinline__ int spread_wave_offset(int tid) { return 64 * ((tid >> 6) % (4 / SPREAD_SLOTS)); }
constexpr int BCHUNK = 4096;  // B points per workgroup
constexpr int LUT_N = 512;    // cells (1/8 binade of d^2 each) of the class lookup table
constexpr int LUT_G = 256;    // GRID: cells of float(d2) from LUT_G0 on (d2 = 1 is cell 1016, d2 < 2^32 ends at 1272): 64 dwords = one per LDS bank
constexpr int LUT_G0 = 1016;
constexpr int LUT_STEPS = 1;  // a cell holds at most ONE threshold (host-checked), flagged in the entry's low bit
constexpr int NCOPY = 32;     // privatised accumulator copies (copy = lane % 32 -> one LDS bank per copy)
constexpr int HIST_BINS_PER_SWEEP_K = 128;   // (= HIST_BINS_PER_SWEEP below: the most lag classes a bracketed selection takes)

enum { OP_SUMS_SQ = 0, OP_SUMS_SQRT = 1, OP_HIST = 2, OP_SUCC = 3, OP_BRACKET = 4 };

template <typename T> struct PairArgs {
    const double *ax, *ay, *bx, *by;
    const T *av, *bv;
    const int64_t *a_off, *b_off;  // per block [nblk + 1]
    const int64_t* wg_off;         // workgroups before each block [nblk + 1] (for this kernel's NT)
    int nblk, nb, pdist;
    const double* thr;  // nb thresholds on d^2
    const uint8_t* lut;  // [LUT_N] (class lower bound << 1 | cell holds a threshold) per 1/8 binade of d^2; nullptr: binary search
    int lut_emin;        // (biased exponent << 3 | top 3 mantissa bits) of LUT entry 0
    // integer-lattice path (GRID kernels): points as packed int16 lattice indexes (x | y << 16), thresholds on the integer
    // squared lattice distance, class lower bound per 1/8 binade of float(d2)
    const uint32_t *a_xy, *b_xy;
    const uint32_t* thr_i;   // [nb]
    const uint8_t* lut_i;    // [LUT_N]
    int lut_i_emin;
    // outputs / state
    double* sums;                  // [nb]
    unsigned long long* counts;    // [nb]
    unsigned long long* hist;      // [nb][256]
    const typename KeyT<T>::type* prefix;  // [nb] selection prefix (hist) or selected key (succ)
    unsigned long long* succ;              // [nb] 8-byte slots, all-ones = none
    int shift, first, bin0, nbs;   // hist digit, first pass flag, LDS sweep window [bin0, bin0 + nbs)
    int64_t wg_base, wg_end;       // units [wg_base, wg_end) belong to this launch (a pass over very many tiles takes several launches)
    int sample;                    // OP_HIST: only the pseudo-randomly chosen 1/64 of the (A tile x B tile) units
    // OP_HIST, dual: TWO selection states per class advance in the same pass (both ends of a bracket): a second prefix array and
    // a second histogram; an element is counted under each prefix it matches
    int dual;
    const typename KeyT<T>::type* prefix2;
    unsigned long long* hist2;
    int has_nan;                   // some value is NaN (unknown for device-resident inputs: assumed): full tiles keep the NaN test
    // OP_BRACKET (bracketed selection, select_run.h): keys in [prefix[k], khi[k]] are candidates
    const typename KeyT<T>::type* khi;
    unsigned long long* cnt3;      // [3][nb]: pairs per class at or above the bracket's low end, below it, inside the bracket
    void* cand_v;                  // candidate |dv| (T; WIDE kernels: the EXACT float64 difference of two float32 values)
    uint16_t* cand_b;              // and their class
    unsigned long long* cand_ctr;  // [0] count, [1] overflow flag
    long long cand_cap;
    int runs;                      // OP_BRACKET on the lattice: the points come in Morton order (xdemhip_pairs_link_sorted) -- run-length counting
};

// Sampled digit passes (bracketed selection): EVERY (A tile x B tile) unit contributes exactly 1/64 of its pairs.  (Round 2
// sampled whole units with probability 1/64: for values of a spatially correlated field the class distributions differ from
// block to block, the number of sampled units per block was binomial (3 +- 1.7), and the mixture weights -- hence the sample
// medians -- were too noisy for any affordable bracket.)
//  * every-a-with-every-b blocks: thread t (A point t of the tile) takes SPREAD_SLOTS B slots (h + lane + u 256 / SPREAD_SLOTS) mod
//    256, h hashed from the unit: the sampled pairs of a unit use each of its 1024 A points SPREAD_SLOTS times and each of its
//    256 B points 4 SPREAD_SLOTS times.  EVERY tile takes part: sampling every 4th tile with 16 slots per lane (a quarter of the tile loads and barriers
//    for the same pairs) was measured -- the wanted ranks then sat 0.63 half widths off the bracket centres instead of 0.04: B
//    points arrive ring by ring, a tile is one distance class of one block, and the class mixtures over the blocks get noisy.  (The first form of this round took 4 B slots for ALL A points: 1024 pairs per sampled B point, and since |v_a - v_b|
//    of a correlated field moves with v_b for all of them at once, a class of 8.5e7 sampled pairs behaved like 2e4 independent
//    draws -- brackets 8 x wider, 5 % of all pairs candidates.)
//  * i < j blocks keep 4 adjacent slots [4 h, 4 h + 4) against all A points (the diagonal test needs the uniform slot index) and
//    the wide brackets that go with them.
__device__ __forceinline__ int unit_sample_slot(int64_t wg, int tile) {
    return (int)(((uint64_t)(wg * 16 + tile) * 0x9E3779B97F4A7C15ull) >> 58);   // 0 .. 63
}

template <typename K> __device__ __forceinline__ void lds_min(K* p, K v);
template <> __device__ __forceinline__ void lds_min<uint32_t>(uint32_t* p, uint32_t v) { atomicMin(p, v); }
template <> __device__ __forceinline__ void lds_min<uint64_t>(uint64_t* p, uint64_t v) {
    atomicMin(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
}

// |dv| >= 0: its IEEE bits are already order-preserving; shifting the (always zero) sign bit out gives the first radix
// digit the full 8 exponent bits instead of sign + 7 (twice the spread of the LDS counters it hits).
__device__ __forceinline__ uint32_t key_abs(float v) { return __float_as_uint(v) << 1; }
__device__ __forceinline__ uint64_t key_abs(double v) { return (uint64_t)__double_as_longlong(v) << 1; }

typedef short v2s16 __attribute__((ext_vector_type(2)));

// r = (mask bit of this lane) ? b : a, the lane mask held in an SGPR pair (a divergent `bool` carried around a loop is
// legalised into a 0/1 VGPR and costs a compare per use; an explicit 64-bit mask stays scalar)
__device__ __forceinline__ uint32_t select_by_mask(uint32_t a, uint32_t b, unsigned long long mask) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
__device__ __forceinline__ float select_by_mask(float a, float b, unsigned long long mask) {
    return __uint_as_float(select_by_mask(__float_as_uint(a), __float_as_uint(b), mask));
}
__device__ __forceinline__ double select_by_mask(double a, double b, unsigned long long mask) {
    const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    const uint32_t lo = select_by_mask((uint32_t)ua, (uint32_t)ub, mask), hi = select_by_mask((uint32_t)(ua >> 32), (uint32_t)(ub >> 32), mask);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <typename T> struct alignas(8) FzEnds { T lo, hi; };   // a class's bracket as values: one 8 / 16-byte LDS read

// n + (mask bit of this lane): one v_addc with the lane mask as the carry-in
__device__ __forceinline__ uint32_t add_mask_bit(uint32_t n, unsigned long long mask) {
    uint32_t r;
    unsigned long long carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=s"(carry_out) : "v"(n), "s"(mask));
    return r;
}

// GRID (implies FAST): the points lie on an integer lattice (raster pixel centres -- the reference's cdist / pdist samplers
// draw raster pixels, xdem/spatialstats.py:1413-1416): coordinates are packed int16 lattice indexes, the squared lattice
// distance of a pair is ONE v_pk_sub_i16 + ONE v_dot2_i32_i16, exact in 32 bits, and the class follows from integer
// thresholds that are the exact pre-images of the float64 thresholds (host: make_grid) -- identical classes, a third fewer
// instructions per pair than the float64 coordinates (5 float64 operations + a 64-bit compare).
// WIDE (float32 values only; context option "vario_diff" = 1 on float32 inputs, round 5): the pass classifies every pair by its
// float32 difference fl(a - b) exactly as the default kernel does -- same instructions in the hot loop --, but a CANDIDATE leaves
// as the exact float64 difference |double(a) - double(b)| (the B value is read again from the LDS tile when, rarely, a pair is a
// candidate).  Rounding is monotone -- fl(d1) < fl(d2) implies d1 < d2 -- so the integer counts of the float32 classification
// are exact counts below / inside / above the bracket for the exact differences too, and the order statistic selected among
// the float64 candidates at rank (r - below) IS the exact float64 one: scikit-gstat's pdist-style float64 differences at the
// speed of the float32 kernels (the float64 kernels spend 60 ms on this pass against 38).
template <typename T, bool WIDE> struct CandT { typedef T type; };
template <> struct CandT<float, true> { typedef double type; };
template <typename T, int OP, bool FAST, int NT, bool GRID = false, bool WIDE = false>
// (1024-thread workgroups: two of them per CU -- 8 waves per SIMD -- need at most 64 VGPRs: second launch-bound = waves per SIMD; float32 values only, the float64
// kernels do not fit that budget without spilling)
__global__ __launch_bounds__(NT, (NT == 1024 && sizeof(T) == 4) ? 8 : 1) void pairs_kernel(const PairArgs<T> a) {
    typedef typename KeyT<T>::type K;
    typedef typename CandT<T, WIDE>::type TC;   // what a candidate is stored as
    static_assert(!WIDE || (sizeof(T) == 4 && OP == OP_BRACKET), "WIDE: float32 values, counting pass only");
    TC* const cand_out = static_cast<TC*>(a.cand_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* s_bx = reinterpret_cast<double*>(smem);
    double* s_by = s_bx + PT;
    double* s_thr = s_by + PT;                       // nb (+ LUT_STEPS + 1 entries of +inf padding)
    K* s_pref = reinterpret_cast<K*>(s_thr + a.nb + LUT_STEPS + 1);  // nb selection prefixes / selected keys / bracket lows (8-byte slots)
    K* s_khi = reinterpret_cast<K*>(reinterpret_cast<uint64_t*>(s_pref) + a.nb);  // nb bracket highs (8-byte slots)
    T* s_bv = reinterpret_cast<T*>(reinterpret_cast<uint64_t*>(s_khi) + a.nb);  // PT
    uint32_t* s_bxy = reinterpret_cast<uint32_t*>(s_bv + PT);                    // PT packed lattice indexes (GRID)
    uint32_t* s_thr_i = s_bxy + PT;                                              // nb + 2 integer thresholds (GRID), padded with all-ones
    unsigned char* acc = reinterpret_cast<unsigned char*>(s_thr_i + ((a.nb + 2 + 1) & ~1));  // (8-byte aligned)
    // OP_SUMS: NCOPY privatised copies per class, copy = lane % 32: lanes of a wave that hit the same class land on
    // different banks (at most 2 lanes per address) instead of serialising 64-way on one LDS word
    // per class a 384-byte record: NCOPY float64 sums (256 B = all 64 LDS banks, one copy per lane % 32: conflict-free), then
    // NCOPY uint32 counts.  Record nb collects the pairs beyond the last edge and is never read back, so full tiles need no
    // class test.  Both atomics of a pair address the record with the same `l * 384` term.
    // (An interleaved 16-byte (sum, count) slot per copy would save an address computation but puts four lanes of every
    // 32-lane group on one bank pair: measured slower.)
    constexpr int REC = NCOPY * 12;
    unsigned char* s_rec = acc;
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(acc);            // OP_HIST: nbs * 256
    K* s_min = reinterpret_cast<K*>(acc);                           // OP_SUCC: nb keys
    unsigned long long* s_c3 = reinterpret_cast<unsigned long long*>(acc);   // OP_BRACKET: [nb + 1][NCOPY] packed counters, then the staging buffer
    // (WIDE: 8-byte candidates in half as many slots -- the staging buffer keeps its size, and with it two workgroups per CU)
    constexpr int STAGE_CAP = WIDE ? SEL_STAGE_CAP / 2 : SEL_STAGE_CAP;
    BlockStage<TC, STAGE_CAP> stage;
    // OP_BRACKET: one PACKED 64-bit counter per class and copy -- bits 0..20 pairs of the class, 21..41 those at or above the
    // bracket's low end, 42..62 those above its high end (a workgroup gives a copy at most 1024 x 4096 / 32 = 2^17 pairs): ONE
    // ds_add_u64 of (1 | ge << 21 | gt << 42) per pair, its value two selects on the compare masks, its address one shift-add
    // (three separate planes needed a plane index, a multiply-add and two more adds).  Class nb = spare: pairs that are no pairs
    // -- beyond the last edge, diagonal, NaN -- count there, so the hot loop needs no predicate.  Then the bracket ends
    // interleaved {low, high} per class (one 8 / 16-byte read per pair; the spare class holds {all-ones, 0}: always "below",
    // never a candidate), then the staging buffer
    K* s_lh = reinterpret_cast<K*>(s_c3 + (a.nb + 1) * NCOPY);
    // ... and the same ends as VALUES {low, high} per class for the run-length loop (float compares of |dv|; an end whose bits are
    // no number compares false with everything, which is what an unreachable end means)
    T* s_lhf = reinterpret_cast<T*>(s_lh + 2 * (a.nb + 1));
    // ... and per class the number of candidates the run-length loop staged (tallied when the wave hands its pending candidates to the
    // staging buffer: the runs count only total and >= low end per pair)
    uint32_t* s_in = reinterpret_cast<uint32_t*>(s_lhf + 2 * (a.nb + 1));
    if (OP == OP_BRACKET) {
        stage.v = reinterpret_cast<TC*>(s_in + ((a.nb + 2) & ~1));
        stage.b = reinterpret_cast<uint16_t*>(stage.v + STAGE_CAP);
        stage.base = reinterpret_cast<unsigned long long*>(stage.b + STAGE_CAP);
        stage.held = reinterpret_cast<int*>(stage.base + 1);
    }

    __shared__ uint8_t s_lut[GRID ? LUT_G : LUT_N];
    // GRID: class of a cell's lower bound and the one threshold above it fused into one 8-byte entry (one LDS read per pair
    // instead of a byte read and a dependent threshold read -- the Matheron pass is bound by the LDS array)
    __shared__ uint2 s_lut2[GRID ? LUT_G : 1];
    // sums on the lattice (round 4): {class of the cell's lower bound c, threshold c, threshold c - 1 (0 for c = 0), threshold c + 1}: the
    // class of a d^2 AND the d^2 interval of that class from one 16-byte read (see the run-length loop)
    __shared__ uint4 s_lut4[(GRID && (OP == OP_SUMS_SQ || OP == OP_SUMS_SQRT || OP == OP_BRACKET)) ? LUT_G : 1];
    // run-length counting pass, float32 values (round 5): everything a lane loads when its run enters class l -- the class's d^2
    // interval {low, width} and its bracket {low, high} as float bits -- in ONE 16-byte record (the interval used to be rebuilt from the
    // cell's table entry with two selects and a subtract next to the 8-byte read of the bracket)
#if defined(XD_NO_CLSREC)   // (measurement build: the round-4 form)
    constexpr bool CLSREC = false;
#else
    constexpr bool CLSREC = GRID && OP == OP_BRACKET && sizeof(T) == 4;
#endif
    __shared__ uint4 s_clsrec[CLSREC ? HIST_BINS_PER_SWEEP_K + 2 : 1];
    const int tid = threadIdx.x;
    if (FAST)  // (NT may be smaller than the table: round 1 loaded only its first NT entries -- wrong classes beyond 32 binades of d^2)
        for (int k = tid; k < (GRID ? LUT_G : LUT_N); k += NT) s_lut[k] = GRID ? a.lut_i[k] : a.lut[k];
    for (int k = tid; k < a.nb + LUT_STEPS + 1; k += NT) s_thr[k] = k < a.nb ? a.thr[k] : (double)INFINITY;
    if (GRID) {
        for (int k = tid; k < a.nb + 2; k += NT) s_thr_i[k] = k < a.nb ? a.thr_i[k] : 0xFFFFFFFFu;
        for (int k = tid; k < LUT_G; k += NT) {
            const uint32_t c = a.lut_i[k];
            s_lut2[k] = make_uint2(c, c < (uint32_t)a.nb ? a.thr_i[c] : 0xFFFFFFFFu);
            if (OP == OP_SUMS_SQ || OP == OP_SUMS_SQRT || OP == OP_BRACKET)
                s_lut4[k] = make_uint4(c, c < (uint32_t)a.nb ? a.thr_i[c] : 0xFFFFFFFFu, (c > 0 && c - 1 < (uint32_t)a.nb) ? a.thr_i[c - 1] : (c > 0 ? 0xFFFFFFFFu : 0u),
                                       c + 1 < (uint32_t)a.nb ? a.thr_i[c + 1] : 0xFFFFFFFFu);
        }
    }
    if (OP == OP_SUMS_SQ || OP == OP_SUMS_SQRT)
        for (int k = tid; k < (a.nb + 1) * REC / 4; k += NT) reinterpret_cast<uint32_t*>(s_rec)[k] = 0u;
    unsigned char* const s_sum_cp = s_rec + (tid & (NCOPY - 1)) * 8;              // this lane's privatised copies
    unsigned char* const s_cnt_cp = s_rec + NCOPY * 8 + (tid & (NCOPY - 1)) * 4;
    auto rec_add = [&](int l, double term) {
        const int off = l * REC;
        atomicAdd(reinterpret_cast<double*>(s_sum_cp + off), term);
        atomicAdd(reinterpret_cast<uint32_t*>(s_cnt_cp + off), 1u);
    };
    // (the first digit's histogram is the same for both states: only table A is filled, the host copies it)
    const bool dual = OP == OP_HIST && a.dual && !a.first;
    if (OP == OP_HIST)
        for (int k = tid; k < a.nbs * SEL_RADIX * (dual ? 2 : 1); k += NT) s_hist[k] = 0;
    if (dual)
        for (int k = tid; k < a.nb; k += NT) s_khi[k] = a.prefix2[k];
    if (OP == OP_SUCC)
        for (int k = tid; k < a.nb; k += NT) s_min[k] = ~(K)0;
    if ((OP == OP_HIST && !a.first) || OP == OP_SUCC || OP == OP_BRACKET)
        for (int k = tid; k < a.nb; k += NT) s_pref[k] = a.prefix[k];
    if (OP == OP_BRACKET) {
        for (int k = tid; k < a.nb; k += NT) s_khi[k] = a.khi[k];
        // (the fast loop compares the raw bits of |d| -- key_abs(d) >> 1 -- so the ends are stored halved: key >= lo <=> bits >=
        // ceil(lo / 2), key <= hi <=> bits <= floor(hi / 2); the spare class gets an unreachable low end)
        for (int k = tid; k <= a.nb; k += NT) {
            const K lo = k < a.nb ? a.prefix[k] : ~(K)0, hi = k < a.nb ? a.khi[k] : (K)0;
            const K lo_h = (K)((lo >> 1) + (lo & 1)), hi_h = (K)(hi >> 1);
            s_lh[2 * k] = lo_h;
            s_lh[2 * k + 1] = hi_h;
            // (lo_h can be 2^(bits - 1) -- the bit pattern of -0 -- when the low end is unreachable: NaN, never >= anything)
            K lo_bits = lo_h, hi_bits = hi_h;
            if (lo_h >> (8 * sizeof(K) - 1)) lo_bits = (K)~(K)0 >> 1;
            T lo_f, hi_f;
            __builtin_memcpy(&lo_f, &lo_bits, sizeof(T));
            __builtin_memcpy(&hi_f, &hi_bits, sizeof(T));
            s_lhf[2 * k] = lo_f;
            s_lhf[2 * k + 1] = hi_f;
            if constexpr (CLSREC) {
                if (k <= HIST_BINS_PER_SWEEP_K) {
                    const uint32_t lo_d2 = k > 0 ? a.thr_i[k - 1] : 0u, hi_d2 = k < a.nb ? a.thr_i[k] : 0xFFFFFFFFu;
                    uint32_t lb, hb;
                    __builtin_memcpy(&lb, &lo_f, 4);
                    __builtin_memcpy(&hb, &hi_f, 4);
                    s_clsrec[k] = make_uint4(lo_d2, hi_d2 - lo_d2, lb, hb);
                }
            }
        }
        for (int k = tid; k < (a.nb + 1) * NCOPY; k += NT) s_c3[k] = 0ull;
        for (int k = tid; k < a.nb + 1; k += NT) s_in[k] = 0u;
        if (tid == 0) *stage.held = 0;
    }

    const int nb = a.nb;
    const K himask = (OP == OP_HIST && !a.first) ? (K)(~(K)0 << (a.shift + 8)) : (K)0;
    int run_l = a.nb;          // spare class: flushing the empty initial run adds (0, 0) to the record nobody reads
    double run_s = 0.0;
    uint32_t run_start = 0;    // position (count of plain-tile pairs so far, the same for every lane) at which the lane's run began
    uint32_t run_pos = 0;      // ... and the current position: a run's length is their difference, no per-pair counter
    uint32_t run_lo = 1u, run_w = 0u;   // GRID: d^2 interval of the run's class (empty: the first pair looks its class up)
    uint32_t run_ge = 0u;               // OP_BRACKET runs: pairs of the run at or above the bracket's low end
    T run_blo = (T)0, run_bhi = (T)0;   // ... and the bracket of the run's class
    // One workgroup = one (A tile x B chunk) unit, except in the SAMPLED digit passes: there a unit is 16 tile loads and 16 k pairs,
    // while zeroing and flushing the [classes][256] LDS tables costs ~25 k LDS writes and thousands of global atomics -- a
    // resident set of workgroups therefore walks over all units (grid-stride) and flushes once.
    __syncthreads();   // the tables above are complete (the barrier-free sampled path below reads them at once)
    for (int64_t wg = a.wg_base + blockIdx.x; wg < a.wg_end; wg += gridDim.x) {
        // which block / A tile / B chunk is this unit?
        int lo = 0, hi = a.nblk;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.wg_off[mid] <= wg) lo = mid; else hi = mid;
        }
        const int r = lo;
        const int64_t a0 = a.a_off[r], na = a.a_off[r + 1] - a0;
        const int64_t b0 = a.pdist ? a0 : a.b_off[r], nbp = a.pdist ? na : a.b_off[r + 1] - b0;
        const int64_t nchunk = (nbp + BCHUNK - 1) / BCHUNK;
        const int64_t local = wg - a.wg_off[r];
        const int64_t ta = local / nchunk, cb = local - ta * nchunk;
        const int64_t ia = ta * NT + tid;  // index inside the block's A set
        const int64_t jb0 = cb * BCHUNK, jb1 = (jb0 + BCHUNK < nbp) ? jb0 + BCHUNK : nbp;
        const bool have_a = ia < na;
        const bool skip_wg = a.pdist && (jb1 <= ta * NT + 1);  // whole chunk at or below the diagonal: no i < j pair
        double px = 0.0, py = 0.0;
        T pv = 0;
        uint32_t pxy = 0;
        if (have_a) {
            if (GRID) pxy = a.a_xy[a0 + ia];
            else { px = a.ax[a0 + ia]; py = a.ay[a0 + ia]; }
            pv = a.av[a0 + ia];
        }
        const double* gbx = a.pdist ? a.ax : a.bx;
        const double* gby = a.pdist ? a.ay : a.by;
        const uint32_t* gbxy = a.pdist ? a.a_xy : a.b_xy;
        const T* gbv = a.pdist ? a.av : a.bv;

        // OP_BRACKET, fast path: a lane keeps at most ONE candidate pending in registers; every 8 pairs the wave moves its pending
        // candidates into the staging buffer with one LDS reservation (the per-pair form -- ballot, leader atomic with return and
        // its round trip whenever any lane of the wave had a candidate, i.e. for half of all wave-pairs -- was most of the 22
        // vector instructions per pair this pass spent beyond the class lookup).  A second candidate while one is pending (a few
        // per thousand lane-octets) is appended directly.
        // OP_SUMS, full tiles: RUN-LENGTH accumulation.  The point sets are uploaded in Morton order (PairSet: neighbouring slots are
        // neighbouring points), so consecutive B points of a tile mostly fall into the same lag class of a lane's A point: the lane
        // keeps the running sum and count of its current class in registers and touches the LDS record -- two atomics -- only when
        // the class changes.  (Round 2 paid both atomics for every pair and the LDS array was busy for the whole kernel.)
        TC pend_v = (TC)0;
        uint32_t pend_l = 0;
        // the value a candidate is stored as: the pair's |dv| as classified, or (WIDE) the exact float64 difference with tile slot j
        auto cand_value = [&](T abs_dv, int j) -> TC {
            if constexpr (WIDE) return (TC)__builtin_fabs((double)pv - (double)s_bv[j]);
            else return abs_dv;
        };
        unsigned long long pend_m = 0;  // lanes with a pending candidate (wave-uniform scalar)
        // Candidates that find the staging buffer full go straight to the global candidate buffer, one global atomic per wave and
        // call.  (Values of a spatially correlated field cluster: all 262144 pairs of a tile -- 1024 neighbouring A points x 256 B
        // points -- can fall into a class's bracket at once, far more than the 8192 staging slots a tile may fill; round 2's "flag
        // an overflow and redo everything with plain passes" made the C5 input of SURVEY 8d five times slower than uniform noise.)
        auto spill = [&](bool mine, TC v, uint32_t l) {   // every lane of the wave calls this
            const unsigned long long ov = __builtin_amdgcn_ballot_w64(mine);
            if (!ov) return;
            const int lane = tid & 63;
            const int leader = __ffsll((long long)ov) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(&a.cand_ctr[0], (unsigned long long)__popcll(ov));
            base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(base >> 32), leader) << 32) |
                   (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)base, leader);
            if (mine) {
                const unsigned long long pos = base + (unsigned long long)__builtin_amdgcn_mbcnt_hi((uint32_t)(ov >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ov, 0u));
                if ((long long)pos < a.cand_cap) { cand_out[pos] = v; a.cand_b[pos] = (uint16_t)l; }
                else a.cand_ctr[1] = 1ull;
            }
        };
        auto flush_pending = [&](bool tally = false) {
            const unsigned long long m = pend_m;
            if (m) {  // (wave-uniform)
                const int lane = tid & 63;
                const int leader = __ffsll((long long)m) - 1;
                int pos0 = 0;
                if (lane == leader) pos0 = atomicAdd(stage.held, __popcll(m));
                pos0 = __builtin_amdgcn_readlane(pos0, leader);
                const bool has = (m >> lane) & 1ull;
                const int pos = pos0 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (has && pos < STAGE_CAP) { stage.v[pos] = pend_v; stage.b[pos] = (uint16_t)pend_l; }
                if (tally && has) atomicAdd(&s_in[pend_l], 1u);
                if (__builtin_expect(pos0 + __popcll(m) > STAGE_CAP, 0)) spill(has && pos >= STAGE_CAP, pend_v, pend_l);   // (wave-uniform test)
                pend_m = 0;
            }
        };
        // Sampled pass over every-a-with-every-b blocks on the integer lattice, straight from global memory: a lane needs 4 B points
        // of each tile (slots h + lane + 64 u: consecutive over the lanes, so the loads coalesce), which is not worth a 256-point
        // LDS tile and two barriers -- that form left a workgroup waiting for one tile load at a time, 2.6 us per tile and 3.4 ms
        // per pass.  Four tiles per trip: 16 independent pairs (32 loads) per lane in flight, no barrier, waves run on their own.
        if constexpr (GRID && OP == OP_HIST) {
            if (a.sample && !a.pdist) {   // (uniform over the launch)
                const int lane = tid & 63;
                for (int64_t j0 = jb0; j0 < jb1; j0 += 4 * PT) {
                    uint32_t bxy[4 * SPREAD_SLOTS];
                    T bvv[4 * SPREAD_SLOTS];
                    bool okk[4 * SPREAD_SLOTS];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int64_t jt = j0 + (int64_t)t * PT;
                        const int h = unit_sample_slot(wg, (int)((jt - jb0) / PT));
                        const int64_t cnt_t = jb1 - jt;   // (<= 0: no such tile)
#pragma unroll
                        for (int u = 0; u < SPREAD_SLOTS; ++u) {
                            const int sj = (h + lane + spread_wave_offset(tid) + (PT / SPREAD_SLOTS) * u) & (PT - 1);
                            const bool ok = have_a && sj < cnt_t;
                            const int64_t src = b0 + (ok ? jt + sj : jb0);
                            okk[SPREAD_SLOTS * t + u] = ok;
                            bxy[SPREAD_SLOTS * t + u] = gbxy[src];
                            bvv[SPREAD_SLOTS * t + u] = gbv[src];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4 * SPREAD_SLOTS; ++q) {
                        const v2s16 d = __builtin_bit_cast(v2s16, pxy) - __builtin_bit_cast(v2s16, bxy[q]);
                        uint32_t d2;
                        asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(d2) : "v"(d));
                        const uint32_t cell = __float_as_uint((float)d2) >> 20;
                        const uint2 e = s_lut2[(cell > LUT_G0 ? cell : (uint32_t)LUT_G0) - LUT_G0];
                        const int lu = (int)e.x + ((e.y <= d2) ? 1 : 0);
                        T dd = pv - bvv[q];
                        dd = dd < 0 ? -dd : dd;
                        const int lb = lu - a.bin0;
                        if (okk[q] && lu < nb && dd == dd && lb >= 0 && lb < a.nbs) {
                            const K key = key_abs(dd);
                            const int digit = (int)((key >> a.shift) & 0xFF);
                            if (a.first || (key & himask) == s_pref[lu]) atomicAdd(&s_hist[lb * SEL_RADIX + digit], 1u);
                            if (dual && (key & himask) == s_khi[lu]) atomicAdd(&s_hist[(a.nbs + lb) * SEL_RADIX + digit], 1u);
                        }
                    }
                }
                continue;   // next unit
            }
        }
        if (!skip_wg)
            for (int64_t j0 = jb0; j0 < jb1; j0 += PT) {
                // sampled pass: only slots [jbeg, jend) of this tile (uniform over the workgroup)
                const bool sampled = OP == OP_HIST && a.sample;
                const int jslot = sampled ? 4 * unit_sample_slot(wg, (int)((j0 - jb0) / PT)) : 0;
                if (OP == OP_BRACKET)  // (starts with the barrier the tile reload needs); flush the staged candidates when half full
                    stage.sync_and_flush_at(STAGE_CAP / 2, cand_out, a.cand_b, &a.cand_ctr[0], a.cand_cap, &a.cand_ctr[1]);
                else
                    __syncthreads();
                const int cnt = (int)((jb1 - j0) < PT ? (jb1 - j0) : PT);
                // sampled pass over every-a-with-every-b blocks: per-lane slots (see unit_sample_slot), the whole tile is loaded
                const bool spread = sampled && !a.pdist;
                const int hslot = jslot >> 2;
                if (tid < cnt) {
                    if (GRID) s_bxy[tid] = gbxy[b0 + j0 + tid];
                    else { s_bx[tid] = gbx[b0 + j0 + tid]; s_by[tid] = gby[b0 + j0 + tid]; }
                    s_bv[tid] = gbv[b0 + j0 + tid];
                }
                __syncthreads();
                if (!have_a && OP != OP_BRACKET) continue;   // (OP_BRACKET: every thread stays for the mid-tile flush barriers)
                // OP_BRACKET, one pair: counters + staged candidate (every lane of the wave must reach the append)
                auto bracket_pair = [&](bool ok, int l, T d, int j) {
                    bool cand = false;
                    if (ok) {
                        const K key = key_abs(d);
                        const int cp = tid & (NCOPY - 1);
                        // one packed counter update per pair (see s_c3)
                        const bool below = key < s_pref[l];
                        cand = !below && key <= s_khi[l];
                        atomicAdd(&s_c3[l * NCOPY + cp], 1ull | (below ? 0ull : (1ull << 21)) | ((below || cand) ? 0ull : (1ull << 42)));
                    }
                    {   // staged append (one LDS reservation per wave); what does not fit the staging buffer spills to global memory
                        const unsigned long long cm = __builtin_amdgcn_ballot_w64(cand);
                        if (cm) {
                            const int lane = tid & 63, leader = __ffsll((long long)cm) - 1;
                            int pos0 = 0;
                            if (lane == leader) pos0 = atomicAdd(stage.held, __popcll(cm));
                            pos0 = __builtin_amdgcn_readlane(pos0, leader);
                            const int pos = pos0 + __popcll(cm & ((1ull << lane) - 1ull));
                            const TC dw = cand ? cand_value(d, j) : (TC)0;
                            if (cand && pos < STAGE_CAP) { stage.v[pos] = dw; stage.b[pos] = (uint16_t)l; }
                            if (pos0 + __popcll(cm) > STAGE_CAP) spill(cand && pos >= STAGE_CAP, dw, (uint32_t)l);
                        }
                    }
                };
                // One pair: class lookup + accumulate.  `ok` folds every skip rule so the fast path stays branch-free
                // up to the (exec-masked) LDS atomics.
                auto pair = [&](int j, bool ok) {
                    const double dx = px - s_bx[j], dy = py - s_by[j];
                    const double s2 = dx * dx + dy * dy;  // not contracted: same rounding as NumPy's dx**2 + dy**2
                    int l;  // class = number of thresholds <= s2
                    if (FAST) {
                        // 1/8-binade cell of s2 -> class lower bound; only cells that contain a threshold (1 in 8 for the
                        // reference's sqrt(2)-geometric edges) need the one exact compare against it
                        int e = (int)((unsigned long long)__double_as_longlong(s2) >> 49) - a.lut_emin;
                        e = e < 0 ? 0 : (e > LUT_N - 1 ? LUT_N - 1 : e);
                        const int c = s_lut[e];
                        l = c >> 1;
                        if (c & 1) l += (s_thr[l] <= s2) ? 1 : 0;
                    } else {
                        l = 0;
                        int h = nb;
                        while (l < h) {
                            const int m = (l + h) >> 1;
                            if (s_thr[m] <= s2) l = m + 1; else h = m;
                        }
                    }
                    T d = pv - s_bv[j];
                    d = d < 0 ? -d : d;
                    ok = ok && (l < nb) && (d == d);  // beyond the last edge (maxlag) / NaN values never form a pair
                    if (OP == OP_BRACKET) { bracket_pair(ok, l, d, j); return; }
                    if (!ok) return;
                    if (OP == OP_SUMS_SQ) {
                        rec_add(l, (double)d * (double)d);
                    } else if (OP == OP_SUMS_SQRT) {
                        rec_add(l, sqrt((double)d));
                    } else if (OP == OP_HIST) {
                        const int lb = l - a.bin0;
                        if (lb < 0 || lb >= a.nbs) return;
                        const K key = key_abs(d);
                        const int digit = (int)((key >> a.shift) & 0xFF);
                        if (a.first || (key & himask) == s_pref[l]) atomicAdd(&s_hist[lb * SEL_RADIX + digit], 1u);
                        if (dual && (key & himask) == s_khi[l]) atomicAdd(&s_hist[(a.nbs + lb) * SEL_RADIX + digit], 1u);
                    } else if (OP == OP_SUCC) {
                        const K key = key_abs(d);
                        if (key > s_pref[l] && key < s_min[l]) lds_min<K>(&s_min[l], key);
                    }
                };

                // classes of the 4 pairs (this lane's A point) x (tile slots j .. j + 3) and their raw value differences, written
                // stage by stage so that the B-point reads, the table reads and the threshold reads are each issued back to back
                auto classify4 = [&](const int (&js)[4], int (&lu)[4], T (&dv)[4]) {
                    if constexpr (GRID) {
                        uint32_t d2[4];
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const v2s16 d = __builtin_bit_cast(v2s16, pxy) - __builtin_bit_cast(v2s16, s_bxy[js[u]]);
                            // (the three-operand form with the inline constant 0: the builtin picks v_dot2c, which needs a zeroed accumulator)
                            asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(d2[u]) : "v"(d));
                            dv[u] = pv - s_bv[js[u]];
                        }
                        uint2 e[4];
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            // 1/8-binade cell of float(d2) (monotone in d2); d2 = 0 (coincident points) clamps onto the first cell
                            const uint32_t cell = __float_as_uint((float)d2[u]) >> 20;
                            e[u] = s_lut2[(cell > LUT_G0 ? cell : (uint32_t)LUT_G0) - LUT_G0];
                        }
    #pragma unroll
                        for (int u = 0; u < 4; ++u) lu[u] = (int)e[u].x + ((e[u].y <= d2[u]) ? 1 : 0);
                    } else {
                        double s2[4];
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const double dx = px - s_bx[js[u]], dy = py - s_by[js[u]];
                            s2[u] = dx * dx + dy * dy;  // not contracted: same rounding as NumPy's dx**2 + dy**2
                            dv[u] = pv - s_bv[js[u]];
                        }
                        int l[4];
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            int e = (int)((unsigned long long)__double_as_longlong(s2[u]) >> 49) - a.lut_emin;
                            e = e < 0 ? 0 : (e > LUT_N - 1 ? LUT_N - 1 : e);
                            l[u] = s_lut[e] >> 1;
                        }
                        double th[4];
    #pragma unroll
                        for (int u = 0; u < 4; ++u) th[u] = s_thr[l[u]];  // next threshold above the cell's lower bound
    #pragma unroll
                        for (int u = 0; u < 4; ++u) lu[u] = l[u] + ((th[u] <= s2[u]) ? 1 : 0);  // class = number of thresholds <= s2
                    }
                };
                if (FAST) {
                    // 4 pairs per trip, written stage by stage so that the 4 B-point reads, the 4 table reads and the 4
                    // threshold reads are each issued back to back (one LDS round trip per stage instead of per pair).
                    // Tile slots beyond cnt hold stale data and are masked by `ok`.
                    const int64_t rel = ia - j0;  // pdist: only B indices j > rel pair with this lane's A point
                    // (a thread without an A point -- last A tile of a block -- pairs with nothing: every slot masked)
                    const int ia_rel = !have_a ? PT : (a.pdist ? (int)(rel < -1 ? -1 : (rel > PT ? PT : rel)) : -1);
                    // Full tiles (the bulk of the pairs): every slot is a pair -- no index, diagonal, class or NaN test and no
                    // exec masking; the class beyond the last edge lands in the spare record.
                    const bool plain_tile = (OP == OP_SUMS_SQ || OP == OP_SUMS_SQRT) && cnt == PT && !a.has_nan &&
                                            (!a.pdist || j0 >= (ta + 1) * (int64_t)NT);
                    if (plain_tile) {
                        if constexpr (GRID) {
                            // Round 4: no class lookup per pair.  A lane keeps the d^2 interval [run_lo, run_lo + run_w) of the lag class
                            // its run is in (integer thresholds: exact pre-images of the float64 edges) -- a pair that stays in the
                            // class costs one subtract and one unsigned compare instead of the table lookup (convert, cell, table
                            // read, threshold compare, class compare: 9 vector instructions); only a pair that LEAVES the class
                            // -- 5 % of the pairs of a lane, a third of the wave-pairs on SURVEY 8d's C5 geometry -- looks its
                            // class up, flushes the run (the run's length is the distance of its positions: no counter per pair)
                            // and loads the new bounds.
                            for (int j = 0; j < PT; j += 4) {
                                uint32_t d2[4];
                                T dv[4];
    #pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const v2s16 d = __builtin_bit_cast(v2s16, pxy) - __builtin_bit_cast(v2s16, s_bxy[j + u]);
                                    asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(d2[u]) : "v"(d));
                                    dv[u] = pv - s_bv[j + u];
                                }
    #pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    if (!((uint32_t)(d2[u] - run_lo) < run_w)) {   // (exec-masked; skipped by the whole wave while every lane stays in its class)
                                        const uint32_t cell = __float_as_uint((float)d2[u]) >> 20;
                                        const uint4 e = s_lut4[(cell > LUT_G0 ? cell : (uint32_t)LUT_G0) - LUT_G0];
                                        const bool up = e.y <= d2[u];   // the cell's one threshold lies at or below d^2: the class above it
                                        const int off = (int)__umul24((unsigned)run_l, (unsigned)REC);
                                        atomicAdd(reinterpret_cast<double*>(s_sum_cp + off), run_s);
                                        atomicAdd(reinterpret_cast<uint32_t*>(s_cnt_cp + off), run_pos - run_start);
                                        run_l = (int)e.x + (up ? 1 : 0);
                                        run_s = 0.0;
                                        run_start = run_pos;
                                        run_lo = up ? e.y : e.z;
                                        run_w = (up ? e.w : e.y) - run_lo;   // (beyond the last edge: thresholds padded with all ones)
                                    }
                                    const double dd = (double)dv[u];
                                    run_s = OP == OP_SUMS_SQ ? __builtin_fma(dd, dd, run_s) : run_s + sqrt(fabs(dd));
                                    run_pos += 1;   // (uniform: a scalar counter)
                                }
                            }
                            continue;
                        }
                        for (int j = 0; j < PT; j += 4) {
                            int lu[4];
                            T dv[4];
                            const int js[4] = {j, j + 1, j + 2, j + 3};
                            classify4(js, lu, dv);
    #pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const double dd = (double)dv[u];
                                if (lu[u] != run_l) {   // (exec-masked; skipped by the whole wave while every lane stays in its class)
                                    const int off = run_l * REC;
                                    atomicAdd(reinterpret_cast<double*>(s_sum_cp + off), run_s);
                                    atomicAdd(reinterpret_cast<uint32_t*>(s_cnt_cp + off), run_pos - run_start);
                                    run_l = lu[u];
                                    run_s = 0.0;
                                    run_start = run_pos;
                                }
                                run_s = OP == OP_SUMS_SQ ? __builtin_fma(dd, dd, run_s) : run_s + sqrt(fabs(dd));
                                run_pos += 1;
                            }
                        }
                        continue;
                    }
                    if constexpr (OP == OP_BRACKET && GRID) {
                        // Round 4: run-length counting.  With the points in Morton order a lane's pairs stay in one lag class for ~20
                        // pairs at a stretch: the lane keeps the class's d^2 interval and bracket in registers and counts in registers
                        // (two compares of |dv| against the bracket ends, two carry adds); the packed LDS counter gets ONE ds_add_u64
                        // per run instead of one per pair, and the class lookup and the read of the bracket ends happen once per run
                        // (19 -> 9 vector instructions and 4 -> 2 LDS operations per pair that stays in its class).
                        // (a lane without an A point -- last A tile of a block -- sits in the spare class for good: an interval nothing leaves,
                        // a low end nothing reaches)
                        if (a.runs && cnt == PT && !a.has_nan && (!a.pdist || j0 >= (ta + 1) * (int64_t)NT)) {
                            if (!have_a && run_w != 0xFFFFFFFFu) {
                                const unsigned long long inc = (unsigned long long)(run_pos - run_start) | ((unsigned long long)run_ge << 21) | ((unsigned long long)run_ge << 42);
                                atomicAdd(s_c3 + (size_t)run_l * NCOPY + (tid & (NCOPY - 1)), inc);
                                run_l = nb; run_ge = 0u; run_start = run_pos; run_lo = 0u; run_w = 0xFFFFFFFFu;
                                run_blo = s_lhf[2 * nb]; run_bhi = s_lhf[2 * nb + 1];
                            }
                            static_assert(NCOPY * 8 == 256, "counter records are 256 bytes");
                            unsigned char* const c3_mine = reinterpret_cast<unsigned char*>(s_c3 + (tid & (NCOPY - 1)));
                            // every fourth tile all lanes look their class up again: a run then holds at most 4 PT = 2^10 pairs and its
                            // counts pack with two shifts (no 64-bit arithmetic)
                            static_assert(4 * PT <= 1024, "a run's count of pairs >= the low end must fit 11 bits");
                            if ((((j0 - jb0) / PT) & 3) == 0 && have_a) run_w = 0u;
                            for (int j = 0; j < PT; j += 4) {
                                uint32_t d2[4];
                                T dv[4];
    #pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const v2s16 d = __builtin_bit_cast(v2s16, pxy) - __builtin_bit_cast(v2s16, s_bxy[j + u]);
                                    asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(d2[u]) : "v"(d));
                                    dv[u] = pv - s_bv[j + u];
                                }
    #pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    if (!((uint32_t)(d2[u] - run_lo) < run_w)) {   // (exec-masked; skipped by the whole wave while every lane stays in its class)
                                        const uint32_t cell = __float_as_uint((float)d2[u]) >> 20;
                                        const uint4 e = s_lut4[(cell > LUT_G0 ? cell : (uint32_t)LUT_G0) - LUT_G0];
                                        const bool up = e.y <= d2[u];
                                        // packed counter: pairs | >= low end | "above" (here: >= low end as well -- the candidates among
                                        // them are tallied in s_in and taken off at the end); a run holds <= 1024 pairs (see below)
                                        const uint32_t inc_lo = (run_ge << 21) | (run_pos - run_start), inc_hi = run_ge << 10;
                                        atomicAdd(reinterpret_cast<unsigned long long*>(c3_mine + ((uint32_t)run_l << 8)), ((unsigned long long)inc_hi << 32) | inc_lo);
                                        run_l = (int)e.x + (up ? 1 : 0);
                                        run_ge = 0u;
                                        run_start = run_pos;
                                        if constexpr (CLSREC) {
                                            const uint4 cr = s_clsrec[run_l];
                                            run_lo = cr.x;
                                            run_w = cr.y;
                                            __builtin_memcpy(&run_blo, &cr.z, 4);
                                            __builtin_memcpy(&run_bhi, &cr.w, 4);
                                        } else {
                                            run_lo = up ? e.y : e.z;
                                            run_w = (up ? e.w : e.y) - run_lo;
                                            const FzEnds<T> be = *reinterpret_cast<const FzEnds<T>*>(s_lhf + 2 * run_l);
                                            run_blo = be.lo;
                                            run_bhi = be.hi;
                                        }
                                    }
                                    const T ad = sizeof(T) == 4 ? (T)__builtin_fabsf((float)dv[u]) : (T)__builtin_fabs((double)dv[u]);
                                    const unsigned long long m_ge = __builtin_amdgcn_ballot_w64(ad >= run_blo);
                                    const unsigned long long in = m_ge & ~__builtin_amdgcn_ballot_w64(ad > run_bhi);   // (an empty bracket, hi < lo: nothing inside)
                                    run_ge = add_mask_bit(run_ge, m_ge);
                                    if (in != 0) {   // (wave-uniform: a fifth of the wave-pairs hold a candidate)
                                        if (__builtin_expect((in & pend_m) != 0, 0)) flush_pending(true);
                                        const unsigned long long take = in & ~pend_m;
                                        pend_v = select_by_mask(pend_v, cand_value(ad, j + u), take);
                                        pend_l = select_by_mask(pend_l, (uint32_t)run_l, take);
                                        pend_m |= take;
                                    }
                                    run_pos += 1;
                                }
                                if ((j & 28) == 28) flush_pending(true);   // every 32 pairs (a second candidate of a lane before that flushes at once)
                                if ((j & 63) == 60 && j + 4 < PT)
                                    stage.sync_and_flush_at(STAGE_CAP / 4, cand_out, a.cand_b, &a.cand_ctr[0], a.cand_cap, &a.cand_ctr[1]);
                            }
                            continue;
                        }
                    }
                    auto run4 = [&](auto plain_tag) {
                        constexpr bool PLAIN = decltype(plain_tag)::value;  // every slot of the tile is a pair: no index / diagonal / NaN tests
                        const int jbeg = spread ? 0 : jslot;
                        const int jend = spread ? 4 : (sampled ? (jslot + 4 < cnt ? jslot + 4 : cnt) : cnt);
                        const int cnt_m = cnt;   // slots at or beyond this hold no point of the tile
                        for (int j = jbeg; j < jend; j += 4) {
                            int lus[4];
                            T dv[4];
                            int js[4];
    #pragma unroll
                            for (int u = 0; u < 4; ++u) js[u] = spread ? ((hslot + (tid & 63) + spread_wave_offset(tid) + (PT / SPREAD_SLOTS) * (j + u)) & (PT - 1)) : j + u;
                            classify4(js, lus, dv);
                            if constexpr (OP == OP_BRACKET) {
                                static_assert(NCOPY * 8 == 256, "counter records are 256 bytes");
                                unsigned char* const c3_mine = reinterpret_cast<unsigned char*>(s_c3 + (tid & (NCOPY - 1)));
                                int lc[4];
                                K lo4[4], hi4[4];
                                unsigned long long in4[4];
        #pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    dv[u] = fabs(dv[u]);
                                    // everything that is no pair goes to the spare class nb (the class lookup itself never exceeds nb:
                                    // "beyond the last edge" IS class nb, so full tiles need no test at all)
                                    if constexpr (PLAIN) lc[u] = lus[u];
                                    else lc[u] = (js[u] < cnt_m && js[u] > ia_rel && dv[u] == dv[u]) ? lus[u] : nb;
                                    if constexpr (sizeof(K) == 4) {   // one 8-byte read (its constant offset folds into the instruction)
                                        const uint2 lh = *reinterpret_cast<const uint2*>(s_lh + 2 * lc[u]);
                                        lo4[u] = lh.x;
                                        hi4[u] = lh.y;
                                    } else {
                                        lo4[u] = s_lh[2 * lc[u]];
                                        hi4[u] = s_lh[2 * lc[u] + 1];
                                    }
                                }
        #pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const K bits = key_abs(dv[u]) >> 1;  // (|d| has no sign bit: its raw bits order like the keys)
                                    // counter plane = (bits >= lo) + (bits > hi): 0 below the bracket, 1 inside, 2 above -- two compares,
                                    // two carry adds, one multiply-add for the address (no selects)
                                    const bool ge_lo = bits >= lo4[u], gt_hi = bits > hi4[u];
                                    // (lane masks handled as scalars: a `bool` combined in C++ comes back as a 0/1 VGPR and a compare)
                                    const unsigned long long m_ge = __builtin_amdgcn_ballot_w64(ge_lo), m_gt = __builtin_amdgcn_ballot_w64(gt_hi);
                                    const unsigned long long m_gt2 = m_ge & m_gt;  // (an empty bracket, hi < lo: everything below or above)
                                    in4[u] = m_ge & ~m_gt;
                                    const unsigned long long inc = (unsigned long long)select_by_mask(1u, 1u | (1u << 21), m_ge) |
                                                                   ((unsigned long long)select_by_mask(0u, 1u << 10, m_gt2) << 32);
                                    atomicAdd(reinterpret_cast<unsigned long long*>(c3_mine + ((uint32_t)lc[u] << 8)), inc);
                                }
                                // Candidates (0.3 % of the pairs): nearly half of the wave-trips hold none in their 256 lane-pairs and
                                // skip this altogether (one scalar test per trip instead of two selects per pair)
                                if ((in4[0] | in4[1] | in4[2] | in4[3]) != 0) {
        #pragma unroll
                                    for (int u = 0; u < 4; ++u) {
                                        // (wave-uniform, rare: a second candidate of some lane before the next flush -- the pending ones
                                        // leave first, one LDS reservation for the wave, and the lane keeps the new one)
                                        if (__builtin_expect((in4[u] & pend_m) != 0, 0)) flush_pending();
                                        const unsigned long long take = in4[u] & ~pend_m;
                                        pend_v = select_by_mask(pend_v, cand_value(dv[u], js[u]), take);
                                        pend_l = select_by_mask(pend_l, (uint32_t)lc[u], take);
                                        pend_m |= take;
                                    }
                                }
                                if ((j & 12) == 12) flush_pending();  // every fourth stage = 16 pairs (brackets hold < 1 % of the pairs)
                                // every 64 B slots (65536 pairs of the workgroup) the staged candidates leave if the buffer is a quarter full:
                                // the brackets of spatially correlated values hold a few per cent of the pairs, a whole tile's worth
                                // would not fit (j and cnt are uniform over the workgroup: every thread meets this barrier)
                                if ((j & 63) == 60 && j + 4 < jend)
                                    stage.sync_and_flush_at(STAGE_CAP / 4, cand_out, a.cand_b, &a.cand_ctr[0], a.cand_cap, &a.cand_ctr[1]);
                                continue;
                            }
        #pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int lu = lus[u];
                                dv[u] = dv[u] < 0 ? -dv[u] : dv[u];
                                const T d = dv[u];
                                const bool ok = (PLAIN || (js[u] < cnt_m && js[u] > ia_rel && d == d)) && lu < nb && (!spread || u < SPREAD_SLOTS);   // (a trip = 4 pairs: the first SPREAD_SLOTS are the sampled ones)
                                if (OP == OP_BRACKET) {
                                    bracket_pair(ok, lu, d, js[u]);
                                } else if (ok) {
                                    if (OP == OP_SUMS_SQ) {
                                        rec_add(lu, (double)d * (double)d);
                                    } else if (OP == OP_SUMS_SQRT) {
                                        rec_add(lu, sqrt((double)d));
                                    } else if (OP == OP_HIST) {
                                        const int lb = lu - a.bin0;
                                        const K key = key_abs(d);
                                        if (lb >= 0 && lb < a.nbs) {
                                            const int digit = (int)((key >> a.shift) & 0xFF);
                                            if (a.first || (key & himask) == s_pref[lu]) atomicAdd(&s_hist[lb * SEL_RADIX + digit], 1u);
                                            if (dual && (key & himask) == s_khi[lu])
                                                atomicAdd(&s_hist[(a.nbs + lb) * SEL_RADIX + digit], 1u);
                                        }
                                    } else if (OP == OP_SUCC) {
                                        const K key = key_abs(d);
                                        if (key > s_pref[lu] && key < s_min[lu]) lds_min<K>(&s_min[lu], key);
                                    }
                                }
                            }
                        }
                    };
                    // (PLAIN needs every thread of the workgroup to hold an A point: uniform, the mid-tile barriers sit inside run4)
                    if (cnt == PT && !a.has_nan && (!a.pdist || j0 >= (ta + 1) * (int64_t)NT) && (ta + 1) * (int64_t)NT <= na) run4(std::true_type());
                    else run4(std::false_type());
                    if (OP == OP_BRACKET) flush_pending();  // (a tile's pair count need not be a multiple of 8)
                } else {
                    if (spread) {
                        for (int u = 0; u < SPREAD_SLOTS; ++u) {
                            const int sj = (hslot + (tid & 63) + spread_wave_offset(tid) + (PT / SPREAD_SLOTS) * u) & (PT - 1);
                            pair(sj, have_a && sj < cnt);
                        }
                    } else {
                        const int jend = sampled ? (jslot + 4 < cnt ? jslot + 4 : cnt) : cnt;
                        for (int j = jslot; j < jend; ++j) pair(j, have_a && (!a.pdist || (j0 + j) > ia));
                    }
                }
            }
    }
    if ((OP == OP_SUMS_SQ || OP == OP_SUMS_SQRT) && run_pos != run_start) {   // the open run of every lane
        const int off = run_l * REC;
        atomicAdd(reinterpret_cast<double*>(s_sum_cp + off), run_s);
        atomicAdd(reinterpret_cast<uint32_t*>(s_cnt_cp + off), run_pos - run_start);
    }
    if (OP == OP_BRACKET && run_pos != run_start) {   // (run-length counting: the open run of every lane)
        const unsigned long long inc = (unsigned long long)(run_pos - run_start) | ((unsigned long long)run_ge << 21) | ((unsigned long long)run_ge << 42);
        atomicAdd(s_c3 + (size_t)run_l * NCOPY + (tid & (NCOPY - 1)), inc);
    }
    __syncthreads();
    if (OP == OP_SUMS_SQ || OP == OP_SUMS_SQRT) {
        for (int k = tid; k < nb; k += NT) {
            unsigned long long c = 0;
            double sm = 0.0;
            for (int q = 0; q < NCOPY; ++q) {
                const unsigned char* r = s_rec + k * REC;
                c += reinterpret_cast<const uint32_t*>(r + NCOPY * 8)[q];
                sm += reinterpret_cast<const double*>(r)[q];
            }
            if (c) { atomicAdd(&a.counts[k], c); atomicAdd(&a.sums[k], sm); }
        }
    } else if (OP == OP_HIST) {
        for (int k = tid; k < a.nbs * SEL_RADIX; k += NT)
            if (s_hist[k]) atomicAdd(&a.hist[(size_t)a.bin0 * SEL_RADIX + k], (unsigned long long)s_hist[k]);
        if (dual)
            for (int k = tid; k < a.nbs * SEL_RADIX; k += NT) {
                const uint32_t c = s_hist[a.nbs * SEL_RADIX + k];
                if (c) atomicAdd(&a.hist2[(size_t)a.bin0 * SEL_RADIX + k], (unsigned long long)c);
            }
    } else if (OP == OP_BRACKET) {
        stage.sync_and_flush_at(0, cand_out, a.cand_b, &a.cand_ctr[0], a.cand_cap, &a.cand_ctr[1]);
        for (int k = tid; k < nb; k += NT) {
            unsigned long long above = 0, below = 0, inside = 0;
            for (int q = 0; q < NCOPY; ++q) {
                const unsigned long long c = s_c3[k * NCOPY + q];
                const unsigned long long total = c & 0x1FFFFFull, ge = (c >> 21) & 0x1FFFFFull, gt = (c >> 42) & 0x1FFFFFull;
                below += total - ge;
                inside += ge - gt;
                above += gt;
            }
            // (run-length loop: its pairs at or above the low end sit in BOTH upper fields; the candidates among them were tallied)
            inside += s_in[k];
            above -= s_in[k];
            if (above + inside) atomicAdd(&a.cnt3[k], above + inside);  // [0]: at or above the bracket's low end
            if (below) atomicAdd(&a.cnt3[nb + k], below);
            if (inside) atomicAdd(&a.cnt3[2 * nb + k], inside);
        }
    } else {
        for (int k = tid; k < nb; k += NT)
            if (s_min[k] != ~(K)0) atomicMin(&a.succ[k], (unsigned long long)s_min[k]);
    }
}

}  // namespace xd

using namespace xd;

struct xdemhip_pairs {
    xdemhip_ctx* ctx = nullptr;
    int val_dtype = XDEMHIP_F32, nblk = 0, nb = 0, pdist = 0;
    bool own = false;
    double *ax = nullptr, *ay = nullptr, *bx = nullptr, *by = nullptr;
    void *av = nullptr, *bv = nullptr;
    int64_t *a_off = nullptr, *b_off = nullptr, *wg_off = nullptr, *wg_off_big = nullptr;
    double *thr = nullptr, *sums = nullptr;
    uint8_t* lut = nullptr;  // null when the edges are too dense for the binade table
    int lut_emin = 0;
    // integer-lattice path (make_grid): packed int16 lattice indexes, integer thresholds, class table over float(d2)
    uint32_t *a_xy = nullptr, *b_xy = nullptr, *thr_i = nullptr;
    uint8_t* lut_i = nullptr;
    int lut_i_emin = 0;
    bool grid = false;
    unsigned long long *counts = nullptr, *hist = nullptr;
    void *prefix = nullptr, *succ = nullptr;
    int64_t n_wg = 0, n_wg_big = 0, n_pairs = 0;
    // bracketed selection state (xdemhip_pairs_medians)
    int sample = 0, has_nan = 1, dual = 0;
    void* prefix2 = nullptr;             // (points into the caller's scratch while a dual pass runs)
    unsigned long long* hist2 = nullptr;
    void* khi = nullptr;
    unsigned long long *cnt3 = nullptr, *cand_ctr = nullptr;
    void* cand_v = nullptr;
    uint16_t* cand_b = nullptr;
    long long cand_cap = 0;
    // the same pair set with the points of every block in Morton order (xdemhip_pairs_link_sorted; not owned): the counting pass
    // of the bracketed selection reads its points
    xdemhip_pairs* sorted = nullptr;
    // a float32 copy of a float64 set whose values are all exactly float32 numbers (xdemhip_pairs_link_shadow; not owned): the
    // bracketed exact-median route runs its passes there and takes float64 candidates (WIDE kernels)
    xdemhip_pairs* shadow = nullptr;
    // small device blocks of xdemhip_pairs_medians, kept between calls (their hipMalloc / hipFree cost ~1 ms per call)
    unsigned char* sel_small = nullptr;
    size_t sel_small_bytes = 0;
    void* sel_scratch = nullptr;
    size_t sel_scratch_bytes = 0;
    std::vector<int64_t> h_a_off, h_b_off;   // host copies of the block offsets (link check)
};

namespace {

constexpr int HIST_BINS_PER_SWEEP = 128;

template <typename T> size_t lds_bytes(int nb, int op, int nbs, bool wide = false) {
    size_t base = sizeof(double) * (2 * PT + nb + LUT_STEPS + 1) + 16 * (size_t)nb + sizeof(T) * PT + 8 + 4 * (size_t)(PT + nb + 4);
    if (op == OP_HIST) return base + (size_t)nbs * SEL_RADIX * 4;
    if (op == OP_BRACKET)
        return base + (size_t)(nb + 1) * NCOPY * 8 + 4 * (size_t)(nb + 1) * sizeof(typename KeyT<T>::type) + 4 * (size_t)(nb + 2) + 8 +
               (size_t)(wide ? SEL_STAGE_CAP / 2 : SEL_STAGE_CAP) * ((wide ? 8 : sizeof(T)) + 2) + 16;
    if (op == OP_SUCC) return base + (size_t)nb * sizeof(typename KeyT<T>::type);
    return base + (size_t)(nb + 1) * NCOPY * 12;
}

template <typename T, int OP, bool WIDE = false> int launch_pairs(xdemhip_pairs* P, int shift, int first, int bin0, int nbs) {
    xdemhip_ctx* ctx = P->ctx;
    constexpr int NT = (OP == OP_HIST || OP == OP_BRACKET) ? 1024 : 256;  // histograms: 16 waves share one 51 KB LDS table -> full occupancy
    PairArgs<T> a;
    a.ax = P->ax; a.ay = P->ay; a.bx = P->bx; a.by = P->by;
    a.av = static_cast<const T*>(P->av); a.bv = static_cast<const T*>(P->bv);
    a.a_off = P->a_off; a.b_off = P->b_off; a.wg_off = (NT == 1024) ? P->wg_off_big : P->wg_off;
    a.nblk = P->nblk; a.nb = P->nb; a.pdist = P->pdist; a.thr = P->thr; a.lut = P->lut; a.lut_emin = P->lut_emin;
    a.a_xy = P->a_xy; a.b_xy = P->b_xy; a.thr_i = P->thr_i; a.lut_i = P->lut_i; a.lut_i_emin = P->lut_i_emin;
    a.sums = P->sums; a.counts = P->counts; a.hist = P->hist;
    a.prefix = static_cast<const typename KeyT<T>::type*>(P->prefix);
    a.succ = static_cast<unsigned long long*>(P->succ);
    a.shift = shift; a.first = first; a.bin0 = bin0; a.nbs = nbs;
    a.sample = P->sample;
    a.dual = (OP == OP_HIST) ? P->dual : 0;
    a.prefix2 = static_cast<const typename KeyT<T>::type*>(P->prefix2);
    a.hist2 = P->hist2;
    a.has_nan = P->has_nan;
    a.khi = static_cast<const typename KeyT<T>::type*>(P->khi);
    a.cnt3 = P->cnt3; a.cand_v = P->cand_v; a.cand_b = P->cand_b; a.cand_ctr = P->cand_ctr; a.cand_cap = P->cand_cap;
    a.runs = 0;
    if (OP == OP_BRACKET && P->sorted && ctx->vario_runs != 0) {
        // same blocks, same offsets, other slot order: counts and candidates are the same multiset
        const xdemhip_pairs* S = P->sorted;
        a.ax = S->ax; a.ay = S->ay; a.bx = S->bx; a.by = S->by;
        a.av = static_cast<const T*>(S->av); a.bv = static_cast<const T*>(S->bv);
        a.a_xy = S->a_xy; a.b_xy = S->b_xy;
        a.runs = 1;
    }
    const size_t lds = lds_bytes<T>(P->nb, OP, (OP == OP_HIST && P->dual) ? 2 * nbs : nbs, WIDE);
    const int64_t n_wg = (NT == 1024) ? P->n_wg_big : P->n_wg;
    // HIP dispatches carry the TOTAL work-item count of a dimension in 32 bits (a larger grid x block product is silently
    // truncated): a pass over more workgroups than 2^31 / NT goes out as several launches, each told where it starts.
    const int64_t per_launch = ctx->pairs_launch_cap > 0 ? (int64_t)ctx->pairs_launch_cap : ((int64_t)1 << 31) / NT;
    const bool grid = P->grid && ctx->vario_grid != 0;
    if (grid) { if (int rc = set_big_lds(ctx, pairs_kernel<T, OP, true, NT, true, WIDE>, lds)) return rc; }
    else if (P->lut) { if (int rc = set_big_lds(ctx, pairs_kernel<T, OP, true, NT, false, WIDE>, lds)) return rc; }
    else { if (int rc = set_big_lds(ctx, pairs_kernel<T, OP, false, NT, false, WIDE>, lds)) return rc; }
    for (int64_t w0 = 0; w0 < n_wg; w0 += per_launch) {
        const int64_t nw = (n_wg - w0) < per_launch ? (n_wg - w0) : per_launch;
        a.wg_base = w0;
        a.wg_end = w0 + nw;
        // sampled digit passes: resident workgroups walking over the units (see the kernel); everything else one unit each
        const int64_t resident = (int64_t)ctx->num_cu * (lds > 80 * 1024 ? 1 : 2);
        const unsigned nlaunch = (unsigned)((OP == OP_HIST && P->sample && nw > resident) ? resident : nw);
        if (grid) hipLaunchKernelGGL((pairs_kernel<T, OP, true, NT, true, WIDE>), dim3(nlaunch), dim3(NT), lds, ctx->stream, a);
        else if (P->lut) hipLaunchKernelGGL((pairs_kernel<T, OP, true, NT, false, WIDE>), dim3(nlaunch), dim3(NT), lds, ctx->stream, a);
        else hipLaunchKernelGGL((pairs_kernel<T, OP, false, NT, false, WIDE>), dim3(nlaunch), dim3(NT), lds, ctx->stream, a);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    return XDEMHIP_OK;
}

template <int OP> int launch_any(xdemhip_pairs* P, int shift = 0, int first = 0, int bin0 = 0, int nbs = 0) {
    return P->val_dtype == XDEMHIP_F32 ? launch_pairs<float, OP>(P, shift, first, bin0, nbs)
                                       : launch_pairs<double, OP>(P, shift, first, bin0, nbs);
}

// smallest double t with sqrt(t) >= e (round-to-nearest sqrt is monotone): d >= e  <=>  d^2-sum >= t
double sq_threshold(double e) {
    if (!(e > 0)) return 0.0;
    double t = e * e;
    while (sqrt(t) >= e) t = nextafter(t, 0.0);
    while (sqrt(t) < e) t = nextafter(t, INFINITY);
    return t;
}

// right-closed classes (context option "vario_edge" = 1): smallest double t with sqrt(t) > e
double sq_threshold_strict(double e) {
    if (!(e > 0)) return nextafter(0.0, INFINITY);
    double t = e * e;
    while (sqrt(t) > e) t = nextafter(t, 0.0);
    while (!(sqrt(t) > e)) t = nextafter(t, INFINITY);
    return t;
}

// 1/8-binade cell of float(d2) as the GRID kernels compute it (round-to-nearest conversion: monotone in d2)
inline int cell_of_u32(uint32_t d2) {
    const float f = (float)d2;
    uint32_t b;
    memcpy(&b, &f, 4);
    return (int)(b >> 20);
}

// Integer-lattice analysis of a pair set (host coordinates).  Succeeds when every coordinate is x0 + g * i (same g for x and
// y, g = G * 2^-k with an integer G) with lattice indexes i < 32768, and when the float64 kernels' arithmetic is exact on
// such points -- (g dx)^2 + (g dy)^2 < 2^53 -- so that their squared distance IS g^2 * d2 with d2 = dx^2 + dy^2 the integer
// squared lattice distance.  Then class(d2) = #{k : T_k <= d2}, T_k = the smallest integer with g^2 T_k >= thr_k, is
// identical to the float64 classification, pair by pair.
struct GridInfo {
    std::vector<uint32_t> a_xy, b_xy, thr_i;
    std::vector<uint8_t> lut;
    int emin = 0;
};
// (host threads of the scans below: the cores this process may use, at most 16; XDEM_HOST_THREADS overrides -- the same rule as
//  xdem_amd/spatialstats.py: _host_threads)
static int host_threads() {
    if (const char* e = getenv("XDEM_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) return v; }
    cpu_set_t set;
    int n = 1;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    return n < 1 ? 1 : (n > 16 ? 16 : n);
}
// f(first, last, thread) over [0, n) cut into one contiguous piece per thread
template <class F> static void host_chunks(int64_t n, int threads, F f) {
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, (n + (1 << 16) - 1) >> 16));   // (pieces of at least 64 K elements)
    if (threads == 1) { f((int64_t)0, n, 0); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back([=]() { f(n * t / threads, n * (t + 1) / threads, t); });
    for (auto& th : pool) th.join();
}

bool make_grid(const double* ax, const double* ay, int64_t na, const double* bx, const double* by, int64_t nbt, const std::vector<double>& thr,
               GridInfo& out) {
    // The scans run over every coordinate of the set (6e8 values for SURVEY 8d's C5 reading A): each is a reduction -- all-of, min, gcd,
    // max -- done in pieces on the host's threads and combined.
    const int T = host_threads();
    const double* arr[4] = {ax, ay, bx, by};
    const int64_t len[4] = {na, na, bx ? nbt : 0, by ? nbt : 0};
    if (na + nbt == 0) return false;
    // 1. a power-of-two scale that makes every coordinate an integer below 2^52
    int k = -1;
    for (int kk = 0; kk <= 20 && k < 0; ++kk) {
        const double sc = ldexp(1.0, kk);
        std::atomic<int> bad(0);
        for (int q = 0; q < 4; ++q)
            host_chunks(len[q], T, [&, q](int64_t i0, int64_t i1, int) {
                const double* v = arr[q];
                bool ok = true;
                for (int64_t i = i0; i < i1 && ok; ++i) { const double w = v[i] * sc; ok = std::isfinite(v[i]) && fabs(w) < 4.5e15 && w == floor(w); }
                if (!ok) bad.store(1);
            });
        if (!bad.load()) k = kk;
    }
    if (k < 0) return false;
    const double sc = ldexp(1.0, k);
    double mn[2] = {INFINITY, INFINITY};
    for (int q = 0; q < 4; ++q) {
        std::vector<double> part((size_t)T, INFINITY);
        host_chunks(len[q], T, [&, q](int64_t i0, int64_t i1, int t) {
            const double* v = arr[q];
            double m = INFINITY;
            for (int64_t i = i0; i < i1; ++i) { const double w = v[i] * sc; m = w < m ? w : m; }
            part[(size_t)t] = m;
        });
        for (double m : part) mn[q & 1] = m < mn[q & 1] ? m : mn[q & 1];
    }
    // 2. lattice constant: gcd of all offsets from the minima; 3. index range
    auto gcd = [](unsigned long long a_, unsigned long long b_) { while (b_) { const unsigned long long t = a_ % b_; a_ = b_; b_ = t; } return a_; };
    unsigned long long G = 0;
    for (int q = 0; q < 4; ++q) {
        std::vector<unsigned long long> part((size_t)T, 0ull);
        host_chunks(len[q], T, [&, q](int64_t i0, int64_t i1, int t) {
            const double* v = arr[q];
            const double m0 = mn[q & 1];
            unsigned long long g = 0;
            for (int64_t i = i0; i < i1; ++i) {
                const unsigned long long u = (unsigned long long)(v[i] * sc - m0);
                if (g != 1 && (g == 0 || u % g != 0)) g = gcd(g, u);   // (most offsets are multiples of the running gcd: one modulo each)
            }
            part[(size_t)t] = g;
        });
        for (unsigned long long g : part) G = gcd(G, g);
    }
    if (G == 0) G = 1;
    unsigned long long imax = 0;
    for (int q = 0; q < 4; ++q) {
        std::vector<unsigned long long> part((size_t)T, 0ull);
        host_chunks(len[q], T, [&, q](int64_t i0, int64_t i1, int t) {
            const double* v = arr[q];
            const double m0 = mn[q & 1];
            unsigned long long m = 0;
            for (int64_t i = i0; i < i1; ++i) { const unsigned long long u = (unsigned long long)(v[i] * sc - m0) / G; m = u > m ? u : m; }
            part[(size_t)t] = m;
        });
        for (unsigned long long m : part) imax = m > imax ? m : imax;
    }
    if (imax > 32767ull) return false;
    if ((long double)G * (long double)imax >= 67108864.0L) return false;  // (G dx)^2 + (G dy)^2 < 2^53
    auto pack = [&](const double* x, const double* y, int64_t n, std::vector<uint32_t>& o) {
        o.resize((size_t)n);
        host_chunks(n, T, [&](int64_t i0, int64_t i1, int) {
            for (int64_t i = i0; i < i1; ++i) {
                const uint32_t ix = (uint32_t)((unsigned long long)(x[i] * sc - mn[0]) / G), iy = (uint32_t)((unsigned long long)(y[i] * sc - mn[1]) / G);
                o[(size_t)i] = ix | (iy << 16);
            }
        });
    };
    pack(ax, ay, na, out.a_xy);
    if (bx) pack(bx, by, nbt, out.b_xy);
    // 4. integer thresholds: smallest n with g2 * n >= thr, g2 = G^2 * 2^-2k (g2 * n is exact in float64 for n <= 2 imax^2)
    const double g2 = ldexp((double)G * (double)G, -2 * k);
    const unsigned long long nmax = 2ull * imax * imax;
    out.thr_i.resize(thr.size());
    for (size_t q = 0; q < thr.size(); ++q) {
        const long double est = (long double)thr[q] / (long double)g2;
        unsigned long long n = est >= 4.0e9L ? 0xFFFFFFFFull : (unsigned long long)ceill(est);
        if (n <= nmax + 2) {  // refine with exact products (n inside the exactly representable range)
            while (n > 0 && g2 * (double)(n - 1) >= thr[q]) --n;
            while (g2 * (double)n < thr[q]) ++n;
        } else if (n > 0xFFFFFFFFull) {
            n = 0xFFFFFFFFull;
        }
        out.thr_i[q] = (uint32_t)(n > 0xFFFFFFFFull ? 0xFFFFFFFFull : n);
    }
    // 5. class table over the cells of float(d2), indexed directly by the cell: entry = classes whose threshold lies at or below
    // the cell's first d2; a cell may hold at most one more threshold (checked), which the kernel resolves with one integer compare
    out.emin = 0;
    out.lut.assign(LUT_G, 0);
    auto first_in_cell = [&](int c) -> unsigned long long {  // smallest d2 with cell(d2) >= c (2^32 if none)
        unsigned long long lo = 0, hi = 0x100000000ull;
        while (lo < hi) {
            const unsigned long long mid = (lo + hi) >> 1;
            if (cell_of_u32((uint32_t)mid) >= c) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    if (thr.size() > 255 || cell_of_u32(nmax > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nmax) >= LUT_G0 + LUT_G) return false;
    for (int i = 0; i < LUT_G; ++i) {
        // (entry 0 also serves d2 = 0: coincident points belong to the class of the smallest distances)
        const unsigned long long lo = i == 0 ? 0 : first_in_cell(LUT_G0 + i), hi = first_in_cell(LUT_G0 + i + 1);
        if (lo >= 0x100000000ull || lo == hi) { out.lut[i] = (uint8_t)(i ? out.lut[i - 1] : 0); continue; }  // no d2 maps here
        int below = 0, inside = 0;
        for (size_t q = 0; q < thr.size(); ++q) {
            below += (unsigned long long)out.thr_i[q] <= lo;
            inside += ((unsigned long long)out.thr_i[q] > lo && (unsigned long long)out.thr_i[q] < hi);
        }
        if (inside > 1) return false;
        out.lut[i] = (uint8_t)below;
    }
    return true;
}

}  // namespace

extern "C" {

void xdemhip_pairs_destroy(xdemhip_pairs* P) {
    if (!P) return;
    (void)hipSetDevice(P->ctx->device);
    if (P->own) {
        void* b[] = {P->ax, P->ay, P->bx, P->by, P->av, P->bv};
        for (void* p : b) if (p) (void)hipFree(p);
    }
    void* b2[] = {P->a_off, P->b_off, P->wg_off, P->wg_off_big, P->thr, P->sums, P->counts, P->hist, P->hist2, P->prefix, P->succ, P->lut,
                  P->a_xy, P->b_xy, P->thr_i, P->lut_i};
    for (void* p : b2) if (p) (void)hipFree(p);
    if (P->cand_v) (void)hipFree(P->cand_v);   // candidate buffers of the bracketed selection (kept between calls)
    if (P->cand_b) (void)hipFree(P->cand_b);
    if (P->sel_small) (void)hipFree(P->sel_small);
    if (P->sel_scratch) (void)hipFree(P->sel_scratch);
    delete P;
}

int xdemhip_pairs_create(xdemhip_ctx* ctx, int n_blocks, const int64_t* a_off, const double* ax, const double* ay, const void* av,
                         const int64_t* b_off, const double* bx, const double* by, const void* bv, int val_dtype,
                         const double* right_edges, int n_bins, int memspace, xdemhip_pairs** out, int64_t* n_pairs) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (!out || n_blocks <= 0 || !a_off || !ax || !ay || !av || !right_edges) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    if (n_bins < 1 || n_bins > 1024) return xd_fail(ctx, XDEMHIP_EINVAL, "n_bins out of range (1..1024)");
    if (val_dtype != XDEMHIP_F32 && val_dtype != XDEMHIP_F64) return xd_fail(ctx, XDEMHIP_EINVAL, "values must be float32 or float64");
    const bool pd = (b_off == nullptr);
    if (!pd && (!bx || !by || !bv)) return xd_fail(ctx, XDEMHIP_EINVAL, "null B arrays");
    for (int k = 1; k < n_bins; ++k)
        if (!(right_edges[k] > right_edges[k - 1])) return xd_fail(ctx, XDEMHIP_EINVAL, "right_edges must be strictly increasing");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // context option "vario_diff" = 1: |dv| in float64 whatever the value dtype (SciPy's pdist / cdist, which scikit-gstat builds
    // on, widen to float64 first): float32 values are widened here, every pass then runs the float64 instantiation
    std::vector<double> wide_a, wide_b;
    if (ctx->vario_diff && val_dtype == XDEMHIP_F32 && memspace == XDEMHIP_HOST) {
        const int64_t na0 = a_off[n_blocks], nb0 = pd ? 0 : b_off[n_blocks];
        wide_a.resize((size_t)na0);
        for (int64_t i = 0; i < na0; ++i) wide_a[(size_t)i] = (double)static_cast<const float*>(av)[i];
        av = wide_a.data();
        if (!pd) {
            wide_b.resize((size_t)nb0);
            for (int64_t i = 0; i < nb0; ++i) wide_b[(size_t)i] = (double)static_cast<const float*>(bv)[i];
            bv = wide_b.data();
        }
        val_dtype = XDEMHIP_F64;
    }
    xdemhip_pairs* P = new xdemhip_pairs();
    P->ctx = ctx; P->val_dtype = val_dtype; P->nblk = n_blocks; P->nb = n_bins; P->pdist = pd;
    const size_t es = val_dtype == XDEMHIP_F32 ? 4 : 8;
    const int64_t na = a_off[n_blocks], nbt = pd ? 0 : b_off[n_blocks];
    std::vector<int64_t> wg(n_blocks + 1, 0), wgb(n_blocks + 1, 0);
    int64_t pairs = 0;
    for (int r = 0; r < n_blocks; ++r) {
        const int64_t a = a_off[r + 1] - a_off[r], b = pd ? a : b_off[r + 1] - b_off[r];
        if (a < 0 || b < 0) { delete P; return xd_fail(ctx, XDEMHIP_EINVAL, "offsets must be non-decreasing"); }
        wg[r + 1] = wg[r] + ((a + 255) / 256) * ((b + BCHUNK - 1) / BCHUNK);
        wgb[r + 1] = wgb[r] + ((a + 1023) / 1024) * ((b + BCHUNK - 1) / BCHUNK);
        pairs += pd ? a * (a - 1) / 2 : a * b;
    }
    P->n_wg = wg[n_blocks];
    P->n_wg_big = wgb[n_blocks];
    P->n_pairs = pairs;
    P->h_a_off.assign(a_off, a_off + n_blocks + 1);
    if (!pd) P->h_b_off.assign(b_off, b_off + n_blocks + 1);
    std::vector<double> thr(n_bins);
    // context option "vario_edge": 0 = classes [e_{k-1}, e_k) (default), 1 = (e_{k-1}, e_k]
    for (int k = 0; k < n_bins; ++k) thr[k] = ctx->vario_edge ? sq_threshold_strict(right_edges[k]) : sq_threshold(right_edges[k]);
    // Class lookup table over d^2: cell i covers [lo_i, lo_{i+1}) with lo_i the double whose top 15 bits (biased
    // exponent, first 3 mantissa bits) are emin + i; entry = (number of thresholds <= lo_i) << 1 | (a threshold lies
    // inside the cell).  Used only if no cell holds more than one threshold and there are < 128 classes (true for the
    // reference's sqrt(2)-geometric edges: one threshold per binade = per 8 cells); otherwise binary search.
    std::vector<uint8_t> lut(LUT_N, 0);
    int emin = 0;
    bool lut_ok = n_bins <= 127;
    if (lut_ok) {
        uint64_t bits;
        memcpy(&bits, &thr[0], 8);
        emin = (int)(bits >> 49) - 1;  // cell 0 lies strictly below every threshold: smaller d^2 clamp onto it
        if (emin < 8) emin = 8;
        auto cell_lo = [&](int i) { uint64_t b = (uint64_t)(emin + i) << 49; double v; memcpy(&v, &b, 8); return v; };
        for (int i = 0; i < LUT_N; ++i) {
            const double lo = cell_lo(i), hi2 = cell_lo(i + 1);
            int below = 0, inside = 0;
            for (int k = 0; k < n_bins; ++k) { below += thr[k] <= lo; inside += (thr[k] > lo && (i == LUT_N - 1 || thr[k] < hi2)); }
            lut[i] = (uint8_t)((below << 1) | (inside ? 1 : 0));
            if (inside > 1) lut_ok = false;  // (the last cell also serves every larger d^2)
        }
        if (!(thr[0] > cell_lo(0))) lut_ok = false;
    }
    P->lut_emin = emin;
    auto fail = [&](int code, const char* m) { xdemhip_pairs_destroy(P); return xd_fail(ctx, code, m); };
#define XD_ALLOC(ptr, bytes) if (hipMalloc(reinterpret_cast<void**>(&(ptr)), (bytes) ? (bytes) : 8) != hipSuccess) return fail(XDEMHIP_ENOMEM, "hipMalloc failed")
    XD_ALLOC(P->a_off, sizeof(int64_t) * (n_blocks + 1));
    XD_ALLOC(P->wg_off, sizeof(int64_t) * (n_blocks + 1));
    XD_ALLOC(P->wg_off_big, sizeof(int64_t) * (n_blocks + 1));
    XD_ALLOC(P->thr, sizeof(double) * n_bins);
    XD_ALLOC(P->sums, sizeof(double) * n_bins);
    XD_ALLOC(P->counts, 8 * n_bins);
    XD_ALLOC(P->hist, 8 * (size_t)n_bins * SEL_RADIX);
    XD_ALLOC(P->hist2, 8 * (size_t)n_bins * SEL_RADIX);
    XD_ALLOC(P->prefix, 8 * n_bins);
    XD_ALLOC(P->succ, 8 * n_bins);
    if (!pd) XD_ALLOC(P->b_off, sizeof(int64_t) * (n_blocks + 1));
    (void)hipMemcpyAsync(P->a_off, a_off, sizeof(int64_t) * (n_blocks + 1), hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(P->wg_off, wg.data(), sizeof(int64_t) * (n_blocks + 1), hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(P->wg_off_big, wgb.data(), sizeof(int64_t) * (n_blocks + 1), hipMemcpyHostToDevice, ctx->stream);
    (void)hipMemcpyAsync(P->thr, thr.data(), sizeof(double) * n_bins, hipMemcpyHostToDevice, ctx->stream);
    if (lut_ok) {
        XD_ALLOC(P->lut, LUT_N);
        (void)hipMemcpyAsync(P->lut, lut.data(), LUT_N, hipMemcpyHostToDevice, ctx->stream);
    }
    if (!pd) (void)hipMemcpyAsync(P->b_off, b_off, sizeof(int64_t) * (n_blocks + 1), hipMemcpyHostToDevice, ctx->stream);
    if (memspace == XDEMHIP_HOST) {
        // NaN values never pair (the kernel tests for them); knowing that there are none lets full tiles skip the test
        auto any_nan = [&](const void* v, int64_t n) {
            if (val_dtype == XDEMHIP_F32) { const float* f = static_cast<const float*>(v); for (int64_t i = 0; i < n; ++i) if (f[i] != f[i]) return true; }
            else { const double* f = static_cast<const double*>(v); for (int64_t i = 0; i < n; ++i) if (f[i] != f[i]) return true; }
            return false;
        };
        P->has_nan = (any_nan(av, na) || (!pd && any_nan(bv, nbt))) ? 1 : 0;
        P->own = true;
        XD_ALLOC(P->ax, 8 * na); XD_ALLOC(P->ay, 8 * na); XD_ALLOC(P->av, es * na);
        (void)hipMemcpyAsync(P->ax, ax, 8 * na, hipMemcpyHostToDevice, ctx->stream);
        (void)hipMemcpyAsync(P->ay, ay, 8 * na, hipMemcpyHostToDevice, ctx->stream);
        (void)hipMemcpyAsync(P->av, av, es * na, hipMemcpyHostToDevice, ctx->stream);
        if (!pd) {
            XD_ALLOC(P->bx, 8 * nbt); XD_ALLOC(P->by, 8 * nbt); XD_ALLOC(P->bv, es * nbt);
            (void)hipMemcpyAsync(P->bx, bx, 8 * nbt, hipMemcpyHostToDevice, ctx->stream);
            (void)hipMemcpyAsync(P->by, by, 8 * nbt, hipMemcpyHostToDevice, ctx->stream);
            (void)hipMemcpyAsync(P->bv, bv, es * nbt, hipMemcpyHostToDevice, ctx->stream);
        }
        // raster-sampled points (the reference's samplers draw pixels): integer-lattice kernels
        GridInfo gi;
        if (lut_ok && ctx->vario_grid != 0 && make_grid(ax, ay, na, pd ? nullptr : bx, pd ? nullptr : by, nbt, thr, gi)) {
            XD_ALLOC(P->a_xy, 4 * na);
            XD_ALLOC(P->thr_i, 4 * n_bins);
            XD_ALLOC(P->lut_i, LUT_G);
            XD_HIP_CHECK(ctx, hipMemcpy(P->a_xy, gi.a_xy.data(), 4 * (size_t)na, hipMemcpyHostToDevice));
            XD_HIP_CHECK(ctx, hipMemcpy(P->thr_i, gi.thr_i.data(), 4 * (size_t)n_bins, hipMemcpyHostToDevice));
            XD_HIP_CHECK(ctx, hipMemcpy(P->lut_i, gi.lut.data(), LUT_G, hipMemcpyHostToDevice));
            if (!pd) {
                XD_ALLOC(P->b_xy, 4 * nbt);
                XD_HIP_CHECK(ctx, hipMemcpy(P->b_xy, gi.b_xy.data(), 4 * (size_t)nbt, hipMemcpyHostToDevice));
            }
            P->lut_i_emin = gi.emin;
            P->grid = true;
        }
    } else {
        P->ax = const_cast<double*>(ax); P->ay = const_cast<double*>(ay); P->av = const_cast<void*>(av);
        P->bx = const_cast<double*>(bx); P->by = const_cast<double*>(by); P->bv = const_cast<void*>(bv);
    }
#undef XD_ALLOC
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(XDEMHIP_EHIP, "upload failed");
    if (n_pairs) *n_pairs = pairs;
    *out = P;
    return XDEMHIP_OK;
}

int xdemhip_pairs_link_sorted(xdemhip_pairs* P, xdemhip_pairs* sorted) {
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (sorted == P) sorted = nullptr;
    if (sorted) {
        if (sorted->ctx != ctx || sorted->val_dtype != P->val_dtype || sorted->nblk != P->nblk || sorted->nb != P->nb ||
            sorted->pdist != P->pdist || sorted->grid != P->grid || sorted->h_a_off != P->h_a_off || sorted->h_b_off != P->h_b_off ||
            sorted->has_nan != P->has_nan)
            return xd_fail(ctx, XDEMHIP_EINVAL, "pairs_link_sorted: the two sets must hold the same blocks (sizes, dtype, edges, context)");
        std::vector<double> t0(P->nb), t1(P->nb);
        XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        XD_HIP_CHECK(ctx, hipMemcpy(t0.data(), P->thr, 8 * (size_t)P->nb, hipMemcpyDeviceToHost));
        XD_HIP_CHECK(ctx, hipMemcpy(t1.data(), sorted->thr, 8 * (size_t)P->nb, hipMemcpyDeviceToHost));
        if (memcmp(t0.data(), t1.data(), 8 * (size_t)P->nb) != 0)
            return xd_fail(ctx, XDEMHIP_EINVAL, "pairs_link_sorted: the two sets must hold the same lag edges");
    }
    P->sorted = sorted;
    return XDEMHIP_OK;
}

int xdemhip_pairs_sums(xdemhip_pairs* P, int kind, double* sums, int64_t* counts) {
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (!sums || !counts || (kind != 0 && kind != 1)) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XD_HIP_CHECK(ctx, hipMemsetAsync(P->sums, 0, 8 * P->nb, ctx->stream));
    XD_HIP_CHECK(ctx, hipMemsetAsync(P->counts, 0, 8 * P->nb, ctx->stream));
    int rc = XDEMHIP_OK;
    if (P->n_wg > 0) {
        XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
        rc = kind == 0 ? launch_any<OP_SUMS_SQ>(P) : launch_any<OP_SUMS_SQRT>(P);
        (void)hipEventRecord(ctx->ev_stop, ctx->stream);
        ctx->timed = rc == XDEMHIP_OK;
        if (rc) return rc;
    }
    XD_HIP_CHECK(ctx, hipMemcpyAsync(sums, P->sums, 8 * P->nb, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipMemcpyAsync(counts, P->counts, 8 * P->nb, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return XDEMHIP_OK;
}

int xdemhip_pairs_hist(xdemhip_pairs* P, int shift, int first, const uint64_t* prefix, uint64_t* hist) {
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    const int key_bits = P->val_dtype == XDEMHIP_F32 ? 32 : 64;
    if (!hist || shift < 0 || shift > key_bits - 8 || (shift & 7) || (!first && !prefix)) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XD_HIP_CHECK(ctx, hipMemsetAsync(P->hist, 0, 8 * (size_t)P->nb * SEL_RADIX, ctx->stream));
    if (!first) {
        if (P->val_dtype == XDEMHIP_F32) {
            std::vector<uint32_t> p32(P->nb);
            for (int k = 0; k < P->nb; ++k) p32[k] = (uint32_t)prefix[k];
            XD_HIP_CHECK(ctx, hipMemcpy(P->prefix, p32.data(), 4 * P->nb, hipMemcpyHostToDevice));
        } else {
            XD_HIP_CHECK(ctx, hipMemcpy(P->prefix, prefix, 8 * P->nb, hipMemcpyHostToDevice));
        }
    }
    if (P->n_wg > 0) {
        XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
        for (int b0 = 0; b0 < P->nb; b0 += HIST_BINS_PER_SWEEP) {
            const int nbs = (P->nb - b0) < HIST_BINS_PER_SWEEP ? (P->nb - b0) : HIST_BINS_PER_SWEEP;
            int rc = launch_any<OP_HIST>(P, shift, first, b0, nbs);
            if (rc) return rc;
        }
        (void)hipEventRecord(ctx->ev_stop, ctx->stream);
        ctx->timed = true;
    }
    XD_HIP_CHECK(ctx, hipMemcpyAsync(hist, P->hist, 8 * (size_t)P->nb * SEL_RADIX, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return XDEMHIP_OK;
}

int xdemhip_pairs_succ(xdemhip_pairs* P, const uint64_t* key, uint64_t* succ) {
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (!key || !succ) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XD_HIP_CHECK(ctx, hipMemsetAsync(P->succ, 0xFF, 8 * P->nb, ctx->stream));
    if (P->val_dtype == XDEMHIP_F32) {
        std::vector<uint32_t> p32(P->nb);
        for (int k = 0; k < P->nb; ++k) p32[k] = (uint32_t)key[k];
        XD_HIP_CHECK(ctx, hipMemcpy(P->prefix, p32.data(), 4 * P->nb, hipMemcpyHostToDevice));
    } else {
        XD_HIP_CHECK(ctx, hipMemcpy(P->prefix, key, 8 * P->nb, hipMemcpyHostToDevice));
    }
    if (P->n_wg > 0) {
        int rc = launch_any<OP_SUCC>(P);
        if (rc) return rc;
    }
    XD_HIP_CHECK(ctx, hipMemcpyAsync(succ, P->succ, 8 * P->nb, hipMemcpyDeviceToHost, ctx->stream));  // all-ones = none
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return XDEMHIP_OK;
}

}  // extern "C"

namespace {

template <typename K> __global__ void extract_prefix_kernel(const SelState<K>* st, K* out, int nb) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nb) out[k] = st[k].prefix;
}

template <typename T> __host__ inline T abs_key_value(typename KeyT<T>::type key) {  // inverse of key_abs
    typename KeyT<T>::type bits = key >> 1;
    T v;
    memcpy(&v, &bits, sizeof(T));
    return v;
}

// 8-bit digit passes over the pairs (all of them, or the 1/64 unit sample) with the selection state kept on the device;
// histograms go through the all-reduce hook, so sharded pair sets select the same global order statistics.
template <typename T>
int pairs_digit_passes(xdemhip_pairs* P, SelState<typename KeyT<T>::type>* d_st, int sample, int mode, const uint64_t* d_given,
                       std::vector<SelState<typename KeyT<T>::type>>& out, int n_passes = 0, uint32_t wide_deff = PAIR_DEFF_WIDE) {
    typedef typename KeyT<T>::type K;
    xdemhip_ctx* ctx = P->ctx;
    const int nb = P->nb, passes = KeyT<T>::passes;
    const int run = (n_passes > 0 && n_passes < passes) ? n_passes : passes;  // (bracket ends: leading digits only)
    XD_HIP_CHECK(ctx, hipMemsetAsync(d_st, 0, sizeof(SelState<K>) * nb, ctx->stream));
    XD_HIP_CHECK(ctx, hipMemsetAsync(P->hist, 0, 8 * (size_t)nb * SEL_RADIX, ctx->stream));
    P->sample = sample;
    for (int p = 0; p < run; ++p) {
        const int shift = 8 * (passes - 1 - p);
        if (P->n_wg_big > 0)
            for (int b0 = 0; b0 < nb; b0 += HIST_BINS_PER_SWEEP) {
                const int nbs = (nb - b0) < HIST_BINS_PER_SWEEP ? (nb - b0) : HIST_BINS_PER_SWEEP;
                int rc = launch_pairs<T, OP_HIST>(P, shift, (int)(p == 0), b0, nbs);
                if (rc) { P->sample = 0; return rc; }
            }
        int rc = xd_allreduce_device(ctx, P->hist, (int64_t)nb * SEL_RADIX, XDEMHIP_RED_SUM_U64);
        if (rc) { P->sample = 0; return rc; }
        hipLaunchKernelGGL((select_advance_kernel<K>), dim3(nb), dim3(64), 0, ctx->stream, d_st, reinterpret_cast<uint64_t*>(P->hist), nb,
                           shift, (int)(p == 0), (int)(p == passes - 1), mode, d_given, (const uint32_t*)nullptr, wide_deff);
        hipLaunchKernelGGL((extract_prefix_kernel<K>), dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, d_st, static_cast<K*>(P->prefix), nb);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    P->sample = 0;
    out.resize(nb);
    XD_HIP_CHECK(ctx, hipMemcpyAsync(out.data(), d_st, sizeof(SelState<K>) * nb, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return XDEMHIP_OK;
}

// Both ends of every class's bracket from ONE set of sampled digit passes: states [0, nb) select the low ends (d_st), states
// [nb, 2 nb) the high ends; the pair kernel counts every sampled pair under each of the two prefixes of its class it matches
// (PairArgs::dual).  `d_pref2` = nb key slots for the second prefix array.
template <typename T>
int pairs_digit_passes_dual(xdemhip_pairs* P, SelState<typename KeyT<T>::type>* d_st, typename KeyT<T>::type* d_pref2, uint32_t wide_deff,
                            std::vector<SelState<typename KeyT<T>::type>>& lo, std::vector<SelState<typename KeyT<T>::type>>& hi, int n_passes) {
    typedef typename KeyT<T>::type K;
    xdemhip_ctx* ctx = P->ctx;
    const int nb = P->nb, passes = KeyT<T>::passes;
    const int run = (n_passes > 0 && n_passes < passes) ? n_passes : passes;
    constexpr int SWEEP = HIST_BINS_PER_SWEEP / 2;   // two [classes][256] tables per sweep in LDS
    XD_HIP_CHECK(ctx, hipMemsetAsync(d_st, 0, sizeof(SelState<K>) * 2 * nb, ctx->stream));
    XD_HIP_CHECK(ctx, hipMemsetAsync(P->hist, 0, 8 * (size_t)nb * SEL_RADIX, ctx->stream));
    XD_HIP_CHECK(ctx, hipMemsetAsync(P->hist2, 0, 8 * (size_t)nb * SEL_RADIX, ctx->stream));
    P->sample = 1; P->dual = 1; P->prefix2 = d_pref2;
    auto leave = [&](int rc) { P->sample = 0; P->dual = 0; P->prefix2 = nullptr; return rc; };
    for (int p = 0; p < run; ++p) {
        const int shift = 8 * (passes - 1 - p);
        if (P->n_wg_big > 0)
            for (int b0 = 0; b0 < nb; b0 += SWEEP) {
                const int nbs = (nb - b0) < SWEEP ? (nb - b0) : SWEEP;
                int rc = launch_pairs<T, OP_HIST>(P, shift, (int)(p == 0), b0, nbs);
                if (rc) return leave(rc);
            }
        int rc = xd_allreduce_device(ctx, P->hist, (int64_t)nb * SEL_RADIX, XDEMHIP_RED_SUM_U64);
        if (rc == XDEMHIP_OK && p > 0) rc = xd_allreduce_device(ctx, P->hist2, (int64_t)nb * SEL_RADIX, XDEMHIP_RED_SUM_U64);
        if (rc) return leave(rc);
        if (p == 0)   // the first digit's histogram serves both states
            if (hipMemcpyAsync(P->hist2, P->hist, 8 * (size_t)nb * SEL_RADIX, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                return leave(xd_fail(ctx, XDEMHIP_EHIP, "histogram copy failed"));
        hipLaunchKernelGGL((select_advance_kernel<K>), dim3(nb), dim3(64), 0, ctx->stream, d_st, reinterpret_cast<uint64_t*>(P->hist), nb,
                           shift, (int)(p == 0), (int)(p == passes - 1), (int)SEL_BRACKET_LO_WIDE, (const uint64_t*)nullptr, (const uint32_t*)nullptr, wide_deff);
        hipLaunchKernelGGL((select_advance_kernel<K>), dim3(nb), dim3(64), 0, ctx->stream, d_st + nb, reinterpret_cast<uint64_t*>(P->hist2), nb,
                           shift, (int)(p == 0), (int)(p == passes - 1), (int)SEL_BRACKET_HI_WIDE, (const uint64_t*)nullptr, (const uint32_t*)nullptr, wide_deff);
        hipLaunchKernelGGL((extract_prefix_kernel<K>), dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, d_st, static_cast<K*>(P->prefix), nb);
        hipLaunchKernelGGL((extract_prefix_kernel<K>), dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, d_st + nb, d_pref2, nb);
        if (hipGetLastError() != hipSuccess) return leave(xd_fail(ctx, XDEMHIP_EHIP, "sampled digit pass failed"));
    }
    leave(0);
    lo.resize(nb); hi.resize(nb);
    XD_HIP_CHECK(ctx, hipMemcpyAsync(lo.data(), d_st, sizeof(SelState<K>) * nb, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipMemcpyAsync(hi.data(), d_st + nb, sizeof(SelState<K>) * nb, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return XDEMHIP_OK;
}

template <typename T>
int pairs_medians_plain(xdemhip_pairs* P, SelState<typename KeyT<T>::type>* d_st, int64_t* counts, double* medians) {
    typedef typename KeyT<T>::type K;
    xdemhip_ctx* ctx = P->ctx;
    const int nb = P->nb;
    std::vector<SelState<K>> st;
    int rc = pairs_digit_passes<T>(P, d_st, 0, SEL_MEDIAN, nullptr, st);
    if (rc) return rc;
    // successor of the selected key per class (upper median of even classes); P->prefix holds the selected keys
    XD_HIP_CHECK(ctx, hipMemsetAsync(P->succ, 0xFF, 8 * (size_t)nb, ctx->stream));
    if (P->n_wg > 0) {
        rc = launch_pairs<T, OP_SUCC>(P, 0, 0, 0, 0);
        if (rc) return rc;
    }
    rc = xd_allreduce_device(ctx, P->succ, nb, XDEMHIP_RED_MIN_U64);
    if (rc) return rc;
    std::vector<uint64_t> succ(nb);
    XD_HIP_CHECK(ctx, hipMemcpyAsync(succ.data(), P->succ, 8 * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < nb; ++k) {
        counts[k] = (int64_t)st[k].count;
        if (st[k].count == 0) { medians[k] = NAN; continue; }
        const T lo = abs_key_value<T>(st[k].prefix);
        if (st[k].count & 1) { medians[k] = (double)lo; continue; }
        const uint64_t k2 = st[k].count / 2;
        const T hi = (st[k].n_le > k2) ? lo : abs_key_value<T>((K)succ[k]);
        medians[k] = (double)(T)((T)(lo + hi) / (T)2);
    }
    return XDEMHIP_OK;
}

constexpr int64_t PAIRS_BRACKET_MIN = (int64_t)4000000000ll;  // pairs; below this the sample units are too few

// WIDE (T = float): `P` is the float32 shadow of a float64 set (xdemhip_pairs_link_shadow) -- sampled passes and the counting pass
// in float32, candidates and their selection in float64 (see pairs_kernel); returns with *answered = false instead of taking the
// plain digit passes when the bracketed route does not apply or fails (the caller then runs the float64 set's own route).
template <typename T, bool WIDE = false>
int pairs_medians_typed(xdemhip_pairs* P, int64_t* counts, double* medians, bool* answered = nullptr) {
    typedef typename KeyT<T>::type K;
    typedef typename CandT<T, WIDE>::type TC;      // candidate values ...
    typedef typename KeyT<TC>::type KC;            // ... and their keys
    if (answered) *answered = false;
    xdemhip_ctx* ctx = P->ctx;
    const int nb = P->nb;
    // small device block: selection states | klo | khi | given | counters[3 nb] | candidate counter, overflow
    unsigned char* d_small = nullptr;
    const size_t off_klo = 2 * (size_t)nb * sizeof(SelState<K>), off_khi = off_klo + 8 * (size_t)nb, off_given = off_khi + 8 * (size_t)nb,
                 off_cnt = off_given + 8 * (size_t)nb, off_ctr = off_cnt + 24 * (size_t)nb, off_rbs = off_ctr + 16, off_pref2 = off_rbs + 8,
                 small_bytes = off_pref2 + 8 * (size_t)nb;
    if (P->sel_small_bytes < small_bytes) {
        if (P->sel_small) (void)hipFree(P->sel_small);
        P->sel_small = nullptr; P->sel_small_bytes = 0;
        if (hipMalloc(reinterpret_cast<void**>(&P->sel_small), small_bytes) != hipSuccess) return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
        P->sel_small_bytes = small_bytes;
    }
    d_small = P->sel_small;
    SelState<K>* d_st = reinterpret_cast<SelState<K>*>(d_small);
    void* scratch = nullptr;
    auto cleanup = [&]() {
        // (the candidate buffers and the small blocks stay with the pair set -- tens of GB whose hipMalloc and first touch cost a
        // second: a second selection on the same set, e.g. after a warm-up call, reuses them; xdemhip_pairs_destroy frees them)
        P->khi = nullptr; P->cnt3 = nullptr; P->cand_ctr = nullptr;
    };
    const bool dbg = getenv("XDEMHIP_DEBUG") != nullptr;
    auto now_ms = [&]() { (void)hipStreamSynchronize(ctx->stream); timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    double t_phase = dbg ? now_ms() : 0.0;
    auto phase = [&](const char* what) { if (dbg) { const double t = now_ms(); fprintf(stderr, "[xdemhip] pair medians: %-28s %8.2f ms\n", what, t - t_phase); t_phase = t; } };
    bool bracket = ctx->selection_mode != 1 && nb <= HIST_BINS_PER_SWEEP && P->n_pairs >= PAIRS_BRACKET_MIN && P->n_wg_big >= 256;
    if (bracket) {
        const size_t need = scratch_size(nb);
        if (P->sel_scratch_bytes < need) {
            if (P->sel_scratch) (void)hipFree(P->sel_scratch);
            P->sel_scratch = nullptr; P->sel_scratch_bytes = 0;
            if (hipMalloc(&P->sel_scratch, need) == hipSuccess) P->sel_scratch_bytes = need;
            else { (void)hipGetLastError(); bracket = false; }
        }
        scratch = P->sel_scratch;
    }
    if (ctx->allreduce) {  // sharded pair sets: every rank must take the same route
        uint64_t can = bracket ? 1 : 0;
        if (ctx->allreduce(&can, 1, XDEMHIP_RED_MIN_U64, ctx->allreduce_user) != 0) { cleanup(); return xd_fail(ctx, XDEMHIP_EHIP, "all-reduce hook failed"); }
        bracket = can != 0;
    }
    int rc = XDEMHIP_OK;
    bool done = false;
    std::vector<K> klo(nb), khi(nb);
    std::vector<uint64_t> lo_count(nb, 0);
    // Design effect assumed for the sample (select.h): every-a-with-every-b blocks are sampled with per-lane B slots (few pairs
    // per point), i < j blocks with 4 B slots against all A points; option "vario_deff" overrides (tests, measurements).  A
    // bracket that misses its rank (the integer counts tell) costs one more attempt with the wide brackets before the plain
    // digit passes take over; with a reduction hook the counts are global, so every rank retries alike.
    uint32_t deff = ctx->vario_deff > 0 ? (uint32_t)ctx->vario_deff : (P->pdist ? PAIR_DEFF_WIDE : PAIR_DEFF_SPREAD);
    for (int attempt = 0; attempt < 3 && bracket && !done; ++attempt) {
        if (attempt >= 1) {   // a bracket missed its rank: 16 x the assumed design effect (4 x the width), then the wide rule
            if (deff >= PAIR_DEFF_WIDE) break;
            deff = (attempt == 1 && deff * 16 < PAIR_DEFF_WIDE) ? deff * 16 : PAIR_DEFF_WIDE;
            if (dbg) fprintf(stderr, "[xdemhip] pair medians: attempt %d with design effect %u\n", attempt + 1, deff);
        }
        if (bracket) {
            std::vector<SelState<K>> lo, hi;
            constexpr int BR_PASSES = 3;  // 24 leading key bits place the bracket ends finely enough
            const K low_mask = (K)(((K)1 << (8 * (KeyT<T>::passes - BR_PASSES))) - 1);
            rc = pairs_digit_passes_dual<T>(P, d_st, reinterpret_cast<K*>(d_small + off_pref2), deff, lo, hi, BR_PASSES);
            if (rc) { cleanup(); return rc; }
            phase("sampled digit passes");
            double expected = 0.0;  // candidates the brackets should hold: 64 x their width in sample ranks, at most the class
            for (int k = 0; k < nb; ++k) {
                const bool have = lo[k].count > 0;
                lo_count[k] = lo[k].count;
                klo[k] = have ? lo[k].prefix : (K)0;
                khi[k] = have ? (K)(hi[k].prefix | low_mask) : (K)~(K)0;
                if (ctx->selection_mode == 2 && have) khi[k] = klo[k];  // test mode: brackets that (almost surely) miss
                const double m = (double)lo[k].count, w = 2.0 * (double)sel_bracket_halfwidth_wide(lo[k].count, deff) + 2.0;
                expected += (P->pdist ? 64.0 : (double)(PT / SPREAD_SLOTS)) * (w < m ? w : m);
            }
            // candidate buffers sized from the brackets (x1.5 + slack), not from the pair count: 5e13 pairs (SURVEY 8d, C5 reading
            // A) need ~2e10 slots, not n_pairs / 64.  Too little memory (on any rank): plain passes.
            {
                const double want = expected * 1.5 + (double)(1 << 20);
                const double most = (double)P->n_pairs + 1024.0;
                const long long need = (long long)(want < most ? want : most);
                uint64_t got = 1;
                if (P->cand_cap < need) {
                    if (P->cand_v) (void)hipFree(P->cand_v);
                    if (P->cand_b) (void)hipFree(P->cand_b);
                    P->cand_v = nullptr; P->cand_b = nullptr; P->cand_cap = 0;
                    if (hipMalloc(&P->cand_v, (size_t)need * sizeof(TC)) != hipSuccess ||
                        hipMalloc(reinterpret_cast<void**>(&P->cand_b), (size_t)need * 2) != hipSuccess) {
                        (void)hipGetLastError();
                        got = 0;
                    } else {
                        P->cand_cap = need;
                    }
                }
                if (ctx->allreduce && ctx->allreduce(&got, 1, XDEMHIP_RED_MIN_U64, ctx->allreduce_user) != 0) {
                    cleanup();
                    return xd_fail(ctx, XDEMHIP_EHIP, "all-reduce hook failed");
                }
                if (!got) {
                    if (P->cand_v) (void)hipFree(P->cand_v);
                    if (P->cand_b) (void)hipFree(P->cand_b);
                    P->cand_v = nullptr; P->cand_b = nullptr; P->cand_cap = 0;
                    bracket = false;
                }
            }
        }
        phase("candidate buffers");
        if (bracket) {
            KC* d_klo = reinterpret_cast<KC*>(d_small + off_klo);   // (8-byte slots: rebase origins of the CANDIDATE keys)
            K* d_khi = reinterpret_cast<K*>(d_small + off_khi);
            uint64_t* d_given = reinterpret_cast<uint64_t*>(d_small + off_given);
            P->cnt3 = reinterpret_cast<unsigned long long*>(d_small + off_cnt);
            P->cand_ctr = reinterpret_cast<unsigned long long*>(d_small + off_ctr);
            P->khi = d_khi;
            hipError_t e = hipMemcpyAsync(P->prefix, klo.data(), sizeof(K) * nb, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d_khi, khi.data(), sizeof(K) * nb, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_small + off_cnt, 0, 24 * (size_t)nb + 16, ctx->stream);
            // candidate keys are rebased for their selection (select_run.h, hist_pass_kernel): low ends and the common shift
            // in key_of()'s key space (|dv| >= 0: key = bits | top bit; the pair passes use bits << 1)
            const KC top = (KC)1 << (8 * sizeof(KC) - 1);
            std::vector<KC> rb_lo(nb);
            KC widest = 0;
            for (int k = 0; k < nb; ++k) {
                if constexpr (WIDE) {
                    // candidates are the exact differences d with fl32(d) in [L, H] (L, H the float32 ends, raw bits klo >> 1 / khi >> 1):
                    // every such d lies in (pred(L), succ(H)) -- the origin and the range of the rebased float64 keys
                    const uint32_t lb = (uint32_t)(klo[k] >> 1), hb = (uint32_t)(khi[k] >> 1);
                    const uint32_t lp = lb > 0 ? lb - 1 : 0u, hs = hb < 0x7f800000u ? hb + 1 : 0x7f800000u;
                    float lf, hf;
                    memcpy(&lf, &lp, 4);
                    memcpy(&hf, &hs, 4);
                    const double ld = (double)lf, hd = (double)hf;
                    uint64_t lk, hk;
                    memcpy(&lk, &ld, 8);
                    memcpy(&hk, &hd, 8);
                    rb_lo[k] = (KC)(lk | top);
                    const KC r = (khi[k] >= klo[k] && hk >= lk) ? (KC)(hk - lk) : (KC)0;
                    widest = r > widest ? r : widest;
                } else {
                    rb_lo[k] = (KC)((klo[k] >> 1) | top);
                    const KC r = khi[k] >= klo[k] ? (KC)((khi[k] >> 1) - (klo[k] >> 1)) : (KC)0;
                    widest = r > widest ? r : widest;
                }
            }
            const uint32_t rbs = rebase_shift_of(widest);
            uint32_t* d_rbs = reinterpret_cast<uint32_t*>(d_small + off_rbs);
            if (e == hipSuccess) e = hipMemcpyAsync(d_klo, rb_lo.data(), sizeof(KC) * nb, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d_rbs, &rbs, 4, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) { cleanup(); return xd_fail(ctx, XDEMHIP_EHIP, "bracket setup failed"); }
            if (P->n_wg_big > 0) {
                XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
                rc = launch_pairs<T, OP_BRACKET, WIDE>(P, 0, 0, 0, 0);
                (void)hipEventRecord(ctx->ev_stop, ctx->stream);
                ctx->timed = rc == XDEMHIP_OK;
            }
            if (rc == XDEMHIP_OK) rc = xd_allreduce_device(ctx, P->cnt3, 3 * (int64_t)nb, XDEMHIP_RED_SUM_U64);
            if (rc == XDEMHIP_OK) rc = xd_allreduce_device(ctx, P->cand_ctr + 1, 1, XDEMHIP_RED_SUM_U64);
            if (rc) { cleanup(); return rc; }
            std::vector<uint64_t> cnt(3 * nb);
            uint64_t ctr[2];
            e = hipMemcpyAsync(cnt.data(), P->cnt3, 24 * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(ctr, P->cand_ctr, 16, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { cleanup(); return xd_fail(ctx, XDEMHIP_EHIP, std::string("bracket pass failed: ") + hipGetErrorString(e)); }
            phase("counting + compaction pass");
            if (dbg) fprintf(stderr, "[xdemhip] pair medians: %llu candidates (%.2f %% of the pairs)\n", (unsigned long long)ctr[0], 100.0 * (double)ctr[0] / (double)P->n_pairs);
            bool ok = ctr[1] == 0;
            std::vector<uint64_t> given(nb);
            for (int k = 0; k < nb; ++k) cnt[k] += cnt[nb + k];  // class total = (keys >= low end) + (keys below it)
            if (dbg) {  // how far from the bracket's centre the wanted rank lies, in half widths (1 = the bracket just missed)
                double worst = 0.0, sum2 = 0.0;
                int nn = 0, kw = -1;
                for (int k = 0; k < nb; ++k) {
                    const double total = (double)cnt[k], lt = (double)cnt[nb + k], in = (double)cnt[2 * nb + k];
                    if (total < 1.0 || in < 1.0 || lo_count[k] == 0) continue;
                    if (in >= total) continue;  // the bracket holds the whole class
                    const double off = fabs(((total - 1.0) * 0.5 - lt) - 0.5 * in) / (0.5 * in);
                    sum2 += off * off; ++nn;
                    if (off > worst) { worst = off; kw = k; }
                }
                fprintf(stderr, "[xdemhip] pair medians: wanted rank off the bracket centre by %.3f half widths at worst (class %d), rms %.3f over %d classes\n",
                        worst, kw, nn ? sqrt(sum2 / nn) : 0.0, nn);
            }
            for (int k = 0; k < nb && ok; ++k) {
                const uint64_t total = cnt[k], lt = cnt[nb + k], in = cnt[2 * nb + k];
                given[k] = ~(uint64_t)0;
                if (total == 0) continue;
                const uint64_t r = (total - 1) / 2;
                const uint64_t need = (total & 1) ? r : r + 1;
                if (lt > r || need - lt >= in) {
                    ok = false;
                    if (getenv("XDEMHIP_DEBUG"))
                        fprintf(stderr, "[xdemhip] pair medians: bracket of class %d missed (total %llu, below %llu, inside %llu, rank %llu, sample %llu)\n",
                                k, (unsigned long long)total, (unsigned long long)lt, (unsigned long long)in, (unsigned long long)r,
                                (unsigned long long)lo_count[k]);
                } else given[k] = r - lt;
            }
            if (!ok && getenv("XDEMHIP_DEBUG")) fprintf(stderr, "[xdemhip] pair medians: candidate overflow flag %llu, candidates %llu of %lld\n",
                                                       (unsigned long long)ctr[1], (unsigned long long)ctr[0], (long long)P->cand_cap);
            if (ok) {
                e = hipMemcpyAsync(d_given, given.data(), 8 * (size_t)nb, hipMemcpyHostToDevice, ctx->stream);
                if (e != hipSuccess) { cleanup(); return xd_fail(ctx, XDEMHIP_EHIP, "bracket ranks upload failed"); }
                std::vector<SelResult<KC>> res;
                rc = select_enqueue<TC>(ctx, static_cast<const TC*>(P->cand_v), P->cand_b, (int64_t)ctr[0], (int64_t)ctr[0], nullptr, nb,
                                        static_cast<unsigned char*>(scratch), SEL_GIVEN, d_given, 0, true, d_klo, d_rbs);
                if (rc == XDEMHIP_OK) rc = select_fetch<TC>(ctx, static_cast<unsigned char*>(scratch), nb, res);
                if (rc) { cleanup(); return rc; }
                phase("selection among candidates");
                for (int k = 0; k < nb; ++k) {
                    counts[k] = (int64_t)cnt[k];
                    if (cnt[k] == 0) { medians[k] = NAN; continue; }
                    res[k].st.count = cnt[k];
                    res[k].st.n_le += cnt[nb + k];
                    res[k].st.prefix = (KC)((KC)(res[k].st.prefix >> rbs) + rb_lo[k]);  // back from the rebased keys
                    if (res[k].succ != ~(uint64_t)0) res[k].succ = (uint64_t)(KC)((KC)((KC)res[k].succ >> rbs) + rb_lo[k]);
                    medians[k] = median_from<TC>(res[k]);
                }
                done = true;
            }
        }
    }
    if (WIDE) {   // (the shadow answers through the bracketed route or not at all)
        cleanup();
        if (answered) *answered = done;
        return rc;
    }
    if (!done) { rc = pairs_medians_plain<T>(P, d_st, counts, medians); phase("plain digit passes"); }
    cleanup();
    phase("cleanup");
    if (answered) *answered = true;
    return rc;
}

}  // namespace

extern "C" {

int xdemhip_pairs_medians(xdemhip_pairs* P, int64_t* counts, double* medians) {
    XdFetchScope fetch_scope_(P ? P->ctx : nullptr);  // (select_fetch queues deferred result copies)
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (!counts || !medians) return xd_fail(ctx, XDEMHIP_EINVAL, "bad argument");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (P->val_dtype == XDEMHIP_F64 && P->shadow && P->shadow->val_dtype == XDEMHIP_F32 && !ctx->allreduce) {
        // float64 differences of float32 values (option "vario_diff" = 1 on float32 inputs): the float32 shadow's passes, float64 candidates
        bool answered = false;
        const int rc = pairs_medians_typed<float, true>(P->shadow, counts, medians, &answered);
        if (rc != XDEMHIP_OK || answered) return rc;
    }
    return P->val_dtype == XDEMHIP_F32 ? pairs_medians_typed<float>(P, counts, medians) : pairs_medians_typed<double>(P, counts, medians);
}

int xdemhip_pairs_takes_brackets(xdemhip_pairs* P, int* yes) {
    if (!P || !yes) return XDEMHIP_EINVAL;
    const xdemhip_ctx* ctx = P->ctx;
    // (the conditions of the bracketed route in pairs_medians_typed, plus: a float32 shadow is only read on hook-less contexts)
    *yes = (ctx->selection_mode != 1 && P->nb <= HIST_BINS_PER_SWEEP && P->n_pairs >= PAIRS_BRACKET_MIN && P->n_wg_big >= 256 && !ctx->allreduce) ? 1 : 0;
    return XDEMHIP_OK;
}

int xdemhip_pairs_link_shadow(xdemhip_pairs* P, xdemhip_pairs* shadow) {
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (shadow) {
        if (shadow->ctx != ctx || P->val_dtype != XDEMHIP_F64 || shadow->val_dtype != XDEMHIP_F32 || shadow->nblk != P->nblk || shadow->nb != P->nb ||
            shadow->pdist != P->pdist || shadow->h_a_off != P->h_a_off || shadow->h_b_off != P->h_b_off)
            return xd_fail(ctx, XDEMHIP_EINVAL, "pairs_link_shadow: a float64 set and a float32 set of the same blocks and edges are required");
    }
    P->shadow = shadow;
    return XDEMHIP_OK;
}

}  // extern "C"
