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
#include "nk_geom.h"

namespace xd {

// ---- aux: gradient -> slope tangent, aspect, valid mask --------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nk_aux_kernel(const T* __restrict__ ref, const T* __restrict__ tba,
                                                     const uint8_t* __restrict__ inlier, NkGeom g,
                                                     T* __restrict__ slope_tan, T* __restrict__ aspect,
                                                     uint8_t* __restrict__ valid, unsigned long long* n_valid,
                                                     int64_t row0, int64_t row1) {
    // raster rows [row0, row1): blockIdx.x tiles the columns, blockIdx.y strides over the rows
    unsigned long long local = 0;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t H = g.H, W = g.W;
    if (j < W)
    for (int64_t i = row0 + blockIdx.y; i < row1; i += gridDim.y) {
        const int64_t p = (i - g.roff) * W + j;
        const T c = ref[p];
        T gy, gx;
        // np.gradient, unit spacing: central differences inside, one-sided on the borders of the RASTER
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

// ---- dh at a shifted position + min / max aspect over its finite pixels -------------------------------------------
struct DhStats {
    uint64_t asp_min, asp_max;  // order-preserving keys (widened to 64 bit) of min / max aspect among finite dh
};
static __global__ void nk_stats_init_kernel(DhStats* s) {
    if (threadIdx.x == 0) { s->asp_min = ~(uint64_t)0; s->asp_max = 0; }
}
// ... and, for the one-pass step, the zeroing of its device block and of the EXT survivor counters in the same launch (two memsets less)
static __global__ void nk_step_init_kernel(DhStats* s, uint64_t* fz, int64_t fz_words, unsigned long long* ext_survivors /* [2] */) {
    for (int64_t w = threadIdx.x; w < fz_words; w += blockDim.x) fz[w] = 0;
    if (threadIdx.x == 0) { s->asp_min = ~(uint64_t)0; s->asp_max = 0; ext_survivors[0] = 0; ext_survivors[1] = 0; }
}

template <typename T>
__global__ __launch_bounds__(256) void nk_dh_kernel(const T* __restrict__ ref, const T* __restrict__ tba,
                                                    const uint8_t* __restrict__ valid, const T* __restrict__ aspect,
                                                    NkGeom g, T* __restrict__ dh, DhStats* stats, int64_t row0, int64_t row1) {
    typedef typename KeyT<T>::type K;
    K kmin = ~(K)0, kmax = 0;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const BiAxis col = bi_axis(j, g.dc, g.W, g.rule);
    if (j < g.W)
    for (int64_t i = row0 + blockIdx.y; i < row1; i += gridDim.y) {
        const int64_t p = (i - g.roff) * g.W + j;
        const BiTap t = bi_combine(g, bi_axis(i, g.dr, g.H, g.rule), col);
        // all loads issued unconditionally (clamped taps) so they overlap; validity is applied afterwards
        const BiVals<T> tv = bi_load<T>(tba, t);
            const T a00 = tv.a00, a01 = tv.a01, a10 = tv.a10, a11 = tv.a11;
        const T rv = ref[p];
        const T av = aspect[p];
        const uint8_t vd = valid[p];  // (every load of the pixel is issued before the first use)
        T val;
        const bool ok = bi_value<T>(g, tba, t, a00, a01, a10, a11, val) & (vd != 0);
        T out = t_sub(rv, val);
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

// ---- min / max aspect without reading the aspect raster every step (round 3) -----------------------------------------------
// The dh pass needs min / max of the aspect over the pixels whose dh is finite (SciPy's bin edges) -- two numbers, each set by
// ONE pixel -- and used to read the 4-byte aspect of every pixel for them.  At plan creation the valid pixels whose aspect lies
// in the lowest / highest ~16 K of the raster are listed (EXT lists); a step evaluates dh at those few pixels only: the minimum
// over the listed pixels with a finite dh IS the minimum over all of them as long as one listed pixel survives (every unlisted
// pixel has a larger aspect); if none survives, or a list came out empty / overfull, the step falls back to the plain route, whose
// kernel reads the aspect (counter [6] of the one-pass step: the plan then stops using its lists).  The same kernel folds the valid mask into a plan-owned copy of
// the reference DEM (NaN where a pixel is not valid): dh = ref - bilinear(tba) is then non-finite by itself and the dh pass
// reads neither the mask nor the aspect: 12 instead of 17 B/pixel.
constexpr int EXT_TARGET = 16384, EXT_CAP = 4 * EXT_TARGET;
template <typename T>
__global__ __launch_bounds__(256) void nk_ext_build_kernel(const T* __restrict__ ref, const uint8_t* __restrict__ valid,
                                                           const T* __restrict__ aspect, int64_t q0, int64_t n, T thr_lo, T thr_hi,
                                                           T* __restrict__ ref_m, int64_t* __restrict__ ext_idx,
                                                           unsigned long long* __restrict__ ext_cnt /* [2] */) {
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = base + threadIdx.x;
        bool lo = false, hi = false;
        if (p < n) {
            const int64_t q = q0 + p;
            const bool v = valid[q] != 0;
            ref_m[q] = v ? ref[q] : (T)NAN;
            const T a = aspect[q];
            lo = v && a < thr_lo;
            hi = v && a > thr_hi;
        }
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const bool mine = w ? hi : lo;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                unsigned long long b = 0;
                if (lane == leader) b = atomicAdd(&ext_cnt[w], (unsigned long long)__popcll(m));
                b = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(b >> 32), leader) << 32) |
                    (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)b, leader);
                const unsigned long long pos = b + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                if (mine && pos < (unsigned long long)EXT_CAP) ext_idx[(int64_t)w * EXT_CAP + (int64_t)pos] = q0 + p;
            }
        }
    }
}
// one thread per listed pixel: is its dh finite at this step's shift?  -> min / max aspect keys, survivors per list
template <typename T>
__global__ __launch_bounds__(256) void nk_ext_eval_kernel(const T* __restrict__ ref_m, const T* __restrict__ tba, const T* __restrict__ aspect,
                                                          NkGeom g, const int64_t* __restrict__ ext_idx, const unsigned long long* __restrict__ ext_cnt,
                                                          DhStats* stats, unsigned long long* survivors /* [2] */) {
    typedef typename KeyT<T>::type K;
    const int w = blockIdx.y;
    const unsigned long long cnt = ext_cnt[w] < (unsigned long long)EXT_CAP ? ext_cnt[w] : (unsigned long long)EXT_CAP;
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    K key = 0;
    if ((unsigned long long)k < cnt) {
        const int64_t q = ext_idx[(int64_t)w * EXT_CAP + k];
        const int64_t li = q / g.W, j = q - li * g.W;
        const BiTap t = bi_locate(g, li + g.roff, j);
        const BiVals<T> tv = bi_load<T>(tba, t);
        T val;
        const bool in = bi_value<T>(g, tba, t, tv.a00, tv.a01, tv.a10, tv.a11, val);
        const T out = t_sub(ref_m[q], val);
        ok = in && t_finite(out);
        key = key_of(aspect[q]);
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
    if (!m) return;
    K best = ok ? key : (w ? (K)0 : ~(K)0);
    for (int off = 32; off > 0; off >>= 1) {
        const K o = k_shfl_down(best, off);
        best = w ? (o > best ? o : best) : (o < best ? o : best);
    }
    if ((threadIdx.x & 63) == 0) {
        if (w) k_atomic_max(&stats->asp_max, (uint64_t)best);
        else k_atomic_min(&stats->asp_min, (uint64_t)best);
        atomicAdd(&survivors[w], (unsigned long long)__popcll(m));
    }
}

// Row-tap table of a chunk of rows (one entry per row, computed once per workgroup into LDS by the one-pass kernel below).
struct NkRowTab { double fr; int k0l; int flags; int rnl; int pad_; };  // upper tap row (buffer-local, clamped), bit 0 = taps inside the raster, bit 1 = d1; rnl: buffer row of the NEAREST pixel (rules 2 / 3), -1 = its neighbourhood leaves the raster

// ---- rules 2 / 3 ("dilate3x3" / "dilate_cross") without a neighbourhood read per pixel (round 5) -------------------------------
// Under these rules a sample is valid iff its four taps are finite AND the 3 x 3 / cross neighbourhood of the pixel NEAREST to the
// tap position holds no non-finite value and stays inside the raster (bi_value above).  The second condition depends on tba alone:
// it is evaluated ONCE per plan into a bit per pixel ("bad" = neighbourhood not clean; border pixels, the columns beyond W in a
// row's last word and one all-ones pad word on either side of every row are bad too), and the streaming kernels then test the bit
// of the nearest pixel -- the same floor(pos + 0.5) per row and per lane as bi_combine -- next to their rule-0 arithmetic: one 8-byte
// word per lane and row, two distinct words per wave.  Results are those of the generic kernel bit for bit (GPU tests per rule).
template <typename T>
__global__ __launch_bounds__(256) void nk_badbits_kernel(const T* __restrict__ tba /* the plan's buffer: rows roff .. roff + nbuf of the raster */,
                                                         int64_t H, int64_t W, int64_t roff, int64_t nbuf, int rule, int64_t wpr,
                                                         uint64_t* __restrict__ out /* [nbuf][wpr] */) {
    const int lane = threadIdx.x & 63;
    const int64_t word = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // data word of the row (pad words: below)
    const int64_t nwords = wpr - 2;
    for (int64_t r = blockIdx.y; r < nbuf; r += gridDim.y) {
        if (word < nwords) {
            const int64_t c = word * 64 + lane;
            // bad: on the raster's border -- or, in a row block, in the first / last row of the buffer where that is not the raster's:
            // the halo rule of the step (HaloTooSmall) keeps every nearest pixel off those rows, so the bit is never consulted
            bool bad = !(roff + r >= 1 && c >= 1 && roff + r + 1 < H && c + 1 < W) || r < 1 || r + 1 >= nbuf;
            if (!bad) {
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx)
                        if (rule == 2 || dy == 0 || dx == 0) bad = bad || !t_finite(tba[(r + dy) * W + (c + dx)]);
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(bad);
            if (lane == 0) out[r * wpr + 1 + word] = (uint64_t)m;
        }
        if (blockIdx.x == 0 && threadIdx.x < 2) out[r * wpr + (threadIdx.x ? wpr - 1 : 0)] = ~(uint64_t)0;
    }
}
// nearest pixel of a row / column position as bi_combine computes it
__device__ __forceinline__ int64_t nk_nearest(double pos) { return (int64_t)floor(pos + 0.5); }
// ---- full-grid translation resample (Coreg.apply for a pure shift) --------------------------------
// out(r, c) = bilinear(src)(r + dr, c + dc) + dz with the same tap convention / NaN rule as the Nuth-Kaab step.
template <typename T>
__global__ __launch_bounds__(256) void shift_bilinear_kernel(const T* __restrict__ src, NkGeom g, T dz, T* __restrict__ out) {
    const int64_t n = g.H * g.W;
    const double invW = 1.0 / (double)g.W;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        int64_t i, j;
        row_col(p, g.W, invW, i, j);
        const BiTap t = bi_locate(g, i, j);
        T val;
        const BiVals<T> tv = bi_load<T>(src, t);
        const bool ok = bi_value<T>(g, src, t, tv.a00, tv.a01, tv.a10, tv.a11, val);
        out[p] = ok ? t_add(val, dz) : (T)NAN;
    }
}

// SciPy's rule for samples at or beyond the rightmost edge (_binned_statistic.py:_bin_numbers): such a sample joins the last
// bin iff np.around(x, decimal) == np.around(last_edge, decimal), decimal = int(-log10(min edge spacing)) + 6, evaluated in
// the sample dtype.  With the automatic edges the largest sample IS the last edge (decimal = NK_AUTO_EDGES: always true);
// explicit edges carry their decimal (computed by the caller exactly like SciPy).
constexpr int NK_AUTO_EDGES = -100000;
template <typename T> __device__ __forceinline__ bool on_last_edge(T x, T last, int decimal) {
    if (decimal == NK_AUTO_EDGES) return true;
    const T p10 = (T)pow(10.0, (double)(decimal < 0 ? -decimal : decimal));
    const T rx = decimal >= 0 ? rint(x * p10) / p10 : rint(x / p10) * p10;
    const T rl = decimal >= 0 ? rint(last * p10) / p10 : rint(last / p10) * p10;
    return rx == rl;
}

// ---- y = (dh - vshift) / slope_tan, aspect bin id, sums for nanmean / nanstd --------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nk_y_kernel(const T* __restrict__ dh, const T* __restrict__ slope_tan,
                                                   const T* __restrict__ aspect, int64_t n, T vshift,
                                                   const T* __restrict__ edges, int nb, T* __restrict__ y,
                                                   uint16_t* __restrict__ bins, double* sums /* [sum, sumsq] */,
                                                   int last_decimal = NK_AUTO_EDGES) {
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
            if (idx == nb && on_last_edge<T>(x, e[nb], last_decimal)) idx = nb - 1;
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
// Aspect-bin cache: the bin of a pixel depends only on its aspect and on the edges, and the edges (SciPy's linspace between the
// min and max aspect of the valid pixels, or the caller's explicit edges) hardly ever change between the steps of a fit.  The
// first pass under a given set of edges stores every pixel's bin id (uint16, 0xFFFF = outside) and later passes read 2 bytes
// instead of digitizing a 4-byte aspect again; a one-thread kernel compares the edges with the record of what the cache holds.
// The aspect-bin cache holds ONE BYTE per pixel (round 6; two until then): every route that reads it takes at most MAX_BINS_PER_SWEEP
// = 128 bins, 0xFF = "no bin" -- the one-pass step touches 13 instead of 14 bytes per pixel, the bin pass of the plain route 9
// instead of 10.  Candidates and samples keep 16-bit bin ids (the selection code is shared with nd_binning).
typedef uint8_t nk_bin_t;
constexpr nk_bin_t NK_NOBIN = 0xFF;
constexpr int NK_BINCACHE_MAX_BINS = 254;
__host__ __device__ inline uint16_t nk_bin16(nk_bin_t b) { return b == NK_NOBIN ? (uint16_t)0xFFFF : (uint16_t)b; }
__host__ __device__ inline nk_bin_t nk_bin8(uint16_t b) { return b == (uint16_t)0xFFFF ? NK_NOBIN : (nk_bin_t)b; }
struct BinCacheRec { double e0, eN; int nb; int fresh; };
template <typename T> __global__ void nk_bin_cache_check_kernel(const T* edges, int nb, BinCacheRec* rec, int force) {
    if (threadIdx.x == 0) {
        const double e0 = (double)edges[0], eN = (double)edges[nb];
        rec->fresh = (!force && rec->nb == nb && rec->e0 == e0 && rec->eN == eN) ? 1 : 0;
    }
}
// (launched once a pass over ALL own pixels has been queued under these edges; nb = 0 invalidates)
template <typename T> __global__ void nk_bin_cache_commit_kernel(const T* edges, int nb, BinCacheRec* rec) {
    if (threadIdx.x == 0) {
        rec->e0 = nb > 0 ? (double)edges[0] : 0.0;
        rec->eN = nb > 0 ? (double)edges[nb] : 0.0;
        rec->nb = nb;
    }
}

template <typename T> struct NkYSource {
    const T* dh;
    const T* slope_tan;
    const T* aspect;
    const T* vshift_p;  // device scalar (uploaded by the host on the plain route)
    const T* edges;  // [nb + 1], device
    double* sums;    // [sum, sumsq], device
    T* e;            // LDS copy of the edges (setup)
    double inv_width;
    T vshift;
    int last_decimal;  // rounding precision of SciPy's rightmost-edge rule (NK_AUTO_EDGES for the automatic edges)
    nk_bin_t* bcache = nullptr;          // per-pixel bin ids (null: no cache)
    const BinCacheRec* rec = nullptr;
    int fresh = -1;                      // 1: read the cache, 0: digitize and fill it, -1: digitize only
    struct Raw { T d, st, x; int64_t p; uint16_t b; };
    struct Acc { double s1 = 0.0, s2 = 0.0; };
    static size_t lds_bytes(int nb) { return sizeof(T) * (size_t)(nb + 1) + 8; }
    __device__ __forceinline__ void setup(unsigned char* lds, int nb) {
        e = reinterpret_cast<T*>((reinterpret_cast<uintptr_t>(lds) + 7) & ~(uintptr_t)7);
        for (int k = threadIdx.x; k <= nb; k += blockDim.x) e[k] = edges[k];
        inv_width = (double)nb / ((double)edges[nb] - (double)edges[0]);
        vshift = *vshift_p;
        fresh = bcache ? __builtin_amdgcn_readfirstlane(rec->fresh) : -1;
    }
    __device__ __forceinline__ void fetch(int64_t p, Raw& r) const {
        r.d = dh[p]; r.st = slope_tan[p]; r.p = p;
        if (fresh == 1) { r.b = nk_bin16(bcache[p]); r.x = (T)0; }
        else { r.x = aspect[p]; r.b = 0xFFFF; }
    }
    __device__ __forceinline__ void blank(Raw& r) const { r.d = (T)NAN; r.st = (T)1; r.x = (T)0; r.p = -1; r.b = 0xFFFF; }
    __device__ __forceinline__ uint16_t digitize(T x, int nb) const {
        int idx = (int)(((double)x - (double)e[0]) * inv_width);  // (same digitize as nk_y_kernel)
        idx = idx < 0 ? 0 : (idx > nb ? nb : idx);
        while (idx > 0 && !(e[idx] <= x)) --idx;
        while (idx < nb && e[idx + 1] <= x) ++idx;
        if (!(e[0] <= x)) idx = -1;
        if (idx == nb && on_last_edge<T>(x, e[nb], last_decimal)) idx = nb - 1;
        return (idx >= 0 && idx < nb) ? (uint16_t)idx : (uint16_t)0xFFFF;
    }
    template <bool ACC> __device__ __forceinline__ bool eval(const Raw& r, int nb, T& v, uint16_t& b, Acc& acc) const {
        v = (T)NAN;
        b = 0xFFFF;
        uint16_t bin = r.b;
        if (fresh != 1) {
            if (fresh == 0) {  // filling pass: every pixel, whatever its dh is this step
                if (r.p >= 0) { bin = digitize(r.x, nb); bcache[r.p] = nk_bin8(bin); }
            } else if (r.d == r.d) {
                bin = digitize(r.x, nb);
            }
        }
        if (!(r.d == r.d)) return false;
        const T yv = t_div(t_sub(r.d, vshift), r.st);
        if (ACC) { acc.s1 += (double)yv; acc.s2 += (double)yv * (double)yv; }
        v = yv;
        if (bin == 0xFFFF) return false;
        b = bin;
        return yv == yv;
    }
    __device__ __forceinline__ void finish(Acc& acc) const {
        double s1 = acc.s1, s2 = acc.s2;
        for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&sums[0], s1); atomicAdd(&sums[1], s2); }
    }
    // the counting pass has a dedicated kernel for cached bins (nk_bins_lean_kernel below): the generic one steps aside then
    static constexpr bool HAS_LEAN = true;
    __device__ __forceinline__ bool skip_counting_pass() const { return fresh == 1; }
    int launch_lean(xdemhip_ctx* ctx, int64_t n, int nb, const typename KeyT<T>::type* d_klo, const typename KeyT<T>::type* d_khi,
                    uint64_t* d_cnt, T* c_vals, uint16_t* c_bins, unsigned long long* d_flags, int64_t c_cap) const;
};

// ---- lean form of the per-bin counting pass (select_run.h: bracket_pass_kernel over NkYSource) for steps whose aspect bins are
// cached: 10 bytes per pixel (dh, slope tangent, bin id), one LDS read for the bin's bracket, one LDS atomic for its class,
// candidates compacted through per-wave staging (no workgroup barrier in the loop).  The generic kernel -- measured VALU-bound at
// about 145 vector instructions per element -- stays in charge of the steps that (re)fill the cache; both are queued, each
// leaves at once when the device-side `fresh` flag says it is the other one's turn.
constexpr int NKB_TILE = 8;
template <typename T>
__global__ __launch_bounds__(HIST_THREADS) void nk_bins_lean_kernel(const T* __restrict__ dh, const T* __restrict__ slope_tan,
                                                                    const nk_bin_t* __restrict__ bcache, const BinCacheRec* rec,
                                                                    const T* vshift_p, int64_t n, int nb, int copies,
                                                                    const typename KeyT<T>::type* __restrict__ klo,
                                                                    const typename KeyT<T>::type* __restrict__ khi,
                                                                    uint64_t* counters /* [3][nb] */, T* out_v, uint16_t* out_b,
                                                                    unsigned long long* ctr, int64_t cap, double* sums) {
    typedef typename KeyT<T>::type K;
    if (rec->fresh != 1) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // workgroup staging exactly as in bracket_pass_kernel (one global atomic per burst of thousands of candidates)
    BlockStage<T> st;
    st.v = reinterpret_cast<T*>(smem);
    st.b = reinterpret_cast<uint16_t*>(st.v + SEL_STAGE_CAP);
    st.base = reinterpret_cast<unsigned long long*>(st.b + SEL_STAGE_CAP);
    st.held = reinterpret_cast<int*>(st.base + 1);
    K* lohi = reinterpret_cast<K*>(st.base + 2);               // [nb][2]
    uint32_t* c = reinterpret_cast<uint32_t*>(lohi + 2 * nb);  // [copies][cs]: 3 counters per bin
    const int cs = (3 * nb) | 1;  // odd copy stride: the copies of one counter fall into different LDS banks
    for (int k = threadIdx.x; k < nb; k += blockDim.x) { lohi[2 * k] = klo[k]; lohi[2 * k + 1] = khi[k]; }
    for (int k = threadIdx.x; k < cs * copies; k += blockDim.x) c[k] = 0;
    if (threadIdx.x == 0) *st.held = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t* cc = c + (threadIdx.x % copies) * cs;
    const T vshift = *vshift_p;
    double s1 = 0.0, s2 = 0.0;
    int it = 0;
    const int64_t step = (int64_t)blockDim.x * NKB_TILE;
    for (int64_t base = (int64_t)blockIdx.x * step; base < n; base += (int64_t)gridDim.x * step) {
        T d[NKB_TILE], sl[NKB_TILE];
        uint16_t b[NKB_TILE];
        if (base + step <= n) {  // (uniform) whole tile inside: no per-element bounds logic
#pragma unroll
            for (int q = 0; q < NKB_TILE; ++q) {
                const int64_t p = base + (int64_t)q * blockDim.x + threadIdx.x;
                d[q] = __builtin_nontemporal_load(dh + p); sl[q] = __builtin_nontemporal_load(slope_tan + p);
                b[q] = nk_bin16(__builtin_nontemporal_load(bcache + p));
            }
        } else {
#pragma unroll
            for (int q = 0; q < NKB_TILE; ++q) {
                const int64_t p = base + (int64_t)q * blockDim.x + threadIdx.x;
                const bool in = p < n;
                const int64_t pc = in ? p : n - 1;
                d[q] = dh[pc]; sl[q] = slope_tan[pc]; b[q] = nk_bin16(bcache[pc]);
                if (!in) d[q] = (T)NAN;
            }
        }
#pragma unroll
        for (int q = 0; q < NKB_TILE; ++q) {
            const T yv = t_div(t_sub(d[q], vshift), sl[q]);
            const bool has = d[q] == d[q];
            if (has) { s1 += (double)yv; s2 += (double)yv * (double)yv; }
            const bool ok = has & (b[q] != 0xFFFF) & (yv == yv);
            bool cand = false;
            if (ok) {
                const K key = key_of(yv);
                const K l = lohi[2 * b[q]], h = lohi[2 * b[q] + 1];
                cand = (key >= l) & (key <= h);
                const int cls = key < l ? 1 : (cand ? 2 : 0);
                atomicAdd(&cc[__umul24((unsigned)cls, (unsigned)nb) + b[q]], 1u);
            }
            st.append_bounded(cand, yv, b[q], &ctr[2]);
        }
        // (same exposure to candidate bursts as the generic kernel: 16384 elements between two looks at the staging buffer)
        if ((++it % (SEL_FLUSH_EVERY * SEL_TILE / NKB_TILE)) == 0) st.sync_and_flush_at(0, out_v, out_b, &ctr[1], cap, &ctr[2]);
    }
    st.sync_and_flush_at(0, out_v, out_b, &ctr[1], cap, &ctr[2]);
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off); s2 += __shfl_down(s2, off); }
    if (lane == 0) { atomicAdd(&sums[0], s1); atomicAdd(&sums[1], s2); }
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

template <typename T>
int NkYSource<T>::launch_lean(xdemhip_ctx* ctx, int64_t n, int nb, const typename KeyT<T>::type* d_klo, const typename KeyT<T>::type* d_khi,
                              uint64_t* d_cnt, T* c_vals, uint16_t* c_bins, unsigned long long* d_flags, int64_t c_cap) const {
    typedef typename KeyT<T>::type K;
    if (!bcache || n <= 0) return XDEMHIP_OK;
    int copies = (28 * 1024) / (nb * 12);
    copies = copies < 1 ? 1 : (copies > 32 ? 32 : copies);
    const size_t lds = (size_t)SEL_STAGE_CAP * (sizeof(T) + 2) + 16 + (size_t)nb * 2 * sizeof(K) + (size_t)((3 * nb) | 1) * 4 * (size_t)copies;
    int rc = set_big_lds(ctx, nk_bins_lean_kernel<T>, lds);
    if (rc) return rc;
    // (grid-stride: two 1024-thread workgroups per CU are resident -- LDS -- and that is the grid)
    hipLaunchKernelGGL((nk_bins_lean_kernel<T>), dim3(grid_for(ctx, n, HIST_THREADS * NKB_TILE, 2)), dim3(HIST_THREADS), lds, ctx->stream, dh,
                       slope_tan, bcache, rec, vshift_p, n, nb, copies, d_klo, d_khi, d_cnt, c_vals, c_bins, d_flags, c_cap, sums);
    XD_HIP_CHECK(ctx, hipGetLastError());
    return XDEMHIP_OK;
}


}  // namespace xd

#include "nk_onepass.h"


// ================================================================================================================
using namespace xd;

struct xdemhip_nk_plan {
    xdemhip_ctx* ctx = nullptr;
    int dtype = XDEMHIP_F32;
    int64_t H = 0, W = 0;                 // raster shape (global)
    int64_t roff = 0, nbuf = 0;           // the buffers hold raster rows [roff, roff + nbuf)
    int64_t row0 = 0, row1 = 0;           // this rank's own rows [row0, row1) (whole raster unless sharded)
    void *ref = nullptr, *tba = nullptr;  // device (owned when own_inputs)
    uint8_t* inlier = nullptr;            // device copy kept for re-partitioning (owned when own_inputs)
    uint8_t* sub_inlier = nullptr;        // xdemhip_nk_subsample: the inlier mask in force after it (owned; `inlier` then points here and
    uint8_t* inlier_user = nullptr;       //  the mask the plan was created with is remembered for its release)
    bool subsampled = false;
    bool own_inputs = false;
    void *slope_tan = nullptr, *aspect = nullptr, *dh = nullptr, *y = nullptr;
    uint8_t* valid = nullptr;
    uint16_t* bins = nullptr;
    nk_bin_t* bcache = nullptr;   // aspect-bin cache (NkYSource), one BYTE per buffer pixel
    bool bcache_force = true;     // the cache does not hold the bins of the current own rows / edges: refill at the next step
    void* ref_m = nullptr;        // reference DEM with NaN where a pixel is not valid (EXT route of the dh pass)
    int64_t* ext_idx = nullptr;   // [2][EXT_CAP] pixels with the lowest / highest aspects
    unsigned long long* ext_cnt = nullptr;  // [0..1] list lengths, [2..3] survivors of the current step
    bool ext_ok = false;
    uint64_t* fz = nullptr;       // device block of the one-pass step (nk_step_onepass): counters, bracket keys, v^, sums
    size_t fz_bytes = 0;
    void* cd_vals = nullptr;      // ... candidates of the median of dh
    int64_t cd_cap = 0;
    void* c_st = nullptr;         // ... slope tangents of the bin candidates (next to ws.c_vals / ws.c_bins)
    double* wg_sums = nullptr;    // ... the pass's five float64 sums per workgroup (added up in a fixed order: nk_sums_reduce)
    size_t wg_sums_cap = 0;
    unsigned char* fz_pack = nullptr;   // ... the step's small results, gathered for one device-to-host copy
    size_t fz_pack_bytes = 0;
    std::vector<unsigned char> fz_host;
    int64_t n_onepass = 0, n_plain = 0;   // steps answered by each route (xdemhip_nk_route_counts)
    // one-pass route: the sample brackets of the NEXT step are a fraction (code fz_narrow, select.h: sel_narrowed / sel_narrow_unit)
    // of what the rule for fully correlated sample lines asks (select.h: sel_bracket_halfwidth); set from how far off the bracket
    // centres the wanted ranks lay in the steps so far, never below fz_unit_min (raised by a miss)
    int fz_narrow = 0;
    double fz_unit_min = 0.25;
    double fz_worst = 0.0;   // largest |wanted rank - bracket centre| seen, in half widths of the FULL rule
    double fz_off2 = 0.0;    // sum of squares of those offsets over all brackets of all steps so far
    int64_t fz_offn = 0;
    // one-pass route, round 6: PREDICTED brackets.  Once a fit has settled its offsets move by a few thousandths of a pixel per
    // step and the 73 exact medians of the previous step, moved by what the Nuth-Kaab model says the shift does to them, bracket
    // this step's medians better than a fresh 1/64 sample does -- the two sample kernels and their six digit passes (~0.26 ms of
    // a 1.66 ms step at C3) are skipped.  What the previous step left (host side; identical on every rank of a partitioned plan):
    bool pr_have = false;             // ... it answered on the one-pass route and everything below is its record
    int pr_nb = 0;
    double pr_sx = 0, pr_sy = 0, pr_rx = 0, pr_ry = 0;   // its shifts and resolutions (georeferenced units)
    double pr_v = 0, pr_wd = 0;       // its vshift; half width (value) of the last SAMPLED bracket of the median of dh
    double pr_st = 0;                 // n / sum(1 / slope_tan): the scale that turns a shift in pixels into a change of dh
    std::vector<double> pr_med, pr_w, pr_mid;   // per bin: exact median of y, half width of the last sampled bracket, centre aspect
    std::vector<unsigned char> pr_ok; // ... the bin had a median
    double pr_err = 1e30, pr_err_d = 1e30;   // how far the prediction of the LAST step would have been off (or was), in sampled half widths: bins / median of dh
    double pr_dpx = 1e30;             // the shift change of the last step, pixels
    int pr_cooldown = 0;              // sampled steps still to run after a predicted bracket missed
    int64_t n_predicted = 0, n_predicted_d = 0, n_predict_miss = 0;   // steps with every bracket predicted / with the bracket of the median of dh predicted / reruns after a miss
    // one-pass step on PARTITIONED plans (reduction hook + xdemhip_set_rank): the two exchange buffers, this rank's own classes and the
    // counters rewritten for the gathered bucket values (nk_mr_* kernels); the ranks' agreement on the route, renewed when what it rests
    // on changes
    uint64_t *mr_a = nullptr, *mr_b = nullptr, *mr_small = nullptr;
    size_t mr_a_words = 0, mr_b_words = 0;
    int64_t mr_key[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    bool mr_can = false;
    int64_t mr_nglob = 0;
    void* scratch = nullptr;  // edges, stats, sums, selection states, successor keys, histograms
    size_t scratch_bytes = 0;
    int max_bins = 0;
    long long n_valid0 = 0;
    xd::SelWorkspace ws;  // bracketed selection (select_run.h): sample + candidate buffers
    uint64_t* badbits = nullptr;   // rules 2 / 3: one bit per buffer pixel, "the neighbourhood of this pixel of tba is not clean" (nk_badbits_kernel)
    int64_t bad_wpr = 0;           // ... 64-bit words per row, one all-ones pad word on either side
    int bin_stat = XDEMHIP_BINSTAT_MEDIAN;
    int nan_rule = 0;
    int custom_decimal = 0;            // ... and the decimal of SciPy's rightmost-edge rule for them
    std::vector<double> custom_edges;  // explicit bin edges (xdemhip_nk_set_bin_edges); empty: SciPy's linspace(min, max, n + 1)
};

namespace {

// scratch (first 16 KiB): bin edges [0, 8208) | device-side step results at OFF_INFO
constexpr size_t OFF_INFO = 12288;  // T vshift (8 B slot) | uint64 n_valid | uint64 flags | double vshift

// column tiles of 256 x enough row-strided workgroups to fill the chip (~16 workgroups per CU)
dim3 grid2d(const xdemhip_ctx* ctx, int64_t W, int64_t rows) {
    const int64_t gx = (W + 255) / 256;
    int64_t gy = ((int64_t)ctx->num_cu * 16 + gx - 1) / gx;
    gy = gy < 1 ? 1 : (gy > rows ? (rows > 0 ? rows : 1) : gy);
    return dim3((unsigned)gx, (unsigned)gy);
}

template <typename T> void make_edges(double smin, double smax, int nb, std::vector<T>& e) {
    e.resize(nb + 1);
    make_edges_into<T>(smin, smax, nb, e.data());
}

// ---- xdemhip_nk_subsample: the valid pixels whose RANK (position among the valid pixels in raster order) is listed stay inliers ----------
// What the caller's `rng.choice(np.flatnonzero(valid), k, replace=False)` selects, without the mask travelling to the host and back:
// the host draws ranks, the device turns them into pixels.  Tiles of 4096 pixels: counts, an exclusive scan over the tiles, then every
// tile ranks its own valid pixels and looks its ranks up in the byte array of marked ranks.
constexpr int SUBS_TILE = 4096;
static __global__ __launch_bounds__(256) void nk_subs_count_kernel(const uint8_t* __restrict__ valid, int64_t n, unsigned long long* __restrict__ tile_cnt) {
    const int64_t t0 = (int64_t)blockIdx.x * SUBS_TILE + (int64_t)threadIdx.x * 16;
    int c = 0;
    if (t0 + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(valid + t0);   // (tiles start at multiples of 4096: aligned)
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b) c += ((w[k] >> (8 * b)) & 0xFFu) ? 1 : 0;
    } else {
        for (int64_t p = t0; p < n && p < t0 + 16; ++p) c += valid[p] ? 1 : 0;
    }
    __shared__ int s[256];
    s[threadIdx.x] = c;
    __syncthreads();
    for (int half = 128; half > 0; half >>= 1) {
        if ((int)threadIdx.x < half) s[threadIdx.x] += s[threadIdx.x + half];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = (unsigned long long)s[0];
}
// exclusive scan over the tile counts, in place (one workgroup walks them in pieces of 1024 with a carry)
static __global__ __launch_bounds__(1024) void nk_subs_scan_kernel(unsigned long long* tile_cnt, int64_t n_tiles, unsigned long long* total) {
    __shared__ unsigned long long s[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0ull;
    __syncthreads();
    for (int64_t b0 = 0; b0 < n_tiles; b0 += 1024) {
        const int64_t k = b0 + threadIdx.x;
        const unsigned long long v = k < n_tiles ? tile_cnt[k] : 0ull;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned long long a = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0ull;
            __syncthreads();
            s[threadIdx.x] += a;
            __syncthreads();
        }
        if (k < n_tiles) tile_cnt[k] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
static __global__ __launch_bounds__(256) void nk_subs_mark_kernel(const int64_t* __restrict__ ranks, int64_t k, int64_t n_ranks, uint8_t* __restrict__ mark,
                                                                  unsigned long long* bad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = ranks[i];
        if (r < 0 || r >= n_ranks) atomicAdd(bad, 1ull);
        else mark[r] = 1;
    }
}
static __global__ __launch_bounds__(256) void nk_subs_apply_kernel(const uint8_t* __restrict__ valid, int64_t n, const unsigned long long* __restrict__ tile_off,
                                                                   const uint8_t* __restrict__ mark, uint8_t* __restrict__ inl_out) {
    const int64_t t0 = (int64_t)blockIdx.x * SUBS_TILE + (int64_t)threadIdx.x * 16;
    uint8_t v[16];
    int c = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        v[k] = (t0 + k < n) ? valid[t0 + k] : (uint8_t)0;
        c += v[k] ? 1 : 0;
    }
    __shared__ int s[256];
    s[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int a = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += a;
        __syncthreads();
    }
    unsigned long long r = tile_off[blockIdx.x] + (unsigned long long)(s[threadIdx.x] - c);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (t0 + k < n) {
            uint8_t o = 0;
            if (v[k]) { o = mark[r]; ++r; }
            inl_out[t0 + k] = o;
        }
    }
}

NkGeom geom_of(const xdemhip_nk_plan* P, double dr, double dc) {
    NkGeom g;
    g.H = P->H; g.W = P->W; g.roff = P->roff; g.dr = dr; g.dc = dc; g.rule = P->nan_rule;
    return g;
}

// aux variables + valid mask for this rank's rows; global valid count through the hook
template <typename T> int nk_aux_typed(xdemhip_nk_plan* P) {
    xdemhip_ctx* ctx = P->ctx;
    const int64_t rows = P->row1 - P->row0;
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(P->scratch) + OFF_STATS);
    XD_HIP_CHECK(ctx, hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
    if (rows > 0) {
        hipLaunchKernelGGL((nk_aux_kernel<T>), grid2d(ctx, P->W, rows), dim3(256), 0, ctx->stream,
                           static_cast<const T*>(P->ref), static_cast<const T*>(P->tba), P->inlier, geom_of(P, 0.0, 0.0),
                           static_cast<T*>(P->slope_tan), static_cast<T*>(P->aspect), P->valid, d_cnt, P->row0, P->row1);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    int rc = xd_allreduce_device(ctx, d_cnt, 1, XDEMHIP_RED_SUM_U64);
    if (rc) return rc;
    unsigned long long c = 0;
    { const int rc_ = xd_d2h(ctx, &c, d_cnt, 8); if (rc_) return rc_; }
    { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
    P->n_valid0 = (long long)c;
    // EXT route (see nk_ext_build_kernel): single-GPU plans only -- with a reduction hook every rank would have to agree on
    // the route at every step; the sharded plans keep reading mask and aspect
    // Partitioned plans (round 5, second half): every rank lists ITS extreme-aspect pixels against the same thresholds (the valid
    // count is the global one), min / max aspect of a step are the min / max over the ranks' survivors, and the lists are usable if
    // no rank's overflowed and each kind has a pixel on SOME rank -- agreed once, here.
    P->ext_ok = false;
    const bool mr = ctx->allreduce != nullptr;
    if (mr && !(ctx->nk_fused_dist != 0 && ctx->world >= 1 && ctx->world <= MR_WORLD_MAX)) return XDEMHIP_OK;
    if (P->n_valid0 >= 8 * (long long)EXT_TARGET) {   // (the same decision on every rank)
        const bool local_ok = P->ref_m != nullptr && rows > 0;
        unsigned long long ec[2] = {0, 0};
        if (local_ok) {
            const double frac = (double)EXT_TARGET / (double)P->n_valid0;
            const T thr_lo = (T)(6.283185307179586 * frac), thr_hi = (T)(6.283185307179586 * (1.0 - frac));
            const int64_t q0 = (P->row0 - P->roff) * P->W, nown = rows * P->W;
            XD_HIP_CHECK(ctx, hipMemsetAsync(P->ext_cnt, 0, 32, ctx->stream));
            hipLaunchKernelGGL((nk_ext_build_kernel<T>), dim3(grid_for(ctx, nown, 256, 16)), dim3(256), 0, ctx->stream, static_cast<const T*>(P->ref),
                               P->valid, static_cast<const T*>(P->aspect), q0, nown, thr_lo, thr_hi, static_cast<T*>(P->ref_m), P->ext_idx, P->ext_cnt);
            XD_HIP_CHECK(ctx, hipGetLastError());
            { const int rc_ = xd_d2h(ctx, ec, P->ext_cnt, 16); if (rc_) return rc_; }
            { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
        }
        // (aspects far from uniform -- a tilted plane, a raster of one slope direction -- leave a list empty or overfull)
        const bool over = ec[0] > (unsigned long long)EXT_CAP || ec[1] > (unsigned long long)EXT_CAP;
        if (mr) {
            uint64_t v[3] = {ec[0], ec[1], (uint64_t)((!local_ok || over) ? 1 : 0)};
            const int rc_ = xd_allreduce_host(ctx, v, 3, XDEMHIP_RED_SUM_U64);
            if (rc_) return rc_;
            P->ext_ok = v[2] == 0 && v[0] >= 1 && v[1] >= 1;
        } else {
            P->ext_ok = local_ok && !over && ec[0] >= 1 && ec[1] >= 1;
        }
    }
    return XDEMHIP_OK;
}

// Least-squares sums of the un-binned fit (NuthKaab(bin_before_fit=False): curve_fit of a cos(b - x) + c on every point,
// xdem/coreg/base.py:975-989).  The model is linear in (A, B, c) = (a cos b, a sin b, c): y = A cos x + B sin x + c, so the
// optimum curve_fit converges to follows from nine float64 sums; x = aspect and y = (dh - vshift) / slope_tan are widened
// to float64 first, as curve_fit does with its inputs.  sums: n, Sc, Ss, Scc, Sss, Scs, Sy, Syc, Sys, Syy.
template <typename T>
__global__ __launch_bounds__(256) void nk_fit_sums_kernel(const T* __restrict__ dh, const T* __restrict__ slope_tan,
                                                          const T* __restrict__ aspect, int64_t n, const T* __restrict__ vshift_p,
                                                          double* sums /* [10] */) {
    const T vshift = *vshift_p;
    double a[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const T d = dh[p];
        const T st = slope_tan[p];
        const T x = aspect[p];
        if (d == d) {
            const double y = (double)t_div(t_sub(d, vshift), st);
            const double c = cos((double)x), sn = sin((double)x);
            a[0] += 1.0; a[1] += c; a[2] += sn; a[3] += c * c; a[4] += sn * sn; a[5] += c * sn;
            a[6] += y; a[7] += y * c; a[8] += y * sn; a[9] += y * y;
        }
    }
    __shared__ double blk[10];
    if (threadIdx.x < 10) blk[threadIdx.x] = 0.0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        double v = a[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&blk[k], v);
    }
    __syncthreads();
    if (threadIdx.x < 10) atomicAdd(&sums[threadIdx.x], blk[threadIdx.x]);
}

// Stages 1 + 2 of a step on the PLAIN route (the fall-back of the one-pass step, and the route of the mean statistic, of the un-binned
// fit and of small rasters): dh at the shifted position (written), min / max aspect over its finite pixels, the exact nanmedian of
// dh.  On return the device holds dh, the vertical shift (OFF_INFO), the bin edges (scratch start; the plan's custom edges if set)
// and, at OFF_INFO + 8, the valid count.
template <typename T>
int nk_stage_a(xdemhip_nk_plan* P, const NkGeom& g, int64_t q0, int64_t n, int nb) {
    typedef typename KeyT<T>::type K;
    xdemhip_ctx* ctx = P->ctx;
    unsigned char* base = static_cast<unsigned char*>(P->scratch);
    DhStats* d_stats = reinterpret_cast<DhStats*>(base + OFF_STATS);
    T* d_edges = reinterpret_cast<T*>(base);
    DhStats hs0;
    hipLaunchKernelGGL(nk_stats_init_kernel, dim3(1), dim3(64), 0, ctx->stream, d_stats);
    XD_HIP_CHECK(ctx, hipGetLastError());
    int rc = XDEMHIP_OK;
    {
        if (n > 0) {
            hipLaunchKernelGGL((nk_dh_kernel<T>), grid2d(ctx, P->W, P->row1 - P->row0), dim3(256), 0, ctx->stream,
                               static_cast<const T*>(P->ref), static_cast<const T*>(P->tba), P->valid, static_cast<const T*>(P->aspect),
                               g, static_cast<T*>(P->dh), d_stats, P->row0, P->row1);
            XD_HIP_CHECK(ctx, hipGetLastError());
        }
        rc = xd_allreduce_device(ctx, &d_stats->asp_min, 1, XDEMHIP_RED_MIN_U64);
        if (rc) return rc;
        rc = xd_allreduce_device(ctx, &d_stats->asp_max, 1, XDEMHIP_RED_MAX_U64);
        if (rc) return rc;
        std::vector<SelResult<K>> g0;
        rc = run_select<T>(ctx, static_cast<const T*>(P->dh) + q0, nullptr, n, 1, base, g0, &P->ws);
        if (rc) return rc;
        { const int rc_ = xd_d2h(ctx, &hs0, d_stats, sizeof hs0); if (rc_) return rc_; }
        { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
        unsigned char info[32] = {0};
        const T vs_t = (T)median_from<T>(g0[0]);
        const uint64_t total = g0[0].st.count, flags = 0;
        const double vs = g0[0].st.count ? (double)vs_t : (double)NAN;
        memcpy(info, &vs_t, sizeof(T));
        memcpy(info + 8, &total, 8);
        memcpy(info + 16, &flags, 8);
        memcpy(info + 24, &vs, 8);
        std::vector<T> edges;
        make_edges<T>((double)val_of((K)hs0.asp_min), (double)val_of((K)hs0.asp_max), nb, edges);
        XD_HIP_CHECK(ctx, hipMemcpyAsync(base + OFF_INFO, info, 32, hipMemcpyHostToDevice, ctx->stream));
        XD_HIP_CHECK(ctx, hipMemcpyAsync(d_edges, edges.data(), sizeof(T) * (nb + 1), hipMemcpyHostToDevice, ctx->stream));
        { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }  // (info / edges are stack / local buffers)
    }
    if (!P->custom_edges.empty()) {  // explicit bin edges (NuthKaab(bin_sizes=<edges>)): SciPy casts them to the sample dtype
        std::vector<T> e(P->custom_edges.size());
        for (size_t k = 0; k < e.size(); ++k) e[k] = (T)P->custom_edges[k];
        XD_HIP_CHECK(ctx, hipMemcpyAsync(d_edges, e.data(), sizeof(T) * e.size(), hipMemcpyHostToDevice, ctx->stream));
        { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
    }
    return XDEMHIP_OK;
}


// exchange buffers of the one-pass step on partitioned plans, sized for `nb` bins and `world` ranks (see the nk_mr_* kernels)
int nk_mr_alloc(xdemhip_nk_plan* P, int nb, int world, size_t es) {
    auto mx = [](size_t a, size_t b) { return a > b ? a : b; };
    const size_t wa = mx(mx(4, 3 + 3 * (size_t)nb + 5 * (size_t)world + (size_t)world * (DSEL_BUCKETS / 2)), 2 + 2 * (size_t)nb + (size_t)world * nb * (SEL_RADIX / 2));
    const size_t wb = mx((size_t)world * DSEL_HDR_WORDS + (size_t)DSEL_CAP * es / 8, (size_t)nb * MR_GSEG * es / 8);
    if (!P->mr_small && hipMalloc(reinterpret_cast<void**>(&P->mr_small), (size_t)(11 + BINSEG_CTR_STRIDE) * P->ws.nb_max * 8) != hipSuccess) {
        (void)hipGetLastError();
        P->mr_small = nullptr;
        return XDEMHIP_ENOMEM;
    }
    if (P->mr_a_words < wa) {
        if (P->mr_a) (void)hipFree(P->mr_a);
        P->mr_a = nullptr; P->mr_a_words = 0;
        if (hipMalloc(reinterpret_cast<void**>(&P->mr_a), wa * 8) != hipSuccess) { (void)hipGetLastError(); P->mr_a = nullptr; return XDEMHIP_ENOMEM; }
        P->mr_a_words = wa;
    }
    if (P->mr_b_words < wb) {
        if (P->mr_b) (void)hipFree(P->mr_b);
        P->mr_b = nullptr; P->mr_b_words = 0;
        if (hipMalloc(reinterpret_cast<void**>(&P->mr_b), wb * 8) != hipSuccess) { (void)hipGetLastError(); P->mr_b = nullptr; return XDEMHIP_ENOMEM; }
        P->mr_b_words = wb;
    }
    return XDEMHIP_OK;
}

// ---- host side of the one-pass step (device code: "Round 4: the ONE-PASS step" above) ------------------------------------------
// *done = false (nothing returned) when the route does not apply or when a bracket missed / a buffer overflowed: the caller
// then runs the plain route, which needs nothing from here.
template <typename T>
int nk_step_onepass(xdemhip_nk_plan* P, const NkGeom& g, int64_t q0, int64_t n, int nb, bool* done, double* vshift, int64_t* n_valid,
                    double* y_mean, double* y_std, double* edges_out, int64_t* counts, double* medians, bool allow_predict = true) {
    typedef typename KeyT<T>::type K;
    *done = false;
    xdemhip_ctx* ctx = P->ctx;
    SelWorkspace* ws = &P->ws;
    const int64_t rows = P->row1 - P->row0;
    const int64_t n_slots = ((((n + SEL_LINE - 1) >> SEL_LINE_LOG2) + 63) >> 6) << SEL_LINE_LOG2;
    const bool mr = ctx->allreduce != nullptr;   // a partitioned plan: this rank's rows, every count and order statistic over all ranks
    const int world = ctx->world, rank = ctx->rank;
    bool cannot = !ctx->nk_fused || !P->fz || !P->cd_vals || !P->c_st || !P->bcache || !P->ext_ok || (g.rule > 1 && !P->badbits) || rows <= 0 ||
                  P->bin_stat != XDEMHIP_BINSTAT_MEDIAN || !(ctx->selection_mode == 0 || ctx->selection_mode == 3) || !ws->d_small ||
                  (int64_t)P->nbuf * P->W < ws->c_cap ||   // (the kept bin candidates go into per-bin segments of the y raster)
                  ws->es != sizeof(T) || nb > ws->nb_max || nb > MAX_BINS_PER_SWEEP || n_slots > ws->s_cap ||
                  (int64_t)(NKZ_CHUNK_MAX + 2) * P->W * (int64_t)sizeof(T) >= ((int64_t)1 << 32);   // (32-bit byte offsets inside a chunk of rows)
    int64_t n_all = n;   // pixels of all ranks
    if (mr && (world < 1 || !ctx->nk_fused_dist || !ctx->nk_fused)) return XDEMHIP_OK;   // (not told the ranks / switched off: context-level, the same on every rank)
    if (mr) {
        // Every rank must take the same route.  What the decision rests on is either fixed for the plan and its options or derives
        // from reduced data (ext_ok), so the ranks agree ONCE -- a host all-reduce of (own pixels, "I cannot") -- and again whenever
        // any of it changes, which it does on all ranks at the same step.
        const int64_t key[8] = {nb, P->row0, P->row1, (int64_t)P->ext_ok, (int64_t)P->bin_stat, (int64_t)ctx->selection_mode,
                                (int64_t)(ctx->nk_fused * 4 + ctx->nk_fused_dist * 2), (int64_t)world * 64 + rank};
        if (memcmp(key, P->mr_key, sizeof key) != 0) {
            cannot = cannot || !ctx->nk_fused_dist || world < 1 || world > MR_WORLD_MAX || rank < 0 || rank >= world ||
                     (int64_t)P->nbuf * P->W < ws->c_cap;
            if (!cannot && nk_mr_alloc(P, nb, world, sizeof(T)) != XDEMHIP_OK) cannot = true;
            uint64_t v[2] = {(uint64_t)n, (uint64_t)(cannot ? 1 : 0)};
            const int rc_ = xd_allreduce_host(ctx, v, 2, XDEMHIP_RED_SUM_U64);
            if (rc_) return rc_;
            P->mr_nglob = (int64_t)v[0];
            P->mr_can = v[1] == 0;
            memcpy(P->mr_key, key, sizeof key);
        }
        if (!P->mr_can) return XDEMHIP_OK;
        n_all = P->mr_nglob;
    } else if (cannot) {
        return XDEMHIP_OK;
    }
    if (n_all < SEL_BRACKET_MIN_N || (ctx->selection_mode == 0 && n_all < SEL_BRACKET_MIN_PER_BIN * nb)) return XDEMHIP_OK;
    unsigned char* scratch = static_cast<unsigned char*>(P->scratch);
    T* d_edges = reinterpret_cast<T*>(scratch);
    DhStats* d_stats = reinterpret_cast<DhStats*>(scratch + OFF_STATS);
    BinCacheRec* d_rec = reinterpret_cast<BinCacheRec*>(scratch + OFF_INFO + 64);
    const SelState<K>* d_st = reinterpret_cast<const SelState<K>*>(scratch + OFF_STATE);
    const int nbm = ws->nb_max;
    uint64_t* fz = P->fz;
    unsigned long long* ctr = reinterpret_cast<unsigned long long*>(fz);   // [1] dh candidates, [2] overflow, [3] miss, [5] y candidates, [6] no survivor in an EXT list
    uint32_t* rbs_d = reinterpret_cast<uint32_t*>(fz + 8);
    uint32_t* rbs_y = reinterpret_cast<uint32_t*>(fz + 9);
    uint64_t* cnt_d = fz + 10;
    // (fz + 13: ranks of the dh selection, written and read on the device only)
    K* klo_d = reinterpret_cast<K*>(fz + 14);
    K* khi_d = reinterpret_cast<K*>(fz + 15);
    T* d_vhat = reinterpret_cast<T*>(fz + 16);
    T* d_delta = reinterpret_cast<T*>(fz + 17);
    double* d_sums = reinterpret_cast<double*>(fz + 18);
    K* klo_y = reinterpret_cast<K*>(fz + 24);
    K* khi_y = reinterpret_cast<K*>(fz + 24 + nbm);
    // (fz + 24 + 2 * nbm: ranks of the bin selections, device only)
    uint64_t* cls_y = fz + 24 + 3 * nbm;
    uint64_t* res_y = fz + 24 + 6 * nbm;
    uint64_t* cnt_y = fz + 24 + 8 * nbm;
    const bool custom = !P->custom_edges.empty();
    const int last_decimal = custom ? P->custom_decimal : NK_AUTO_EDGES;
    if (custom) {  // explicit bin edges: SciPy casts them to the sample dtype (rare path: a blocking copy)
        std::vector<T> e(P->custom_edges.size());
        for (size_t k = 0; k < e.size(); ++k) e[k] = (T)P->custom_edges[k];
        XD_HIP_CHECK(ctx, hipMemcpyAsync(d_edges, e.data(), sizeof(T) * e.size(), hipMemcpyHostToDevice, ctx->stream));
        { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
    }
    const T* ref_m = static_cast<const T*>(P->ref_m);
    const T* tba = static_cast<const T*>(P->tba);
    const T* st_all = static_cast<const T*>(P->slope_tan);
    static_assert(MAX_BINS_PER_SWEEP <= NK_BINCACHE_MAX_BINS, "the one-pass step's bins fit the byte-wide cache");
    // 1. min / max aspect of this step from the EXT lists -> edges, freshness of the bin cache; (re)fill of the cache
    hipLaunchKernelGGL(nk_step_init_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_stats, fz, (int64_t)(P->fz_bytes / 8), P->ext_cnt + 2);
    hipLaunchKernelGGL((nk_ext_eval_kernel<T>), dim3(EXT_CAP / 256, 2), dim3(256), 0, ctx->stream, ref_m, tba, static_cast<const T*>(P->aspect), g,
                       P->ext_idx, P->ext_cnt, d_stats, P->ext_cnt + 2);
    int rc = XDEMHIP_OK;
    // (partitioned plans: min / max aspect and the survivors of all ranks arrive with the first histogram all-reduce of the dh sample's
    //  selection below -- per-rank slots behind its two histograms -- so the edges and the bin cache, which only the y^ sample and the
    //  pass need, follow that selection)
    uint64_t* ext_slots = reinterpret_cast<uint64_t*>(scratch + off_hist(2)) + 2 * SEL_RADIX;
    if (mr) hipLaunchKernelGGL(nk_mr_ext_pack_kernel, dim3(1), dim3(64), 0, ctx->stream, d_stats, P->ext_cnt + 2, rank, world, ext_slots);
    auto edges_and_bins = [&]() -> int {
        hipLaunchKernelGGL((nk_fz_prep_kernel<T>), dim3(1), dim3(64), 0, ctx->stream, d_stats, P->ext_cnt + 2, nb, (int)custom, d_edges, d_rec,
                           (int)P->bcache_force, ctr);
        hipLaunchKernelGGL((nk_bin_fill_kernel<T>), dim3(grid_for(ctx, n, 256, 16)), dim3(256), sizeof(T) * (nb + 1), ctx->stream,
                           static_cast<const T*>(P->aspect) + q0, n, d_edges, nb, last_decimal, d_rec, P->bcache + q0);
        hipLaunchKernelGGL((nk_bin_cache_commit_kernel<T>), dim3(1), dim3(64), 0, ctx->stream, d_edges, nb, d_rec);
        XD_HIP_CHECK(ctx, hipGetLastError());
        P->bcache_force = false;
        return XDEMHIP_OK;
    };
    if (!mr) { rc = edges_and_bins(); if (rc) return rc; }
    // Round 6: brackets PREDICTED from the previous step instead of sampled.  A bin median of y = (dh - vshift) / slope_tan moves by
    // -(dE sin(aspect) + dN cos(aspect)) when the tap position moves by (dE, dN) pixels -- the Nuth-Kaab model itself -- and the
    // median of dh stays to first order; how well that held is MEASURED at every step (pr_err: the worst miss of the centres in
    // half widths of the sampled brackets, for the step just finished), and a step is predicted only when the last one would have
    // been hit with room to spare and the shift change has not grown.  The brackets are then 0.25 ... 1 sampled widths around the
    // predicted centres; the integer counts of the pass prove them like any sampled bracket, a miss reruns THIS step sampled
    // (and the next one): exact either way.  Identical decisions on every rank of a partitioned plan (everything is derived from
    // reduced results and the call's arguments).
    const double dE = g.dc - P->pr_sx, dN = -(g.dr - P->pr_sy);   // (pr_sx / pr_sy hold the previous tap offsets in pixels)
    const double dpx = sqrt(dE * dE + dN * dN);
    // (the measured miss of the centres is proportional to the shift change -- 0.90 / 0.11 / 0.012 half widths at 0.2 / 0.02 / 0.002 px
    //  on C3: the model's error is its second-order term -- so the miss to EXPECT at this step is the last one scaled by the ratio of
    //  the shift changes, with a floor for the jitter of the medians themselves)
    const bool pr_scaled = P->pr_dpx > 0 && P->pr_dpx < 1e29;
    const double pr_expect = pr_scaled ? fmax(0.02, P->pr_err * (dpx / P->pr_dpx)) : 1e30;      // bin medians
    const double pr_expect_d = pr_scaled ? fmax(0.02, P->pr_err_d * (dpx / P->pr_dpx)) : 1e30;  // median of dh (assumed not to move)
    bool predict = allow_predict && ctx->nk_predict != 0 && P->pr_have && P->pr_nb == nb && nb <= NK_PREDICT_MAX_BINS && P->pr_cooldown == 0 &&
                   pr_expect <= 0.20 && pr_expect_d <= 0.20 && dpx <= 0.05 && P->pr_wd > 0;
    // ... and on steps that still move too far for the bins, the bracket of the MEDIAN OF DH alone may be predicted: that median
    // hardly follows the shift (0.19 / 0.29 sampled half widths off at 0.1 / 0.5 px on C3), and without its three digit passes the y^
    // sample can be taken straight away (the bins' brackets are sampled as ever)
    const bool predict_d = !predict && allow_predict && ctx->nk_predict != 0 && P->pr_have && P->pr_nb == nb && P->pr_cooldown == 0 &&
                           pr_expect_d <= 0.35 && dpx <= 0.6 && P->pr_wd > 0;
    if (P->pr_cooldown > 0 && allow_predict) --P->pr_cooldown;
    double pr_h = 1.0;   // bracket half widths of this step in sampled half widths
    T* s_v = static_cast<T*>(ws->s_vals);
    bool fused = false;
    constexpr int BR_PASSES = 3;
    const K low_mask = (K)(((K)1 << (8 * (KeyT<T>::passes - BR_PASSES))) - 1);
    const int narrow = P->fz_narrow;
    if (predict) {
        pr_h = fmin(1.0, fmax(0.25, 2.0 * pr_expect + 0.15));
        NkPredicted<T> pr;
        // (the bracket of the median of dh may not be padded: its half width is the margin delta / slope_tan of EVERY pixel's y, and on
        //  flat ground a centimetre of it turns whole waves into bin candidates -- session r06h: overflow flag at 0.04 px)
        const double hd = fmin(1.0, fmax(0.25, 2.0 * pr_expect_d + 0.15)) * P->pr_wd;
        pr.dlo = (T)(P->pr_v - hd);
        pr.dhi = (T)(P->pr_v + hd);
        for (int b = 0; b < nb; ++b) {
            if (!P->pr_ok[b]) { pr.lo[b] = (T)1; pr.hi[b] = (T)0; continue; }   // (an empty bin stays empty: the aspects do not move)
            const double c = P->pr_med[b] - (dE * sin(P->pr_mid[b]) + dN * cos(P->pr_mid[b]));
            const double h = pr_h * P->pr_w[b] + 0.05 * dpx;
            pr.lo[b] = (T)(c - h);
            pr.hi[b] = (T)(c + h);
        }
        if (mr) {   // min / max aspect and the EXT survivors of all ranks: their own small exchange (they ride on the dh sample's first histogram otherwise)
            rc = xd_allreduce_device(ctx, ext_slots, 4 * (int64_t)world, XDEMHIP_RED_SUM_U64);
            if (rc) return rc;
            hipLaunchKernelGGL(nk_mr_ext_unpack_kernel, dim3(1), dim3(64), 0, ctx->stream, ext_slots, world, d_stats, P->ext_cnt + 2);
            rc = edges_and_bins();
            if (rc) return rc;
        }
        hipLaunchKernelGGL((nk_predict_kernel<T>), dim3(1), dim3(64), 0, ctx->stream, pr, nb, klo_d, khi_d, rbs_d, d_vhat, d_delta, klo_y, khi_y, rbs_y, ctr);
        XD_HIP_CHECK(ctx, hipGetLastError());
    } else {
    // 2. sample of dh -> bracket of its median, v^, delta
    // (round 5: the sample kernels also reset the selection that runs on their sample -- select_reset_slice)
    if (!predict_d) {
        hipLaunchKernelGGL((nk_sample_dh_kernel<T>), dim3(grid_for(ctx, n_slots, 256, 8)), dim3(256), 0, ctx->stream, ref_m, tba, g, q0, n,
                           1.0 / (double)P->W, n_slots, s_v, select_reset_plan<K>(scratch, 1, SEL_BRACKET_DUAL));
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    if (predict_d) {
        // no dh sample in memory and no selection on it: the bracket comes from the previous median, and the y^ sample is formed
        // straight from the rasters (nk_sample_dy_kernel below)
        NkPredicted<T> pr;
        const double hd = fmin(1.0, fmax(0.25, 2.0 * pr_expect_d + 0.15)) * P->pr_wd;
        pr.dlo = (T)(P->pr_v - hd);
        pr.dhi = (T)(P->pr_v + hd);
        if (mr) {
            rc = xd_allreduce_device(ctx, ext_slots, 4 * (int64_t)world, XDEMHIP_RED_SUM_U64);
            if (rc) return rc;
        }
        hipLaunchKernelGGL((nk_predict_kernel<T>), dim3(1), dim3(64), 0, ctx->stream, pr, 0, klo_d, khi_d, rbs_d, d_vhat, d_delta, klo_y, khi_y, rbs_y, ctr);
        XD_HIP_CHECK(ctx, hipGetLastError());
    } else {
    // (round 5: the passes advance their own states and the last one writes the bracket ends -- hist_pass_kernel<T, true>; `fused`
    //  tells whether that form ran)
    rc = select_enqueue<T>(ctx, s_v, nullptr, n_slots, n_slots, nullptr, 1, scratch, SEL_BRACKET_DUAL, nullptr, BR_PASSES, false, nullptr, nullptr,
                           false, narrow, klo_d, khi_d, rbs_d, low_mask, &fused, true, mr ? 0 : -1, mr ? 4 * (int64_t)world : 0);
    if (rc) return rc;
    if (!fused) hipLaunchKernelGGL((bracket_finish_kernel<K>), dim3(1), dim3(64), 0, ctx->stream, d_st, 1, 0, low_mask, klo_d, khi_d, rbs_d);
    }
    if (mr) {
        hipLaunchKernelGGL(nk_mr_ext_unpack_kernel, dim3(1), dim3(64), 0, ctx->stream, ext_slots, world, d_stats, P->ext_cnt + 2);
        rc = edges_and_bins();
        if (rc) return rc;
    }
    // 3. sample of y^ per aspect bin -> brackets of the bin medians (round 5: v^ and delta formed by the sample kernel itself)
    if (predict_d)
        hipLaunchKernelGGL((nk_sample_dy_kernel<T>), dim3(grid_for(ctx, n_slots, 256, 8)), dim3(256), 0, ctx->stream, ref_m, tba, g, q0, n,
                           1.0 / (double)P->W, n_slots, s_v, ws->s_bins, st_all + q0, P->bcache + q0, klo_d, khi_d, d_vhat, d_delta, ctr,
                           select_reset_plan<K>(scratch, nb, SEL_BRACKET_DUAL));
    else
    hipLaunchKernelGGL((nk_sample_y_kernel<T>), dim3(grid_for(ctx, n_slots, 256, 8)), dim3(256), 0, ctx->stream, s_v, ws->s_bins, st_all + q0,
                       P->bcache + q0, n, n_slots, d_vhat, klo_d, khi_d, d_vhat, d_delta, ctr, select_reset_plan<K>(scratch, nb, SEL_BRACKET_DUAL));
    XD_HIP_CHECK(ctx, hipGetLastError());
    rc = select_enqueue<T>(ctx, s_v, nb == 1 ? nullptr : ws->s_bins, n_slots, n_slots, nullptr, nb, scratch, SEL_BRACKET_DUAL, nullptr, BR_PASSES,
                           false, nullptr, nullptr, false, narrow, klo_y, khi_y, rbs_y, low_mask, &fused, true);
    if (rc) return rc;
    if (!fused) hipLaunchKernelGGL((bracket_finish_kernel<K>), dim3(1), dim3(64), 0, ctx->stream, d_st, nb, 0, low_mask, klo_y, khi_y, rbs_y);
    XD_HIP_CHECK(ctx, hipGetLastError());
    }   // (sampled brackets)
    // 4. the one pass
    int n_wg = 0;
    {
        dim3 grid = grid2d(ctx, P->W, rows);
        if ((rows + grid.y - 1) / grid.y > NKZ_CHUNK_MAX) grid.y = (unsigned)((rows + NKZ_CHUNK_MAX - 1) / NKZ_CHUNK_MAX);
        n_wg = (int)(grid.x * grid.y);
        if ((size_t)n_wg > P->wg_sums_cap) {   // (per-workgroup slots of the pass's five float64 sums: nk_sums_reduce)
            if (P->wg_sums) (void)hipFree(P->wg_sums);
            P->wg_sums = nullptr; P->wg_sums_cap = 0;
            if (hipMalloc(reinterpret_cast<void**>(&P->wg_sums), (size_t)n_wg * 5 * sizeof(double)) != hipSuccess) {
                (void)hipGetLastError();
                return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc(per-workgroup sums of the one-pass step) failed");
            }
            P->wg_sums_cap = (size_t)n_wg;
        }
        int copies = (4608) / (nb * 12);   // (static 26 KB + this: five workgroups of 256 threads per CU)
        copies = copies < 1 ? 1 : (copies > 16 ? 16 : copies);
        const size_t lds = (size_t)nb * sizeof(FzPair<T>) + (size_t)((3 * nb) | 1) * 4 * (size_t)copies + 64 * 4;
        T* cy_d = static_cast<T*>(ws->c_vals);
#define XD_NK_FZ(RULE)                                                                                                               \
    hipLaunchKernelGGL((nk_fused_kernel<T, RULE>), grid, dim3(256), lds, ctx->stream, ref_m, tba, st_all, P->bcache, g, P->row0, P->row1,  \
                       P->nbuf, nb, copies, klo_d, khi_d, d_vhat, d_delta, klo_y, khi_y, cnt_d, cls_y, static_cast<T*>(P->cd_vals),   \
                       P->cd_cap, cy_d, static_cast<T*>(P->c_st), ws->c_bins, ws->c_cap, ctr, P->wg_sums, P->badbits, P->bad_wpr)
        if (g.rule == 0) XD_NK_FZ(0);
        else if (g.rule == 1) XD_NK_FZ(1);
        else XD_NK_FZ(2);
#undef XD_NK_FZ
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    // (partitioned plans: this rank's classes, the counters rewritten for the gathered bucket values, the true counters of the bins)
    uint64_t* cls_loc = mr ? P->mr_small : nullptr;
    uint64_t* cls_f = mr ? cls_loc + 3 * nbm : nullptr;
    uint64_t* res_f = mr ? cls_f + 3 * nbm : nullptr;
    uint64_t* cnt_true = mr ? res_f + 2 * nbm : nullptr;
    unsigned long long* segf_ctr = mr ? reinterpret_cast<unsigned long long*>(cnt_true + 3 * nbm) : nullptr;   // [nb] x BINSEG_CTR_STRIDE words
    // (few workgroups for the histogram / gather of the dh candidates: every one of them ends with an atomic on ONE word, which
    //  serialise -- 256 workgroups with an atomic per wave spent 80 us there)
    int dsel_grid = grid_for(ctx, n / 32 + 1, HIST_THREADS * 8, 1);
    dsel_grid = dsel_grid > 64 ? 64 : dsel_grid;
    const int64_t words7 = 3 + 3 * (int64_t)nb + 5 * (int64_t)world;
    uint32_t* mr_rows = mr ? reinterpret_cast<uint32_t*>(P->mr_a + words7) : nullptr;   // [world][DSEL_BUCKETS]
    if (mr) {
        // exchange 7: the pass's counters summed, the five float64 sums of every rank gathered and added in rank order -- and, in the
        // same all-reduce, the 4096-bucket histogram of this rank's dh candidates in its own row (the candidates and the bracket
        // are there once the pass is through; only the choice of the bucket needs the summed counters)
        XD_HIP_CHECK(ctx, hipMemsetAsync(mr_rows, 0, (size_t)world * DSEL_BUCKETS * 4, ctx->stream));
        hipLaunchKernelGGL(nk_mr_counts_pack_kernel, dim3(1), dim3(256), 0, ctx->stream, cnt_d, cls_y, P->wg_sums, n_wg, nb, rank, world, P->mr_a, cls_loc);
        hipLaunchKernelGGL((nk_dhsel_hist_kernel<T>), dim3(dsel_grid), dim3(HIST_THREADS), 0, ctx->stream, static_cast<const T*>(P->cd_vals), P->cd_cap, ctr + 1,
                           klo_d, khi_d, mr_rows + (size_t)rank * DSEL_BUCKETS);
        XD_HIP_CHECK(ctx, hipGetLastError());
        rc = xd_allreduce_device(ctx, P->mr_a, words7 + (int64_t)world * (DSEL_BUCKETS / 2), XDEMHIP_RED_SUM_U64);
        if (rc) return rc;
        hipLaunchKernelGGL(nk_mr_counts_unpack_kernel, dim3(1), dim3(256), 0, ctx->stream, P->mr_a, nb, world, cnt_d, cls_y, d_sums);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    // 5. exact median of dh among its candidates -> vshift
    {   // value buckets of the bracket, three launches (nk_dhsel_* above)
        uint32_t* dsel = reinterpret_cast<uint32_t*>(fz + 24 + (11 + BINSEG_CTR_STRIDE) * nbm);
        K* dsel_keys = reinterpret_cast<K*>(reinterpret_cast<unsigned char*>(fz) + P->fz_bytes);
        const int grid = dsel_grid;
        const size_t lds = (size_t)DSEL_CAP * sizeof(K) + (size_t)(BINSEL_COPIES * (SEL_RADIX + 1) + 1) * 4 + (size_t)(SEL_RADIX + 4 + 16 + 2) * 8;
        rc = set_big_lds(ctx, nk_dhsel_final_kernel<T>, lds);
        if (rc) return rc;
        if (mr) {
            // (the rows of exchange 7) -> the global histogram, the bucket, this rank's offset in the bucket's key list
            hipLaunchKernelGGL((nk_mr_dh_base_kernel<T>), dim3(1), dim3(HIST_THREADS), 0, ctx->stream, mr_rows, world, rank, cnt_d, klo_d, khi_d, dsel, ctr);
            // exchange 8: the bucket's keys of all ranks, each at its offset, + every rank's header words
            uint64_t* slots = P->mr_b;
            K* gk = reinterpret_cast<K*>(P->mr_b + (size_t)world * DSEL_HDR_WORDS);
            const int64_t words8 = (int64_t)world * DSEL_HDR_WORDS + (int64_t)DSEL_CAP * (int64_t)sizeof(K) / 8;
            XD_HIP_CHECK(ctx, hipMemsetAsync(P->mr_b, 0, (size_t)words8 * 8, ctx->stream));
            hipLaunchKernelGGL((nk_dhsel_gather_kernel<T>), dim3(grid), dim3(HIST_THREADS), 0, ctx->stream, static_cast<const T*>(P->cd_vals), P->cd_cap, ctr + 1,
                               cnt_d, klo_d, khi_d, dsel, gk, ctr, 0);
            hipLaunchKernelGGL(nk_mr_dh_hdr_pack_kernel, dim3(1), dim3(64), 0, ctx->stream, dsel, rank, slots);
            XD_HIP_CHECK(ctx, hipGetLastError());
            rc = xd_allreduce_device(ctx, P->mr_b, words8, XDEMHIP_RED_SUM_U64);
            if (rc) return rc;
            hipLaunchKernelGGL(nk_mr_dh_hdr_merge_kernel, dim3(1), dim3(64), 0, ctx->stream, slots, world, dsel);
            hipLaunchKernelGGL((nk_dhsel_final_kernel<T>), dim3(1), dim3(HIST_THREADS), lds, ctx->stream, P->cd_cap, ctr + 1, cnt_d, klo_d, khi_d, dsel, gk,
                               ctr, scratch + OFF_INFO, 0);
        } else {
        hipLaunchKernelGGL((nk_dhsel_hist_kernel<T>), dim3(grid), dim3(HIST_THREADS), 0, ctx->stream, static_cast<const T*>(P->cd_vals), P->cd_cap, ctr + 1,
                           klo_d, khi_d, dsel + 2 * DSEL_HDR_WORDS, P->wg_sums, n_wg, d_sums);
        hipLaunchKernelGGL((nk_dhsel_gather_kernel<T>), dim3(grid), dim3(HIST_THREADS), 0, ctx->stream, static_cast<const T*>(P->cd_vals), P->cd_cap, ctr + 1,
                           cnt_d, klo_d, khi_d, dsel, dsel_keys, ctr);
        hipLaunchKernelGGL((nk_dhsel_final_kernel<T>), dim3(1), dim3(HIST_THREADS), lds, ctx->stream, P->cd_cap, ctr + 1, cnt_d, klo_d, khi_d, dsel, dsel_keys,
                           ctr, scratch + OFF_INFO);
        }
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    // 6. the bin candidates with the exact vshift -> counts, exact medians among those inside the brackets
    {
        // round 5: kept values into per-bin segments (in the y raster, which this route does not use), one workgroup per bin selects
        unsigned long long* seg_ctr = reinterpret_cast<unsigned long long*>(fz + 24 + 11 * nbm);   // [nb] x BINSEG_CTR_STRIDE words
        const size_t lds = (size_t)nb * (3 * 8 + 2 * sizeof(K) + 3 * 4) + 16;
        hipLaunchKernelGGL((nk_resolve_scatter_kernel<T>), dim3(grid_for(ctx, n / 16 + 1, HIST_THREADS * BINSEG_U, 2)), dim3(HIST_THREADS), lds, ctx->stream,
                           static_cast<const T*>(ws->c_vals), static_cast<const T*>(P->c_st), ws->c_bins, ws->c_cap, ctr + 5,
                           reinterpret_cast<const T*>(scratch + OFF_INFO), nb, klo_y, khi_y, cls_y, res_y, seg_ctr, static_cast<T*>(P->y),
                           (int64_t)P->nbuf * P->W, ctr, cls_loc);
        const size_t lds2 = (size_t)BINSEL_KEY_BYTES + (size_t)(BINSEL_COPIES * (SEL_RADIX + 1) + 1) * 4 + (size_t)(SEL_RADIX + 8) * 8;
        rc = set_big_lds(ctx, nk_bin_select_kernel<T>, lds2);
        if (rc) return rc;
        if (mr) {
            // exchange 9: per bin the value-bucket histogram of this rank's segment, one row per rank; the resolved counters; local flags
            const int64_t words9 = 2 + 2 * (int64_t)nb + (int64_t)world * nb * (SEL_RADIX / 2);
            XD_HIP_CHECK(ctx, hipMemsetAsync(P->mr_a, 0, (size_t)words9 * 8, ctx->stream));
            hipLaunchKernelGGL((nk_mr_bin_hist_kernel<T>), dim3(nb), dim3(HIST_THREADS), 0, ctx->stream, static_cast<const T*>(P->y), cls_loc, res_y, seg_ctr, nb, klo_y,
                               khi_y, rank, world, P->mr_a, ctr);
            XD_HIP_CHECK(ctx, hipGetLastError());
            rc = xd_allreduce_device(ctx, P->mr_a, words9, XDEMHIP_RED_SUM_U64);
            if (rc) return rc;
            // exchange 10: per bin the chosen bucket's values of all ranks (+ each rank's smallest value above it)
            const int64_t words10 = (int64_t)nb * MR_GSEG * (int64_t)sizeof(T) / 8;
            XD_HIP_CHECK(ctx, hipMemsetAsync(P->mr_b, 0, (size_t)words10 * 8, ctx->stream));
            hipLaunchKernelGGL((nk_mr_bin_gather_kernel<T>), dim3(nb), dim3(HIST_THREADS), 0, ctx->stream, static_cast<const T*>(P->y), cls_loc, cls_y, P->mr_a, seg_ctr, nb,
                               klo_y, khi_y, rank, world, reinterpret_cast<T*>(P->mr_b), cls_f, res_f, segf_ctr, cnt_true, ctr);
            XD_HIP_CHECK(ctx, hipGetLastError());
            rc = xd_allreduce_device(ctx, P->mr_b, words10, XDEMHIP_RED_SUM_U64);
            if (rc) return rc;
            hipLaunchKernelGGL((nk_bin_select_kernel<T>), dim3(nb), dim3(HIST_THREADS), lds2, ctx->stream, reinterpret_cast<const T*>(P->mr_b), cls_f, res_f, segf_ctr,
                               nb, klo_y, khi_y, rbs_y, reinterpret_cast<SelState<K>*>(scratch + OFF_STATE), reinterpret_cast<uint64_t*>(scratch + off_succ(nb)),
                               cnt_y, ctr, (int64_t)MR_GSEG);
        } else {
        hipLaunchKernelGGL((nk_bin_select_kernel<T>), dim3(nb), dim3(HIST_THREADS), lds2, ctx->stream, static_cast<const T*>(P->y), cls_y, res_y, seg_ctr, nb,
                           klo_y, khi_y, rbs_y, reinterpret_cast<SelState<K>*>(scratch + OFF_STATE), reinterpret_cast<uint64_t*>(scratch + off_succ(nb)),
                           cnt_y, ctr);
        }
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    // 7. everything the step hands back: packed into one block on the device, one copy, one synchronisation
    std::vector<uint64_t> cnt(3 * (size_t)nb);
    std::vector<K> klo(nb);
    std::vector<T> edges(nb + 1);
    unsigned long long h_ctr[16];   // ctr[8] | rbs_d | rbs_y | cnt_d[3] (total, below, inside) | ...
    uint64_t h_rbs = 0;
    unsigned char info[32];
    double sums[5];
    T h_vhat = (T)0;
    std::vector<SelState<K>> h_st(nb);
    std::vector<uint64_t> h_succ(nb);
    std::vector<uint64_t> cnt_t(mr ? 3 * (size_t)nb : 0);   // partitioned plans: (total, below, inside) of the bins' brackets (cnt holds the rewritten ones)
    std::vector<K> khi(nb);
    K h_kd[2] = {0, 0};   // bracket of the median of dh (the next step's prediction keeps the widths of the last sampled brackets)
    const int n_pk = mr ? 14 : 13;
    {
        FzPack pk;
        void* dsts[14] = {cnt.data(), klo.data(), h_ctr, &h_rbs, info, sums, &h_vhat, edges.data(), h_st.data(), h_succ.data(), khi.data(), &h_kd[0], &h_kd[1],
                          cnt_t.data()};
        const void* srcs[14] = {cnt_y, klo_y, ctr, rbs_y, scratch + OFF_INFO, d_sums, d_vhat, d_edges, scratch + OFF_STATE, scratch + off_succ(nb), khi_y, klo_d,
                                khi_d, cnt_true};
        const size_t sizes[14] = {8 * 3 * (size_t)nb, sizeof(K) * nb, 128, 8, 32, 40, sizeof(T), sizeof(T) * (nb + 1), sizeof(SelState<K>) * nb, 8 * (size_t)nb,
                                  sizeof(K) * nb, sizeof(K), sizeof(K), 8 * 3 * (size_t)nb};
        uint32_t off = 0;
        pk.n = n_pk;
        for (int k = 0; k < n_pk; ++k) {
            pk.src[k] = static_cast<const unsigned char*>(srcs[k]);
            pk.bytes[k] = (uint32_t)sizes[k];
            pk.off[k] = off;
            off += (uint32_t)((sizes[k] + 15) & ~(size_t)15);
        }
        if (off > P->fz_pack_bytes) return xd_fail(ctx, XDEMHIP_EINVAL, "one-pass step: result block too small");
        P->fz_host.resize(off);
        // (round 5: the block is written straight into the pinned staging buffer where there is room -- no copy behind the kernel)
        unsigned char* pin = xd_pin_claim(ctx, P->fz_host.data(), off);
        pk.dst = pin ? pin : P->fz_pack;
        hipLaunchKernelGGL(nk_fz_pack_kernel, dim3(1), dim3(256), 0, ctx->stream, pk);
        XD_HIP_CHECK(ctx, hipGetLastError());
        if (!pin) { const int rc_ = xd_d2h(ctx, P->fz_host.data(), P->fz_pack, off); if (rc_) return rc_; }
        { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
        for (int k = 0; k < n_pk; ++k) memcpy(dsts[k], P->fz_host.data() + pk.off[k], sizes[k]);
    }
    std::vector<SelResult<K>> hs(nb);
    for (int k = 0; k < nb; ++k) { hs[k].st = h_st[k]; hs[k].succ = h_succ[k]; }
    if ((h_ctr[2] != 0 || h_ctr[3] != 0) && getenv("XDEMHIP_DEBUG"))
        fprintf(stderr, "[xdemhip] one-pass step falls through: overflow flag %llu, miss flag %llu, dh candidates %llu, bin candidates %llu\n", h_ctr[2], h_ctr[3],
                h_ctr[1], h_ctr[5]);
    if ((h_ctr[2] != 0 || h_ctr[3] != 0) && (predict || predict_d)) {
        // a PREDICTED bracket missed (or overflowed): this step once more with sampled brackets, and the next one sampled too; the
        // prediction has to earn its way back through a measured error
        ++P->n_predict_miss;
        P->pr_cooldown = 1;
        P->pr_err = P->pr_err_d = 1e30;
        return nk_step_onepass<T>(P, g, q0, n, nb, done, vshift, n_valid, y_mean, y_std, edges_out, counts, medians, false);
    }
    if (h_ctr[2] != 0 || h_ctr[3] != 0) {   // overflow / a bracket missed / no extreme-aspect survivor: the plain route takes the step
        if (h_ctr[6] != 0) P->ext_ok = false;   // (no listed extreme-aspect pixel kept a finite dh: this plan reads the aspect from now on -- identical on every rank, the flag derives from reduced counts)
        if (narrow > 0) {   // (narrowed brackets may be what missed: not that narrow again)
            P->fz_unit_min = fmin(1.0, 1.5 * sel_narrow_unit((uint32_t)narrow));
            P->fz_narrow = 0;
        }
        P->pr_have = false;
        return XDEMHIP_OK;
    }
    if (!predict) {
        // How centred were the brackets?  |wanted rank - centre| in half widths, scaled to the full rule.  Lines of a sample that
        // are not fully correlated (the rule's worst case) leave the ranks within a small fraction of it: the next step then takes
        // brackets half or a quarter as wide -- fewer candidates staged, resolved and selected from (measured on C3: 1.88 -> 1.73
        // -> 1.62 ms per step); a miss costs that step the plain route and caps the narrowing (exact either way).
        auto off_of = [&](uint64_t tot, uint64_t lt, uint64_t in) {
            return fabs(((double)(tot - 1) * 0.5 - (double)lt) - 0.5 * (double)in) / (0.5 * (double)in);
        };
        // (offsets are measured in half widths of the bracket that was used; all statistics are kept in units of the FULL rule)
        const double unit = sel_narrow_unit((uint32_t)narrow);
        double worst = 0.0;
        auto take = [&](uint64_t tot, uint64_t lt, uint64_t in) {
            if (tot < 4096 || in < 64 || in >= tot) return;
            const double o = off_of(tot, lt, in) * unit;
            worst = o > worst ? o : worst;
            P->fz_off2 += o * o;
            P->fz_offn += 1;
        };
        if (!predict_d) take(h_ctr[10], h_ctr[11], h_ctr[12]);   // (a predicted bracket says nothing about the centring of sample brackets)
        for (int b = 0; b < nb; ++b) {
            if (mr) take(cnt_t[b], cnt_t[nb + b], cnt_t[2 * nb + b]);
            else take(cnt[b], cnt[nb + b], cnt[2 * nb + b]);
        }
        P->fz_worst = worst > P->fz_worst ? worst : P->fz_worst;
        // rms offset = one standard deviation of the sample ranks in units of the full half width (independent sample elements:
        // 1 / 17, the rule being 6 sigma of 8-element lines that are fully correlated; 0.062-0.068 measured on C3): the next brackets
        // keep >= 4.8 sigma (a miss in 1e-4 of the steps with 73 brackets each -- it costs that step the plain route, nothing else) and
        // >= 1.5 x the worst offset ever seen, in sixteenths of the rule.  (Rounds 4-5 halved or quartered only, at 7 sigma and 2.2 x:
        // C3 sat at one half with 7.7 sigma of room; five sixteenths stage 38 % fewer candidates.)
        int next = 0;
        if (P->fz_offn >= 24) {
            const double sigma = sqrt(P->fz_off2 / (double)P->fz_offn);
            static const int qs[] = {4, 5, 6, 7, 8, 10, 12};
            for (int q : qs) {
                const double u = (double)q / 16.0;
                if (4.8 * sigma <= u && 1.5 * P->fz_worst <= u && u >= P->fz_unit_min) { next = q == 8 ? 1 : (q == 4 ? 2 : (int)SEL_NARROW_16THS + q); break; }
            }
        }
        if (ctx->nk_narrow >= 0) next = ctx->nk_narrow;   // option "nk_narrow": -1 = this rule, 0 / 1 / 2 fixed
        if (getenv("XDEMHIP_DEBUG"))
            fprintf(stderr, "[xdemhip] one-pass step: brackets x %.4f, offsets rms %.3f worst %.3f of the full half width (%lld brackets) -> next x %.4f\n", unit,
                    P->fz_offn ? sqrt(P->fz_off2 / (double)P->fz_offn) : 0.0, P->fz_worst, (long long)P->fz_offn, sel_narrow_unit((uint32_t)next));
        P->fz_narrow = next;
    }
    uint64_t total;
    double vs;
    memcpy(&total, info + 8, 8);
    memcpy(&vs, info + 24, 8);
    *n_valid = (int64_t)total;
    if (total == 0) return xd_fail(ctx, XDEMHIP_EINVAL, "The subsample contains no more valid values.");
    *vshift = vs;
    const int rbs = (int)(uint32_t)h_rbs;
    for (int b = 0; b < nb; ++b) {
        const uint64_t tot = cnt[b], lt = cnt[nb + b];
        if (tot == 0) { hs[b].st.count = 0; counts[b] = 0; medians[b] = NAN; continue; }
        hs[b].st.count = tot;
        hs[b].st.n_le += lt;
        hs[b].st.prefix = (K)((K)(hs[b].st.prefix >> rbs) + klo[b]);  // back from the rebased candidate keys
        if (hs[b].succ != ~(uint64_t)0) hs[b].succ = (uint64_t)(K)((K)((K)hs[b].succ >> rbs) + klo[b]);
        counts[b] = (int64_t)tot;
        medians[b] = median_from<T>(hs[b]);
    }
    for (int k = 0; k <= nb; ++k) edges_out[k] = (double)edges[k];
    // nanmean / nanstd of y from the sums of y^ and the first-order correction in (v^ - vshift)
    const double dlt = (double)h_vhat - vs, cntd = (double)total;
    const double s1 = sums[0] + dlt * sums[2];
    const double s2 = sums[1] + 2.0 * dlt * sums[3] + dlt * dlt * sums[4];
    const double mean = s1 / cntd, var = s2 / cntd - mean * mean;
    *y_mean = mean;
    *y_std = var > 0 ? sqrt(var) : 0.0;
    {
        // Round 6: what the NEXT step's prediction rests on, and how well THIS step was (or would have been) predicted by the model
        // "a bin median moves by -(dE sin(aspect) + dN cos(aspect))" (dE = change of the column offset of the taps, dN = minus the
        // change of their row offset, pixels) and "the median of dh stays".  (Were the signs wrong for some layout, the measured
        // misses would be large and no step would ever be predicted.)
        std::vector<double> mid(nb);
        for (int b = 0; b < nb; ++b) mid[b] = 0.5 * ((double)edges[b] + (double)edges[b + 1]);
        if (P->pr_have && P->pr_nb == nb) {
            double worst = 0.0;
            for (int b = 0; b < nb; ++b) {
                if (!P->pr_ok[b] || counts[b] == 0 || !(P->pr_w[b] > 0)) continue;
                const double e = fabs(medians[b] - (P->pr_med[b] - (dE * sin(P->pr_mid[b]) + dN * cos(P->pr_mid[b])))) / P->pr_w[b];
                worst = e > worst ? e : worst;
            }
            P->pr_err = worst;
            P->pr_err_d = P->pr_wd > 0 ? fabs(vs - P->pr_v) / P->pr_wd : 1e30;
            P->pr_dpx = fmax(dpx, 1e-5);   // (a repeated step measures the jitter floor: the ratio rule then stays conservative)
            if (getenv("XDEMHIP_DEBUG"))
                fprintf(stderr, "[xdemhip] one-pass step (%s, half widths x %.2f): shift change %.2e px, centres off by %.3f (bins) / %.3f (dh) sampled half widths\n",
                        predict ? "PREDICTED brackets" : (predict_d ? "sampled brackets, PREDICTED bracket of the median of dh" : "sampled brackets"), pr_h, dpx, P->pr_err, P->pr_err_d);
        } else {
            P->pr_err = P->pr_err_d = 1e30;
            P->pr_dpx = 1e30;
        }
        if (!predict || (int)P->pr_w.size() != nb) {   // the widths of SAMPLED brackets only (predicted ones are fractions of them)
            // (kept in units of HALF the rule, the brackets the prediction's thresholds were tuned on: narrower sample brackets -- the
            //  sixteenths above -- must not make the prediction shyer or its brackets thinner)
            const double norm = narrow >= (int)SEL_NARROW_16THS ? 0.5 / sel_narrow_unit((uint32_t)narrow) : 1.0;
            P->pr_w.assign(nb, 0.0);
            for (int b = 0; b < nb; ++b)
                if (counts[b] > 0 && khi[b] >= klo[b]) P->pr_w[b] = norm * 0.5 * ((double)val_of(khi[b]) - (double)val_of(klo[b]));
            if (!predict_d) P->pr_wd = h_kd[1] >= h_kd[0] ? norm * 0.5 * ((double)val_of(h_kd[1]) - (double)val_of(h_kd[0])) : 0.0;
        }
        P->pr_med.assign(medians, medians + nb);
        P->pr_mid = mid;
        P->pr_ok.assign(nb, 0);
        for (int b = 0; b < nb; ++b) P->pr_ok[b] = (counts[b] > 0 && P->pr_w[b] > 0 && std::isfinite(P->pr_w[b]) && std::isfinite(medians[b])) ? 1 : 0;
        P->pr_v = vs;
        P->pr_st = sums[2] > 0 ? cntd / sums[2] : 0.0;
        P->pr_sx = g.dc;
        P->pr_sy = g.dr;
        P->pr_nb = nb;
        P->pr_have = true;
        if (predict) ++P->n_predicted;
        if (predict_d) ++P->n_predicted_d;
    }
    *done = true;
    return XDEMHIP_OK;
}

template <typename T>
int nk_step_typed(xdemhip_nk_plan* P, double shift_x, double shift_y, double res_x, double res_y, int nb, double* vshift,
                  int64_t* n_valid, double* y_mean, double* y_std, double* edges_out, int64_t* counts, double* medians,
                  double* fit_sums /* non-null: the un-binned least-squares sums instead of the binning */) {
    typedef typename KeyT<T>::type K;
    xdemhip_ctx* ctx = P->ctx;
    const int64_t q0 = (P->row0 - P->roff) * P->W;
    const int64_t n = (P->row1 - P->row0) * P->W;
    unsigned char* base = static_cast<unsigned char*>(P->scratch);
    double* d_sums = reinterpret_cast<double*>(base + OFF_SUMS);
    const T* dh = static_cast<const T*>(P->dh) + q0;
    const T* st = static_cast<const T*>(P->slope_tan) + q0;
    const T* asp = static_cast<const T*>(P->aspect) + q0;
    T* y = static_cast<T*>(P->y) + q0;
    uint16_t* bins = P->bins + q0;
    T* d_edges = reinterpret_cast<T*>(base);
    T* d_vshift = reinterpret_cast<T*>(base + OFF_INFO);
    if (!P->custom_edges.empty()) nb = (int)P->custom_edges.size() - 1;

    // tba is sampled at (row - shift_y / res_y, col + shift_x / res_x); the taps of this rank's rows must lie in its buffers
    const double dr = -shift_y / res_y, dc = shift_x / res_x;
    const NkGeom g = geom_of(P, dr, dc);
    {
        // Direction-agnostic on purpose: every rank of a partitioned fit must take the same decision (a rank that raised
        // while its neighbours entered the next all-reduce would dead-lock the group), and all interior block borders
        // carry the same halo depth.
        // rules 0 / 1 read the four bilinear taps (rows floor(r + dr), + 1); rules 2 / 3 look one row around the NEAREST pixel,
        // i.e. round(r + dr) +- 1: one more row once the fractional part of |dr| reaches one half
        const int64_t need = (int64_t)floor(fabs(dr)) + (P->nan_rule >= 2 ? 2 : 1);
        const bool top_ok = P->row0 == 0 || P->row0 - P->roff >= need;
        const bool bottom_ok = P->row1 == P->H || P->roff + P->nbuf - P->row1 >= need;
        if (!top_ok || !bottom_ok)
            return xd_fail(ctx, XDEMHIP_EINVAL, "halo too small: the vertical shift moves the bilinear taps outside this rank's row block + halo");
    }
    if (!fit_sums) {   // round 4: one data pass (14 B/pixel) where the plan and the step qualify; anything it cannot prove falls through
        bool done = false;
        const int rc1 = nk_step_onepass<T>(P, g, q0, n, nb, &done, vshift, n_valid, y_mean, y_std, edges_out, counts, medians);
        if (rc1) return rc1;
        if (done) { ++P->n_onepass; return XDEMHIP_OK; }
    }
    int rc = nk_stage_a<T>(P, g, q0, n, nb);
    if (rc) return rc;
    // stages 3 + 4: y = (dh - vshift) / slope_tan, its bins, the per-bin statistic (vshift and the edges are read from the device)
    std::vector<SelResult<K>> hs;
    bool bins_done = false, bins_passed = false, tail_queued = false, committed = false;
    unsigned char info[32];
    std::vector<T> edges_early(nb + 1);
    double sums_early[2] = {0.0, 0.0};
    double* d_bsum = reinterpret_cast<double*>(base + off_hist(nb));
    unsigned long long* d_bcnt = reinterpret_cast<unsigned long long*>(d_bsum + nb);
    double* d_fit = reinterpret_cast<double*>(base + off_hist(1));
    NkYSource<T> src{dh, st, asp, d_vshift, d_edges, d_sums, nullptr, 0.0, (T)0,
                     P->custom_edges.empty() ? NK_AUTO_EDGES : P->custom_decimal};
    BinCacheRec* d_rec = reinterpret_cast<BinCacheRec*>(base + OFF_INFO + 64);
    if (!fit_sums && P->bcache && nb <= NK_BINCACHE_MAX_BINS) {   // (one byte per cached bin id)
        src.bcache = P->bcache + q0;
        src.rec = d_rec;
        hipLaunchKernelGGL((nk_bin_cache_check_kernel<T>), dim3(1), dim3(64), 0, ctx->stream, d_edges, nb, d_rec, (int)P->bcache_force);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    XD_HIP_CHECK(ctx, hipMemsetAsync(d_sums, 0, 16, ctx->stream));
    if (fit_sums) {
        XD_HIP_CHECK(ctx, hipMemsetAsync(d_fit, 0, 80, ctx->stream));
        if (n > 0) {
            hipLaunchKernelGGL((nk_fit_sums_kernel<T>), dim3(grid_for(ctx, n, 256, 16)), dim3(256), 0, ctx->stream, dh, st, asp, n, d_vshift,
                               d_fit);
            XD_HIP_CHECK(ctx, hipGetLastError());
        }
        rc = xd_allreduce_device(ctx, d_fit, 10, XDEMHIP_RED_SUM_F64);
        if (rc) return rc;
    } else if (P->bin_stat == XDEMHIP_BINSTAT_MEAN) {
        // bin_statistic = np.nanmean: one pass, per-bin float64 sums and counts (and the two global sums)
        if (nb > 3072) return xd_fail(ctx, XDEMHIP_EINVAL, "n_bins too large for the mean statistic");
        rc = run_bin_sums<T, NkYSource<T>>(ctx, src, n, nb, d_bsum, d_bcnt);
        if (rc) return rc;
        if (src.bcache) {  // the pass visited every own pixel: the cache now holds the bins of these edges
            hipLaunchKernelGGL((nk_bin_cache_commit_kernel<T>), dim3(1), dim3(64), 0, ctx->stream, d_edges, nb, d_rec);
            P->bcache_force = false;
        }
    } else {
        // per-bin exact medians, bracketed route: y and the bin ids are computed on the fly by the sample / counting
        // passes (NkYSource), the counting pass accumulates the sums
        // On a single GPU everything this step still has to hand back -- the cache commit, vshift / counts / flags, the edges
        // and the two sums -- is queued behind the route BEFORE its one synchronisation (three host round trips of ~50 us
        // otherwise).  With an all-reduce hook the sums need their reduction first: the separate copies below remain.
        const std::function<int()> tail = [&]() -> int {
            if (ctx->allreduce) return XDEMHIP_OK;
            if (src.bcache) {
                hipLaunchKernelGGL((nk_bin_cache_commit_kernel<T>), dim3(1), dim3(64), 0, ctx->stream, d_edges, nb, d_rec);
                XD_HIP_CHECK(ctx, hipGetLastError());
                P->bcache_force = false;
                committed = true;
            }
            { const int rc_ = xd_d2h(ctx, info, base + OFF_INFO, 32); if (rc_) return rc_; }
            { const int rc_ = xd_d2h(ctx, edges_early.data(), d_edges, sizeof(T) * (nb + 1)); if (rc_) return rc_; }
            { const int rc_ = xd_d2h(ctx, sums_early, d_sums, 16); if (rc_) return rc_; }
            tail_queued = true;
            return XDEMHIP_OK;
        };
        rc = run_select_bracketed<T, NkYSource<T>>(ctx, src, n, nb, base, hs, &P->ws, &bins_done, &bins_passed, &tail);
        if (rc) return rc;
        if (src.bcache && !committed) {
            // the counting pass ran over every own pixel (whether or not its brackets held): the cache is filled; a route
            // that never launched it leaves the record untouched and the cache marked stale
            if (bins_passed) {
                hipLaunchKernelGGL((nk_bin_cache_commit_kernel<T>), dim3(1), dim3(64), 0, ctx->stream, d_edges, nb, d_rec);
                P->bcache_force = false;
            } else {
                P->bcache_force = true;
            }
        }
    }
    if (!tail_queued) {
        { const int rc_ = xd_d2h(ctx, info, base + OFF_INFO, 32); if (rc_) return rc_; }
        { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
    }
    uint64_t total, flags;
    double vs;
    memcpy(&total, info + 8, 8);
    memcpy(&flags, info + 16, 8);
    memcpy(&vs, info + 24, 8);
    (void)flags;
    ++P->n_plain;
    *n_valid = (int64_t)total;
    if (total == 0) return xd_fail(ctx, XDEMHIP_EINVAL, "The subsample contains no more valid values.");
    *vshift = vs;
    if (!fit_sums && P->bin_stat == XDEMHIP_BINSTAT_MEDIAN && !bins_done) {
        // small grids, plain mode, a missed bracket: y and bin-id arrays + plain digit passes
        XD_HIP_CHECK(ctx, hipMemsetAsync(d_sums, 0, 16, ctx->stream));
        if (n > 0) {
            hipLaunchKernelGGL((nk_y_kernel<T>), dim3(grid_for(ctx, n, 256, 16)), dim3(256), sizeof(T) * (nb + 1), ctx->stream, dh, st, asp, n,
                               (T)vs, d_edges, nb, y, bins, d_sums, P->custom_edges.empty() ? NK_AUTO_EDGES : P->custom_decimal);
            XD_HIP_CHECK(ctx, hipGetLastError());
        }
        rc = run_select_core<T>(ctx, y, bins, n, nb, base, hs, SEL_MEDIAN, nullptr);
        if (rc) return rc;
    }
    const double cnt = (double)total;
    if (tail_queued && bins_done) {
        // everything arrived with the route's own synchronisation
        const double mean = sums_early[0] / cnt;
        const double var = sums_early[1] / cnt - mean * mean;
        *y_mean = mean;
        *y_std = var > 0 ? sqrt(var) : 0.0;
        for (int k = 0; k < nb; ++k) {
            counts[k] = (int64_t)hs[k].st.count;
            medians[k] = median_from<T>(hs[k]);
        }
        for (int k = 0; k <= nb; ++k) edges_out[k] = (double)edges_early[k];
        return XDEMHIP_OK;
    }
    std::vector<T> edges(nb + 1);
    { const int rc_ = xd_d2h(ctx, edges.data(), d_edges, sizeof(T) * (nb + 1)); if (rc_) return rc_; }
    if (fit_sums) {
        { const int rc_ = xd_d2h(ctx, fit_sums, d_fit, 80); if (rc_) return rc_; }
        { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
        const double mean = fit_sums[6] / cnt;
        const double var = fit_sums[9] / cnt - mean * mean;
        *y_mean = mean;
        *y_std = var > 0 ? sqrt(var) : 0.0;
        return XDEMHIP_OK;
    }
    rc = xd_allreduce_device(ctx, d_sums, 2, XDEMHIP_RED_SUM_F64);
    if (rc) return rc;
    double sums[2];
    { const int rc_ = xd_d2h(ctx, sums, d_sums, 16); if (rc_) return rc_; }
    std::vector<double> bs(nb);
    std::vector<unsigned long long> bc(nb);
    if (P->bin_stat == XDEMHIP_BINSTAT_MEAN) {
        { const int rc_ = xd_d2h(ctx, bs.data(), d_bsum, 8 * (size_t)nb); if (rc_) return rc_; }
        { const int rc_ = xd_d2h(ctx, bc.data(), d_bcnt, 8 * (size_t)nb); if (rc_) return rc_; }
    }
    { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
    const double mean = sums[0] / cnt;
    const double var = sums[1] / cnt - mean * mean;
    *y_mean = mean;
    *y_std = var > 0 ? sqrt(var) : 0.0;
    for (int k = 0; k < nb; ++k) {
        if (P->bin_stat == XDEMHIP_BINSTAT_MEAN) {
            counts[k] = (int64_t)bc[k];
            medians[k] = bc[k] ? (double)(T)(bs[k] / (double)bc[k]) : NAN;  // np.nanmean returns the sample dtype
        } else {
            counts[k] = (int64_t)hs[k].st.count;
            medians[k] = median_from<T>(hs[k]);
        }
    }
    for (int k = 0; k <= nb; ++k) edges_out[k] = (double)edges[k];
    return XDEMHIP_OK;
}

// NuthKaab(bin_statistic=<any callable>) (xdem/coreg/affine.py:2404 -> nd_binning -> scipy.stats.binned_statistic with the callable):
// a Python callable cannot run here, so the step hands back what it would be called on -- y = (dh - vshift) / slope_tan and the aspect-bin
// id of every pixel of the plan's rows, in raster order (NaN / 0xFFFF where the pixel has no dh) -- next to vshift, the valid count, the
// moments of y and the bin edges.  Plain route (stored dh); whole-raster plans only: a callable needs all of a bin's values in one place.
template <typename T>
int nk_step_values(xdemhip_nk_plan* P, double shift_x, double shift_y, double res_x, double res_y, int nb, double* vshift, int64_t* n_valid,
                   double* y_mean, double* y_std, double* edges_out, void* y_out, uint16_t* bins_out, int memspace) {
    xdemhip_ctx* ctx = P->ctx;
    if (ctx->allreduce) return xd_fail(ctx, XDEMHIP_EINVAL, "per-pixel values of a step: whole-raster plans only (a partitioned plan holds a part of every bin)");
    if (P->row0 != 0 || P->row1 != P->H || P->roff != 0) return xd_fail(ctx, XDEMHIP_EINVAL, "per-pixel values of a step: the plan must cover the whole raster");
    const int64_t n = P->H * P->W;
    unsigned char* base = static_cast<unsigned char*>(P->scratch);
    double* d_sums = reinterpret_cast<double*>(base + OFF_SUMS);
    T* d_edges = reinterpret_cast<T*>(base);
    if (!P->custom_edges.empty()) nb = (int)P->custom_edges.size() - 1;
    const NkGeom g = geom_of(P, -shift_y / res_y, shift_x / res_x);
    int rc = nk_stage_a<T>(P, g, 0, n, nb);
    if (rc) return rc;
    unsigned char info[32];
    { const int rc_ = xd_d2h(ctx, info, base + OFF_INFO, 32); if (rc_) return rc_; }
    { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
    uint64_t total;
    double vs;
    memcpy(&total, info + 8, 8);
    memcpy(&vs, info + 24, 8);
    ++P->n_plain;
    *n_valid = (int64_t)total;
    if (total == 0) return xd_fail(ctx, XDEMHIP_EINVAL, "The subsample contains no more valid values.");
    *vshift = vs;
    XD_HIP_CHECK(ctx, hipMemsetAsync(d_sums, 0, 16, ctx->stream));
    hipLaunchKernelGGL((nk_y_kernel<T>), dim3(grid_for(ctx, n, 256, 16)), dim3(256), sizeof(T) * (nb + 1), ctx->stream, static_cast<const T*>(P->dh),
                       static_cast<const T*>(P->slope_tan), static_cast<const T*>(P->aspect), n, (T)vs, d_edges, nb, static_cast<T*>(P->y), P->bins,
                       d_sums, P->custom_edges.empty() ? NK_AUTO_EDGES : P->custom_decimal);
    XD_HIP_CHECK(ctx, hipGetLastError());
    P->bcache_force = true;   // (nothing here maintains the bin cache of the other routes)
    std::vector<T> edges(nb + 1);
    double sums[2];
    { const int rc_ = xd_d2h(ctx, edges.data(), d_edges, sizeof(T) * (nb + 1)); if (rc_) return rc_; }
    { const int rc_ = xd_d2h(ctx, sums, d_sums, 16); if (rc_) return rc_; }
    const hipMemcpyKind kind = memspace == XDEMHIP_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    XD_HIP_CHECK(ctx, hipMemcpyAsync(y_out, P->y, (size_t)n * sizeof(T), kind, ctx->stream));
    XD_HIP_CHECK(ctx, hipMemcpyAsync(bins_out, P->bins, (size_t)n * 2, kind, ctx->stream));
    { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
    const double cnt = (double)total, mean = sums[0] / cnt, var = sums[1] / cnt - mean * mean;
    *y_mean = mean;
    *y_std = var > 0 ? sqrt(var) : 0.0;
    for (int k = 0; k <= nb; ++k) edges_out[k] = (double)edges[k];
    return XDEMHIP_OK;
}

// Shared by the two creation entry points.  Buffers hold raster rows [roff, roff + nbuf); own rows [row0, row1).
int nk_create_impl(xdemhip_ctx* ctx, const void* ref, const void* tba, const uint8_t* inlier, int dtype, int64_t H, int64_t W,
                   int64_t roff, int64_t nbuf, int64_t row0, int64_t row1, int memspace, bool global_count,
                   xdemhip_nk_plan** out_plan, int64_t* n_valid) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (!ref || !tba || !out_plan) return xd_fail(ctx, XDEMHIP_EINVAL, "null argument");
    if (H < 2 || W < 2) return xd_fail(ctx, XDEMHIP_EINVAL, "Shape of array too small to calculate a numerical gradient, at least 2 elements are required.");
    if (dtype != XDEMHIP_F32 && dtype != XDEMHIP_F64) return xd_fail(ctx, XDEMHIP_EINVAL, "dtype must be float32 or float64");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t es = dtype == XDEMHIP_F32 ? 4 : 8;
    const size_t n = (size_t)nbuf * (size_t)W;
    xdemhip_nk_plan* P = new xdemhip_nk_plan();
    P->ctx = ctx; P->dtype = dtype; P->H = H; P->W = W; P->roff = roff; P->nbuf = nbuf; P->row0 = row0; P->row1 = row1;
    P->nan_rule = ctx->nk_nan_rule;
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
        hipMalloc(reinterpret_cast<void**>(&P->bins), n * 2) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&P->bcache), n * sizeof(nk_bin_t)) != hipSuccess || hipMalloc(&P->scratch, P->scratch_bytes) != hipSuccess)
        return fail(XDEMHIP_ENOMEM, "hipMalloc failed");
    if ((int64_t)n >= SEL_BRACKET_MIN_N && sel_ws_create(ctx, (int64_t)n, es, MAX_BINS_PER_SWEEP, P->ws) != XDEMHIP_OK)
        return fail(XDEMHIP_ENOMEM, "hipMalloc failed");
    // EXT buffers of the one-pass step (large plans): masked copy of the reference DEM + the lists of extreme-aspect pixels;
    // without the memory for them the plan simply keeps the plain route, which reads mask and aspect
    if ((int64_t)n >= SEL_BRACKET_MIN_N && ctx->nk_fused != 0) {
        if (hipMalloc(&P->ref_m, n * es) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&P->ext_idx), (size_t)2 * EXT_CAP * 8) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&P->ext_cnt), 32) != hipSuccess) {
            (void)hipGetLastError();
            if (P->ref_m) (void)hipFree(P->ref_m);
            if (P->ext_idx) (void)hipFree(P->ext_idx);
            if (P->ext_cnt) (void)hipFree(P->ext_cnt);
            P->ref_m = nullptr; P->ext_idx = nullptr; P->ext_cnt = nullptr;
        }
    }
    // one-pass step (large single-GPU plans with the EXT buffers): its device block and candidate buffers; without the memory the
    // plan keeps the plain route
    if (P->ref_m && P->ws.d_small && ctx->nk_fused != 0) {
        // (... + places taken in the per-bin candidate segments: one line per bin; + header and histogram of the dh selection --
        //  all zeroed at the start of a step; the keys of its chosen bucket sit behind, outside the zeroed part)
        P->fz_bytes = (size_t)(24 + (11 + BINSEG_CTR_STRIDE) * P->ws.nb_max + DSEL_HDR_WORDS + DSEL_BUCKETS / 2) * 8;
        P->cd_cap = (int64_t)n / 8 + 4096;
        P->fz_pack_bytes = (size_t)P->ws.nb_max * (8 * 3 + 8 + 16 + 64 + 8 + 8 * 3 + 16) + 1024;
        if (hipMalloc(reinterpret_cast<void**>(&P->fz), P->fz_bytes + (size_t)DSEL_CAP * 8) != hipSuccess || hipMalloc(&P->cd_vals, (size_t)P->cd_cap * es) != hipSuccess ||
            hipMalloc(&P->c_st, (size_t)P->ws.c_cap * es) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&P->fz_pack), P->fz_pack_bytes) != hipSuccess) {
            (void)hipGetLastError();
            if (P->fz) (void)hipFree(P->fz);
            if (P->cd_vals) (void)hipFree(P->cd_vals);
            if (P->c_st) (void)hipFree(P->c_st);
            if (P->fz_pack) (void)hipFree(P->fz_pack);
            P->fz = nullptr; P->cd_vals = nullptr; P->c_st = nullptr; P->fz_pack = nullptr;
        }
    }
    // rules 2 / 3 on plans large enough for the streaming kernels (whole rasters and row blocks alike): the per-pixel neighbourhood
    // flags of the tba buffer, once
    if (P->nan_rule >= 2 && (int64_t)n >= SEL_BRACKET_MIN_N) {
        P->bad_wpr = (W + 63) / 64 + 2;
        if (hipMalloc(reinterpret_cast<void**>(&P->badbits), (size_t)P->bad_wpr * (size_t)nbuf * 8) != hipSuccess) {
            (void)hipGetLastError();
            P->badbits = nullptr;   // (the plan keeps the generic kernels)
        } else {
            const dim3 bgrid((unsigned)((P->bad_wpr - 2 + 3) / 4), (unsigned)(nbuf < 4096 ? nbuf : 4096));
            if (dtype == XDEMHIP_F32)
                hipLaunchKernelGGL((nk_badbits_kernel<float>), bgrid, dim3(256), 0, ctx->stream, static_cast<const float*>(P->tba), H, W, roff, nbuf, P->nan_rule, P->bad_wpr, P->badbits);
            else
                hipLaunchKernelGGL((nk_badbits_kernel<double>), bgrid, dim3(256), 0, ctx->stream, static_cast<const double*>(P->tba), H, W, roff, nbuf, P->nan_rule, P->bad_wpr, P->badbits);
            if (hipGetLastError() != hipSuccess) return fail(XDEMHIP_EHIP, "nk_badbits_kernel launch failed");
        }
    }
    const xdemhip_allreduce_fn hook = ctx->allreduce;
    if (!global_count) ctx->allreduce = nullptr;  // whole-raster plan: local pass; xdemhip_nk_set_rows re-partitions with the hook
    int rc = dtype == XDEMHIP_F32 ? nk_aux_typed<float>(P) : nk_aux_typed<double>(P);
    ctx->allreduce = hook;
    if (rc != XDEMHIP_OK) { xdemhip_nk_destroy(P); return rc; }
    if (n_valid) *n_valid = P->n_valid0;
    *out_plan = P;
    return XDEMHIP_OK;
}

}  // namespace

extern "C" {

void xdemhip_nk_destroy(xdemhip_nk_plan* P) {
    if (!P) return;
    (void)hipSetDevice(P->ctx->device);
    uint8_t* inl_created = P->subsampled ? P->inlier_user : P->inlier;
    if (P->own_inputs) { (void)hipFree(P->ref); (void)hipFree(P->tba); if (inl_created) (void)hipFree(inl_created); }
    if (P->sub_inlier) (void)hipFree(P->sub_inlier);
    void* bufs[] = {P->slope_tan, P->aspect, P->dh, P->y, P->valid, P->bins, P->bcache, P->scratch, P->ref_m, P->ext_idx, P->ext_cnt,
                    P->fz, P->cd_vals, P->c_st, P->fz_pack, P->badbits, P->mr_a, P->mr_b, P->mr_small, P->wg_sums};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    xd::sel_ws_free(P->ws);
    delete P;
}

int xdemhip_nk_create(xdemhip_ctx* ctx, const void* ref, const void* tba, const uint8_t* inlier, int dtype, int64_t H, int64_t W,
                      int memspace, xdemhip_nk_plan** out_plan, int64_t* n_valid) {
    XdFetchScope fetch_scope_(ctx);
    return nk_create_impl(ctx, ref, tba, inlier, dtype, H, W, 0, H, 0, H, memspace, false, out_plan, n_valid);
}

int xdemhip_nk_create_block(xdemhip_ctx* ctx, const void* ref_block, const void* tba_block, const uint8_t* inlier_block, int dtype,
                            int64_t H, int64_t W, int64_t row_begin, int64_t row_end, int64_t halo_top, int64_t halo_bottom,
                            int memspace, xdemhip_nk_plan** out_plan, int64_t* n_valid) {
    XdFetchScope fetch_scope_(ctx);
    if (!ctx) return XDEMHIP_EINVAL;
    if (row_begin < 0 || row_end < row_begin || row_end > H || halo_top < 0 || halo_bottom < 0 || halo_top > row_begin ||
        row_end + halo_bottom > H)
        return xd_fail(ctx, XDEMHIP_EINVAL, "bad row block");
    // np.gradient reads one neighbour row on each side of an own row: a block that does not start / end at the raster's
    // border needs at least one halo row there
    if ((row_begin > 0 && halo_top < 1) || (row_end < H && halo_bottom < 1))
        return xd_fail(ctx, XDEMHIP_EINVAL, "a row block inside the raster needs >= 1 halo row towards each neighbour");
    return nk_create_impl(ctx, ref_block, tba_block, inlier_block, dtype, H, W, row_begin - halo_top, (row_end - row_begin) + halo_top + halo_bottom,
                          row_begin, row_end, memspace, true, out_plan, n_valid);
}

int xdemhip_nk_set_rows(xdemhip_nk_plan* P, int64_t row_begin, int64_t row_end, int64_t* n_valid) {
    XdFetchScope fetch_scope_(P ? P->ctx : nullptr);
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (row_begin < P->roff || row_end < row_begin || row_end > P->roff + P->nbuf) return xd_fail(ctx, XDEMHIP_EINVAL, "bad row range");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    P->row0 = row_begin;
    P->row1 = row_end;
    P->bcache_force = true;  // other own rows: their bins were never cached
    P->pr_have = false;      // ... and their medians are not the previous step's
    // valid mask / aux rasters outside the range are never read by this rank; recount the global number of valid pixels
    int rc = P->dtype == XDEMHIP_F32 ? nk_aux_typed<float>(P) : nk_aux_typed<double>(P);
    if (rc) return rc;
    if (n_valid) *n_valid = P->n_valid0;
    return XDEMHIP_OK;
}

int xdemhip_nk_subsample(xdemhip_nk_plan* P, const int64_t* ranks, int64_t k, int memspace, int64_t* n_valid) {
    XdFetchScope fetch_scope_(P ? P->ctx : nullptr);
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (ctx->allreduce || P->row0 != 0 || P->row1 != P->H || P->roff != 0 || P->nbuf != P->H)
        return xd_fail(ctx, XDEMHIP_EINVAL, "xdemhip_nk_subsample: whole-raster plans of one process only (partitioned plans: pass the drawn mask as inlier mask)");
    if (!ranks || k < 1 || k > P->n_valid0) return xd_fail(ctx, XDEMHIP_EINVAL, "xdemhip_nk_subsample: 1 <= k <= the plan's valid pixels");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int64_t n = P->H * P->W, n_tiles = (n + SUBS_TILE - 1) / SUBS_TILE, n_ranks = P->n_valid0;
    if (n_tiles > 0x7FFFFFFF) return xd_fail(ctx, XDEMHIP_EINVAL, "xdemhip_nk_subsample: raster too large");
    uint8_t* mark = nullptr;
    unsigned long long* tiles = nullptr;   // [n_tiles] counts -> offsets, [n_tiles] total, [n_tiles + 1] ranks out of range
    int64_t* d_ranks = nullptr;
    auto release = [&]() {
        if (mark) (void)hipFree(mark);
        if (tiles) (void)hipFree(tiles);
        if (d_ranks) (void)hipFree(d_ranks);
    };
    if (hipMalloc(reinterpret_cast<void**>(&mark), (size_t)n_ranks) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&tiles), (size_t)(n_tiles + 2) * 8) != hipSuccess ||
        (memspace == XDEMHIP_HOST && hipMalloc(reinterpret_cast<void**>(&d_ranks), (size_t)k * 8) != hipSuccess) ||
        (!P->sub_inlier && hipMalloc(reinterpret_cast<void**>(&P->sub_inlier), (size_t)n) != hipSuccess)) {
        (void)hipGetLastError();
        release();
        return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed (xdemhip_nk_subsample)");
    }
    auto fail = [&](int code, const char* msg) { release(); return xd_fail(ctx, code, msg); };
    const int64_t* rk = ranks;
    if (memspace == XDEMHIP_HOST) {
        if (hipMemcpyAsync(d_ranks, ranks, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail(XDEMHIP_EHIP, "copy of the ranks failed");
        rk = d_ranks;
    }
    if (hipMemsetAsync(mark, 0, (size_t)n_ranks, ctx->stream) != hipSuccess || hipMemsetAsync(tiles + n_tiles, 0, 16, ctx->stream) != hipSuccess)
        return fail(XDEMHIP_EHIP, "hipMemsetAsync failed");
    hipLaunchKernelGGL(nk_subs_mark_kernel, dim3(grid_for(ctx, k, 256, 8)), dim3(256), 0, ctx->stream, rk, k, n_ranks, mark, tiles + n_tiles + 1);
    hipLaunchKernelGGL(nk_subs_count_kernel, dim3((unsigned)n_tiles), dim3(256), 0, ctx->stream, P->valid, n, tiles);
    hipLaunchKernelGGL(nk_subs_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, tiles, n_tiles, tiles + n_tiles);
    hipLaunchKernelGGL(nk_subs_apply_kernel, dim3((unsigned)n_tiles), dim3(256), 0, ctx->stream, P->valid, n, tiles, mark, P->sub_inlier);
    if (hipGetLastError() != hipSuccess) return fail(XDEMHIP_EHIP, "xdemhip_nk_subsample: kernel launch failed");
    unsigned long long chk[2] = {0, 0};
    { const int rc_ = xd_d2h(ctx, chk, tiles + n_tiles, 16); if (rc_) { release(); return rc_; } }
    { const int rc_ = xd_sync(ctx); if (rc_) { release(); return rc_; } }
    release();
    if ((long long)chk[0] != P->n_valid0) return xd_fail(ctx, XDEMHIP_EINVAL, "xdemhip_nk_subsample: the valid mask changed under the call");
    if (chk[1] != 0) return xd_fail(ctx, XDEMHIP_EINVAL, "xdemhip_nk_subsample: a rank is outside [0, n_valid)");
    if (!P->subsampled) { P->inlier_user = P->inlier; P->subsampled = true; }
    P->inlier = P->sub_inlier;
    // (as after xdemhip_nk_set_rows: other valid pixels -- no cached bins, no previous medians, no measured sample offsets)
    P->bcache_force = true;
    P->pr_have = false;
    P->fz_narrow = 0; P->fz_unit_min = 0.25; P->fz_worst = 0.0; P->fz_off2 = 0.0; P->fz_offn = 0;
    const int rc = P->dtype == XDEMHIP_F32 ? nk_aux_typed<float>(P) : nk_aux_typed<double>(P);
    if (rc) return rc;
    if (n_valid) *n_valid = P->n_valid0;
    return XDEMHIP_OK;
}

int xdemhip_nk_predict_counts(xdemhip_nk_plan* P, int64_t* predicted, int64_t* predicted_dh_only, int64_t* missed) {
    if (!P) return XDEMHIP_EINVAL;
    if (predicted) *predicted = P->n_predicted;
    if (predicted_dh_only) *predicted_dh_only = P->n_predicted_d;
    if (missed) *missed = P->n_predict_miss;
    return XDEMHIP_OK;
}

int xdemhip_nk_route_counts(xdemhip_nk_plan* P, int64_t* onepass, int64_t* plain) {
    if (!P) return XDEMHIP_EINVAL;
    if (onepass) *onepass = P->n_onepass;
    if (plain) *plain = P->n_plain;
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
    { const int rc_ = xd_sync(ctx); if (rc_) return rc_; }
    const size_t es = P->dtype == XDEMHIP_F32 ? 4 : 8, n = (size_t)P->nbuf * (size_t)P->W;  // (the plan's buffer rows)
    if (slope_tan) XD_HIP_CHECK(ctx, hipMemcpy(slope_tan, P->slope_tan, n * es, hipMemcpyDeviceToHost));
    if (aspect) XD_HIP_CHECK(ctx, hipMemcpy(aspect, P->aspect, n * es, hipMemcpyDeviceToHost));
    if (valid) XD_HIP_CHECK(ctx, hipMemcpy(valid, P->valid, n, hipMemcpyDeviceToHost));
    return XDEMHIP_OK;
}

int xdemhip_nk_step(xdemhip_nk_plan* P, double shift_x, double shift_y, double res_x, double res_y, int n_bins, double* vshift,
                    int64_t* n_valid, double* y_mean, double* y_std, double* edges, int64_t* counts, double* medians) {
    XdFetchScope fetch_scope_(P ? P->ctx : nullptr);
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (!vshift || !n_valid || !y_mean || !y_std || !edges || !counts || !medians) return xd_fail(ctx, XDEMHIP_EINVAL, "null output");
    if (n_bins < 1 || n_bins > P->max_bins) return xd_fail(ctx, XDEMHIP_EINVAL, "n_bins out of range (1..1024)");
    // explicit edges fix the number of bins: the caller's output arrays are sized by ITS n_bins, so the two must agree
    if (!P->custom_edges.empty() && n_bins != (int)P->custom_edges.size() - 1)
        return xd_fail(ctx, XDEMHIP_EINVAL, "n_bins must equal the number of explicit bin edges - 1 (xdemhip_nk_set_bin_edges)");
    if (!(res_x > 0) || !(res_y > 0)) return xd_fail(ctx, XDEMHIP_EINVAL, "resolution must be > 0");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    int rc = P->dtype == XDEMHIP_F32
                 ? nk_step_typed<float>(P, shift_x, shift_y, res_x, res_y, n_bins, vshift, n_valid, y_mean, y_std, edges, counts, medians, nullptr)
                 : nk_step_typed<double>(P, shift_x, shift_y, res_x, res_y, n_bins, vshift, n_valid, y_mean, y_std, edges, counts, medians, nullptr);
    (void)hipEventRecord(ctx->ev_stop, ctx->stream);
    ctx->timed = (rc == XDEMHIP_OK);
    return rc;
}

int xdemhip_nk_step_fit(xdemhip_nk_plan* P, double shift_x, double shift_y, double res_x, double res_y, double* vshift, int64_t* n_valid,
                        double* y_mean, double* y_std, double* sums) {
    XdFetchScope fetch_scope_(P ? P->ctx : nullptr);
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (!vshift || !n_valid || !y_mean || !y_std || !sums) return xd_fail(ctx, XDEMHIP_EINVAL, "null output");
    if (!(res_x > 0) || !(res_y > 0)) return xd_fail(ctx, XDEMHIP_EINVAL, "resolution must be > 0");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    int rc = P->dtype == XDEMHIP_F32
                 ? nk_step_typed<float>(P, shift_x, shift_y, res_x, res_y, 72, vshift, n_valid, y_mean, y_std, nullptr, nullptr, nullptr, sums)
                 : nk_step_typed<double>(P, shift_x, shift_y, res_x, res_y, 72, vshift, n_valid, y_mean, y_std, nullptr, nullptr, nullptr, sums);
    (void)hipEventRecord(ctx->ev_stop, ctx->stream);
    ctx->timed = (rc == XDEMHIP_OK);
    return rc;
}

int xdemhip_nk_step_values(xdemhip_nk_plan* P, double shift_x, double shift_y, double res_x, double res_y, int n_bins, double* vshift,
                           int64_t* n_valid, double* y_mean, double* y_std, double* edges, void* y_out, uint16_t* bins_out, int memspace) {
    XdFetchScope fetch_scope_(P ? P->ctx : nullptr);
    if (!P) return XDEMHIP_EINVAL;
    xdemhip_ctx* ctx = P->ctx;
    if (!vshift || !n_valid || !y_mean || !y_std || !edges || !y_out || !bins_out) return xd_fail(ctx, XDEMHIP_EINVAL, "null output");
    if (!(res_x > 0) || !(res_y > 0)) return xd_fail(ctx, XDEMHIP_EINVAL, "resolution must be > 0");
    if (memspace != XDEMHIP_HOST && memspace != XDEMHIP_DEVICE) return xd_fail(ctx, XDEMHIP_EINVAL, "memspace must be XDEMHIP_HOST or XDEMHIP_DEVICE");
    if (n_bins < 1 || n_bins > P->max_bins) return xd_fail(ctx, XDEMHIP_EINVAL, "n_bins out of range (1..1024)");
    if (!P->custom_edges.empty() && n_bins != (int)P->custom_edges.size() - 1)
        return xd_fail(ctx, XDEMHIP_EINVAL, "n_bins must equal the number of explicit bin edges - 1 (xdemhip_nk_set_bin_edges)");
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    const int rc = P->dtype == XDEMHIP_F32
                       ? nk_step_values<float>(P, shift_x, shift_y, res_x, res_y, n_bins, vshift, n_valid, y_mean, y_std, edges, y_out, bins_out, memspace)
                       : nk_step_values<double>(P, shift_x, shift_y, res_x, res_y, n_bins, vshift, n_valid, y_mean, y_std, edges, y_out, bins_out, memspace);
    (void)hipEventRecord(ctx->ev_stop, ctx->stream);
    ctx->timed = (rc == XDEMHIP_OK);
    return rc;
}

int xdemhip_nk_set_bin_edges(xdemhip_nk_plan* P, const double* edges, int n_edges, int decimal) {
    if (!P) return XDEMHIP_EINVAL;
    P->bcache_force = true;
    P->pr_have = false;
    if (n_edges == 0) { P->custom_edges.clear(); return XDEMHIP_OK; }
    if (!edges || n_edges < 2 || n_edges - 1 > MAX_BINS_PER_SWEEP) return xd_fail(P->ctx, XDEMHIP_EINVAL, "bin edges: 2 .. 129 increasing values");
    for (int k = 1; k < n_edges; ++k)
        if (!(edges[k] > edges[k - 1])) return xd_fail(P->ctx, XDEMHIP_EINVAL, "bin edges must increase strictly");
    P->custom_edges.assign(edges, edges + n_edges);
    P->custom_decimal = decimal;
    return XDEMHIP_OK;
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
    XdPrefault prefault;
    if (memspace == XDEMHIP_HOST) {
        d_src = d_out = nullptr;
        if (hipMalloc(&d_src, bytes) != hipSuccess || hipMalloc(&d_out, bytes) != hipSuccess) {
            if (d_src) (void)hipFree(d_src);
            return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
        }
        (void)hipMemcpyAsync(d_src, src, bytes, hipMemcpyHostToDevice, ctx->stream);
        xd_prefault_start(prefault, out, bytes, ctx->host_copy_threads);   // (the caller's output pages, while the raster travels and the kernel runs)
    }
    const int64_t n = H * W;
    XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    NkGeom g;
    g.H = H; g.W = W; g.roff = 0; g.dr = shift_row_px; g.dc = shift_col_px; g.rule = ctx->nk_nan_rule;
    if (dtype == XDEMHIP_F32)
        hipLaunchKernelGGL((shift_bilinear_kernel<float>), dim3(grid_for(ctx, n, 256, 16)), dim3(256), 0, ctx->stream,
                           static_cast<const float*>(d_src), g, (float)dz, static_cast<float*>(d_out));
    else
        hipLaunchKernelGGL((shift_bilinear_kernel<double>), dim3(grid_for(ctx, n, 256, 16)), dim3(256), 0, ctx->stream,
                           static_cast<const double*>(d_src), g, dz, static_cast<double*>(d_out));
    (void)hipEventRecord(ctx->ev_stop, ctx->stream);
    ctx->timed = true;
    int rc = XDEMHIP_OK;
    if (hipGetLastError() != hipSuccess) rc = xd_fail(ctx, XDEMHIP_EHIP, "shift kernel launch failed");
    if (memspace == XDEMHIP_HOST) {
        prefault.join();
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
    XdFetchScope fetch_scope_(ctx);
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
