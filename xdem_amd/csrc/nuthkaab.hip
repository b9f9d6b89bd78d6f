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

// Geometry of a plan's buffers.  Every per-pixel array (ref, tba, valid, slope_tan, aspect, dh) covers raster rows
// [roff, roff + nbuf) -- the whole raster, or a rank's row block plus its halo rows (multi-GPU) -- and is indexed by the
// LOCAL linear index q = (row - roff) * W + col.
struct NkGeom {
    int64_t H, W;      // raster shape (global)
    int64_t roff;      // raster row of buffer row 0
    double dr, dc;     // tap position = (row + dr, col + dc)
    int rule;          // NaN rule of the bilinear taps (context option "nk_nan_rule")
};

// ---- bilinear sample of tba at a shifted position ------------------------------------------------------------------
// geoutils' _interp_points (un-vendored, absent here) is restated as: bilinear, float64 weights, result rounded to the DEM
// dtype.  How nodata spreads is NOT pinned by anything in this image, so it is switchable (context option "nk_nan_rule"):
//   0 "4tap"      NaN if any of the four taps is non-finite or outside the raster, zero weights included (what
//                 scipy.ndimage.map_coordinates(order=1) does to NaN: 0 * NaN = NaN)                         [default]
//   1 "weighted"  taps with zero weight are ignored: at integer shifts the last row / column keep their values
//   2 "dilate3x3" NaN if any pixel of the 3 x 3 neighbourhood of the NEAREST pixel is non-finite or outside
//   3 "dilate_cross" the same with the 4-connected cross instead of the square (SciPy's default binary-dilation structure)
struct BiTap {
    int64_t q00;       // local index of the top-left tap; the others are q00 + dc1, q00 + drw, q00 + drw + dc1
    int64_t drw;       // W, or 0 where the lower row is ignored (rule 1, zero row weight)
    int dc1;           // 1, or 0 where the right column is ignored
    int64_t qn;        // nearest pixel (rule 2), -1 if its 3 x 3 neighbourhood leaves the raster
    double fr, fc;
    bool in;
};
// The tap position separates into a row part and a column part (kernels that walk down a column compute the latter once).
struct BiAxis { int64_t k0; int d1; double f; double pos; bool in; };
__device__ __forceinline__ BiAxis bi_axis(int64_t idx, double shift, int64_t extent, int rule) {
    BiAxis a;
    a.pos = t_add((double)idx, shift);
    const double k0f = floor(a.pos);
    a.f = t_sub(a.pos, k0f);
    a.k0 = (int64_t)k0f;
    // a node exactly on the upper edge needs no tap beyond it (any linear interpolator returns the node value there)
    a.d1 = (a.f == 0.0 && (rule == 1 || a.k0 + 1 >= extent)) ? 0 : 1;
    a.in = a.k0 >= 0 && a.k0 + a.d1 < extent;
    return a;
}
__device__ __forceinline__ BiTap bi_combine(const NkGeom& g, const BiAxis& r, const BiAxis& c) {
    BiTap t;
    t.fr = r.f;
    t.fc = c.f;
    t.in = r.in && c.in;
    t.q00 = t.in ? (r.k0 - g.roff) * g.W + c.k0 : 0;
    t.drw = t.in ? (int64_t)r.d1 * g.W : 0;
    t.dc1 = t.in ? c.d1 : 0;
    t.qn = -1;
    if (g.rule >= 2) {
        const int64_t rn = (int64_t)floor(r.pos + 0.5), cn = (int64_t)floor(c.pos + 0.5);
        if (rn >= 1 && cn >= 1 && rn + 1 < g.H && cn + 1 < g.W) t.qn = (rn - g.roff) * g.W + cn;
    }
    return t;
}
__device__ __forceinline__ BiTap bi_locate(const NkGeom& g, int64_t i, int64_t j) {
    return bi_combine(g, bi_axis(i, g.dr, g.H, g.rule), bi_axis(j, g.dc, g.W, g.rule));
}
template <typename T> struct BiVals { T a00, a01, a10, a11; };
template <typename T> __device__ __forceinline__ BiVals<T> bi_load(const T* __restrict__ img, const BiTap& t) {
    const T* q = img + t.q00;
    BiVals<T> v;
    v.a00 = q[0]; v.a01 = q[t.dc1]; v.a10 = q[t.drw]; v.a11 = q[t.drw + t.dc1];
    return v;
}
template <typename T>
__device__ __forceinline__ bool bi_value(const NkGeom& g, const T* __restrict__ img, const BiTap& t, T a00, T a01, T a10, T a11, T& out) {
    bool ok = t.in && t_finite(a00) && t_finite(a01) && t_finite(a10) && t_finite(a11);
    if (g.rule >= 2) {
        ok = ok && t.qn >= 0;
        if (ok)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                    if (g.rule == 2 || dy == 0 || dx == 0) ok = ok && t_finite(img[t.qn + dy * g.W + dx]);  // rule 3: the cross only
    }
    const double v00 = a00, v01 = a01, v10 = a10, v11 = a11;
    const double top = t_add(v00, t_mul(t.fc, t_sub(v01, v00)));
    const double bot = t_add(v10, t_mul(t.fc, t_sub(v11, v10)));
    out = (T)t_add(top, t_mul(t.fr, t_sub(bot, top)));
    return ok;
}
// row / column of a local linear index (W <= 2^31, q < 2^52: one float64 multiply and a correction step)
__device__ __forceinline__ void row_col(int64_t q, int64_t W, double invW, int64_t& li, int64_t& j) {
    li = (int64_t)((double)q * invW);
    j = q - li * W;
    if (j < 0) { --li; j += W; }
    else if (j >= W) { ++li; j -= W; }
}

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

// nk_dh_kernel + the counting / compaction pass of the global median's bracketed selection in one kernel: per pixel the
// same arithmetic as nk_dh_kernel, then the order-preserving key of dh is compared with the bracket [klo, khi] of the
// median (from the sample): counts of all / below / inside in registers, the few inside (~1-2 %) are compacted through a
// small per-WAVE LDS staging buffer (no workgroup barrier anywhere in the row loop: the waves keep streaming independently)
// that a wave empties with one global atomic once more than half of its 512 slots are taken (a row adds at most 64).
constexpr int NKF_STAGE = 512;
template <typename T>
__global__ __launch_bounds__(256) void nk_dh_count_kernel(const T* __restrict__ ref, const T* __restrict__ tba,
                                                          const uint8_t* __restrict__ valid, const T* __restrict__ aspect,
                                                          NkGeom g, T* __restrict__ dh, DhStats* stats, int64_t row0, int64_t row1,
                                                          const typename KeyT<T>::type* __restrict__ klo_p,
                                                          const typename KeyT<T>::type* __restrict__ khi_p, uint64_t* counters /* [3] */,
                                                          T* out_v, unsigned long long* ctr /* [1] candidates, [2] overflow */, int64_t cap) {
    typedef typename KeyT<T>::type K;
    __shared__ T stage_all[4][NKF_STAGE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T* stage = stage_all[wave];
    int held = 0;  // wave-uniform
    const K klo = *klo_p, khi = *khi_p;
    K kmin = ~(K)0, kmax = 0;
    uint32_t n_all = 0, n_below = 0, n_in = 0;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool jin = j < g.W;
    const BiAxis col = bi_axis(j, g.dc, g.W, g.rule);
    auto flush = [&]() {
        unsigned long long b0 = 0;
        if (lane == 0) b0 = atomicAdd(&ctr[1], (unsigned long long)held);
        b0 = __shfl(b0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int k = lane; k < held; k += 64) {
            if ((int64_t)(b0 + k) < cap) out_v[b0 + k] = stage[k];
            else ctr[2] = 1ull;
        }
        held = 0;
    };
    // This workgroup walks a contiguous chunk of rows down its 256 columns.  The taps of output row i + 1 sit one raster row
    // below those of row i, so the lower tap pair of a row is carried over as the upper pair of the next one (two tap loads per
    // pixel instead of four), and every load of row i + 1 is issued before row i is evaluated: one row of latency is always
    // in flight.  (The carry is taken only when the tap rows really are consecutive: pos = i + dr is rounded per row.)
    const int64_t chunk = (row1 - row0 + gridDim.y - 1) / gridDim.y;
    int64_t i = row0 + (int64_t)blockIdx.y * chunk;
    const int64_t iend = (i + chunk < row1) ? i + chunk : row1;
    const int64_t jj = jin ? j : 0;  // lanes beyond the raster follow column 0 (loads stay in bounds) and discard everything
    BiAxis rax = bi_axis(i < iend ? i : row0, g.dr, g.H, g.rule);
    BiTap t = bi_combine(g, rax, col);
    T a00 = (T)0, a01 = (T)0, a10 = (T)0, a11 = (T)0, rv = (T)0, av = (T)0;
    uint8_t vd = 0;
    int64_t p = (i - g.roff) * g.W + jj;
    if (i < iend) {
        const BiVals<T> tv = bi_load<T>(tba, t);
        a00 = tv.a00; a01 = tv.a01; a10 = tv.a10; a11 = tv.a11;
        rv = ref[p]; av = aspect[p]; vd = valid[p];
    }
    for (; i < iend; ++i) {
        // ---- loads of the next row
        BiAxis rn = rax;
        BiTap tn = t;
        T n00 = (T)0, n01 = (T)0, n10 = (T)0, n11 = (T)0, nrv = (T)0, nav = (T)0;
        uint8_t nvd = 0;
        const int64_t pn = p + g.W;
        if (i + 1 < iend) {
            rn = bi_axis(i + 1, g.dr, g.H, g.rule);
            tn = bi_combine(g, rn, col);
            nrv = ref[pn]; nav = aspect[pn]; nvd = valid[pn];
            if (rax.in && rn.in && rax.d1 == 1 && rn.k0 == rax.k0 + 1) {  // (wave-uniform)
                n00 = a10; n01 = a11;
                if (rn.d1) { const T* q = tba + tn.q00 + tn.drw; n10 = q[0]; n11 = q[tn.dc1]; }
                else { n10 = n00; n11 = n01; }
            } else {
                const BiVals<T> tv = bi_load<T>(tba, tn);
                n00 = tv.a00; n01 = tv.a01; n10 = tv.a10; n11 = tv.a11;
            }
        }
        // ---- this row
        bool cand = false;
        T out = (T)NAN;
        {
            T val;
            const bool ok = bi_value<T>(g, tba, t, a00, a01, a10, a11, val) & (vd != 0) & jin;
            out = t_sub(rv, val);
            if (ok && t_finite(out)) {
                const K ka = key_of(av);
                kmin = ka < kmin ? ka : kmin;
                kmax = ka > kmax ? ka : kmax;
                const K key = key_of(out);
                ++n_all;
                if (key < klo) ++n_below;
                else if (key <= khi) { ++n_in; cand = true; }
            } else {
                out = (T)NAN;
            }
            if (jin) dh[p] = out;
        }
        const unsigned long long mask = __ballot(cand);
        if (mask) {
            if (cand) stage[held + __popcll(mask & ((1ull << lane) - 1ull))] = out;
            held += __popcll(mask);
            if (held > NKF_STAGE / 2) flush();
        }
        rax = rn; t = tn; p = pn;
        a00 = n00; a01 = n01; a10 = n10; a11 = n11; rv = nrv; av = nav; vd = nvd;
    }
    if (held > 0) flush();
    unsigned long long c0 = n_all, c1 = n_below, c2 = n_in;
    for (int off = 32; off > 0; off >>= 1) {
        const K a = k_shfl_down(kmin, off), b = k_shfl_down(kmax, off);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
        c0 += __shfl_down(c0, off); c1 += __shfl_down(c1, off); c2 += __shfl_down(c2, off);
    }
    if (lane == 0) {
        if (kmin != ~(K)0) k_atomic_min(&stats->asp_min, (uint64_t)kmin);
        if (kmax != 0) k_atomic_max(&stats->asp_max, (uint64_t)kmax);
        if (c0) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[0]), c0);
        if (c1) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[1]), c1);
        if (c2) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[2]), c2);
    }
}

// ---- min / max aspect without reading the aspect raster every step (round 3) -----------------------------------------------
// The dh pass needs min / max of the aspect over the pixels whose dh is finite (SciPy's bin edges) -- two numbers, each set by
// ONE pixel -- and used to read the 4-byte aspect of every pixel for them.  At plan creation the valid pixels whose aspect lies
// in the lowest / highest ~16 K of the raster are listed (EXT lists); a step evaluates dh at those few pixels only: the minimum
// over the listed pixels with a finite dh IS the minimum over all of them as long as one listed pixel survives (every unlisted
// pixel has a larger aspect); if none survives, or a list came out empty / overfull, the step falls back to the kernel that
// reads the aspect (flag bit 2 -> the step's second attempt).  The same kernel folds the valid mask into a plan-owned copy of
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

// ---- the lean form of the pass above for NaN rules 0 and 1 (rule 2 reads a 3 x 3 neighbourhood per pixel and keeps the generic
// kernel).  Measured on MI355X the generic kernel is VALU-bound (about 166 vector instructions per row of 64 pixels: per-row tap
// geometry recomputed by every lane, four float64 lerps, 64-bit addressing), not HBM-bound.  Here
//   * the row part of the tap geometry (fraction, upper tap row, flags) is computed once per workgroup into an LDS table;
//   * the horizontal lerp of a tap row is carried in float64 from one output row to the next -- the lower tap row of row i is
//     the upper tap row of row i + 1 and the column fraction never changes -- so a row costs two tap loads, one horizontal
//     and one vertical lerp (same operations in the same order as bi_value: results are bit-identical);
//   * addresses are a uniform row base plus constant 32-bit lane offsets; counters are wave-level popcounts of ballots;
//   * the loads of row i + 1 are issued before row i is evaluated.
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
constexpr int NK_CHUNK_MAX = 512;
constexpr int NK_PF = 4;
constexpr int NKL_ROWS = 4;      // rows between two looks at the staging buffer
constexpr int NKL_CAP = 4096;    // staging slots per workgroup (flushed once fewer than 2 x NKL_ROWS rows would still fit)
template <typename T, int RULE, bool EXT = false>   // EXT: `ref` is the masked copy (NaN where not valid), no mask / aspect reads
__global__ __launch_bounds__(256) void nk_dh_count_lean_kernel(const T* __restrict__ ref, const T* __restrict__ tba,
                                                               const uint8_t* __restrict__ valid, const T* __restrict__ aspect,
                                                               NkGeom g, T* __restrict__ dh, DhStats* stats, int64_t row0, int64_t row1,
                                                               int64_t nbuf, const typename KeyT<T>::type* __restrict__ klo_p,
                                                               const typename KeyT<T>::type* __restrict__ khi_p, uint64_t* counters /* [3] */,
                                                               T* out_v, unsigned long long* ctr /* [1] candidates, [2] overflow */,
                                                               int64_t cap, const uint64_t* __restrict__ badbits = nullptr, int64_t bad_wpr = 0) {
    typedef typename KeyT<T>::type K;
    __shared__ NkRowTab tab[NK_CHUNK_MAX + 1];
    // candidates collect in a workgroup staging buffer and leave in bursts of more than NKL_FLUSH values: one global atomic per
    // burst (an atomic per wave and row group would serialise on the one counter, ~10 ns each)
    __shared__ T stage[NKL_CAP];
    __shared__ int s_held;
    __shared__ unsigned long long s_base;
    __shared__ unsigned long long s_red[4][5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_held = 0;
    const K klo = *klo_p, khi = *khi_p;
    const int64_t chunk = (row1 - row0 + gridDim.y - 1) / gridDim.y;  // <= NK_CHUNK_MAX (launcher)
    const int64_t i0 = row0 + (int64_t)blockIdx.y * chunk;
    const int nrow = (int)((i0 + chunk < row1 ? i0 + chunk : row1) - i0);
    for (int r = threadIdx.x; r <= nrow && r <= NK_CHUNK_MAX; r += blockDim.x) {
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
    if (nrow <= 0) return;  // (uniform over the workgroup)
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool jin = j < g.W;
    const BiAxis col = bi_axis(j, g.dc, g.W, RULE);
    const bool cin = col.in & jin;
    // rules 2 / 3: word and bit of this lane's nearest column in a row of the bad-bit mask (columns left of the raster -> the left pad word)
    int64_t cnc = nk_nearest(col.pos);
    cnc = cnc < -1 ? -1 : (cnc > g.W ? g.W : cnc);
    const uint32_t bad_ob = (uint32_t)(2 + (cnc >> 5)) * 4u;   // byte offset in the row (32-bit halves of the words: one register per row in flight)
    const int bad_sh = (int)(cnc & 31);
    const char* const bad_base = reinterpret_cast<const char*>(badbits);
    const int64_t bad_rowb = bad_wpr * 8;
    const uint32_t c0 = cin ? (uint32_t)col.k0 : 0u, c1 = c0 + (cin ? (uint32_t)col.d1 : 0u);
    const uint32_t jl = jin ? (uint32_t)j : 0u;
    const double fc = col.f;
    auto hlerp = [&](T a, T b) -> double {
        const double v0 = a, v1 = b;
        return t_add(v0, t_mul(fc, t_sub(v1, v0)));
    };
    T fmin_a = (T)INFINITY, fmax_a = -(T)INFINITY;
    uint32_t n_all = 0, n_below = 0, n_in = 0;  // wave-uniform
    auto block_flush = [&](int threshold) {  // every thread of the workgroup
        __syncthreads();
        const int h = s_held;
        if (h > threshold) {
            if (threadIdx.x == 0) s_base = atomicAdd(&ctr[1], (unsigned long long)h);
            __syncthreads();
            const unsigned long long b0 = s_base;
            for (int k = threadIdx.x; k < h; k += blockDim.x) {
                if ((int64_t)(b0 + k) < cap) out_v[b0 + k] = stage[k];
                else ctr[2] = 1ull;
            }
            __syncthreads();
            if (threadIdx.x == 0) s_held = 0;
            __syncthreads();
        }
    };
    // state carried down the column: the horizontal lerp `hl` of buffer tap row `have`
    int have = -1;
    double hl = 0.0;
    // Software pipeline, NK_PF rows deep: the loads of row r + NK_PF are issued while row r is evaluated (one row in flight per
    // wave leaves the kernel latency-bound at about half of the HBM rate: 32 waves x 1 KiB per CU in flight against ~2 us).
    struct Pre { T b0, b1, rv, av; uint8_t vd; uint32_t bw; };
    Pre pre[NK_PF];
    const int64_t rb0 = (i0 - g.roff) * g.W;
    auto issue = [&](int rr, Pre& q) {  // rr clamped: entry `nrow` of the table repeats the last row
        const int rc = rr < nrow ? rr : nrow - 1;
        const int tk = __builtin_amdgcn_readfirstlane(tab[rc].k0l), tf = __builtin_amdgcn_readfirstlane(tab[rc].flags);
        if (RULE == 2) {
            const int rnl = __builtin_amdgcn_readfirstlane(tab[rc].rnl);
            q.bw = rnl >= 0 ? *reinterpret_cast<const uint32_t*>(bad_base + (int64_t)rnl * bad_rowb + bad_ob) : ~0u;
        } else {
            q.bw = 0;
        }
        const T* rowp = tba + (int64_t)(tk + ((tf >> 1) & 1)) * g.W;
        const int64_t rb = rb0 + (int64_t)rc * g.W;
        // (streaming hints: every input of this pass is read once per step)
        q.b0 = __builtin_nontemporal_load(rowp + c0); q.b1 = __builtin_nontemporal_load(rowp + c1);
        q.rv = __builtin_nontemporal_load(ref + rb + jl);
        if (!EXT) { q.av = __builtin_nontemporal_load(aspect + rb + jl); q.vd = __builtin_nontemporal_load(valid + rb + jl); }
        else { q.av = (T)0; q.vd = 1; }
    };
#pragma unroll
    for (int u = 0; u < NK_PF; ++u) issue(u, pre[u]);
    for (int r0 = 0; r0 < nrow; r0 += NK_PF) {
#pragma unroll
        for (int u = 0; u < NK_PF; ++u) {
            const int r = r0 + u;
            if (r < nrow) {
                const T b0v = pre[u].b0, b1v = pre[u].b1, rv = pre[u].rv, av = pre[u].av;
                const uint8_t vd = pre[u].vd;
                const bool nb_clean = RULE != 2 || ((pre[u].bw >> bad_sh) & 1u) == 0;   // rules 2 / 3: neighbourhood of the nearest pixel
                issue(r + NK_PF, pre[u]);
                const int k0l = __builtin_amdgcn_readfirstlane(tab[r].k0l), fl = __builtin_amdgcn_readfirstlane(tab[r].flags);
                const double fr = tab[r].fr;
                const int64_t rb = rb0 + (int64_t)r * g.W;
                double top;
                if (have == k0l) {
                    top = hl;
                } else {  // chunk start, or a step of the tap row other than +1 (pos = i + dr is rounded per row): fetch the upper row
                    const T* up = tba + (int64_t)k0l * g.W;
                    top = hlerp(up[c0], up[c1]);
                }
                double bot = top;
                if (fl & 2) bot = hlerp(b0v, b1v);
                have = k0l + ((fl >> 1) & 1);
                hl = bot;
                const T val = (T)t_add(top, t_mul(fr, t_sub(bot, top)));
                T out = t_sub(rv, val);
                // (a non-finite tap makes val, hence out, non-finite by itself -- also through a zero weight: 0 * NaN = 0 * inf = NaN)
                const bool ok = ((fl & 1) != 0) & cin & (vd != 0) & t_finite(out) & nb_clean;
                const K key = key_of(out);
                const bool below = ok & (key < klo);
                const bool cand = ok & (key >= klo) & (key <= khi);
                out = ok ? out : (T)NAN;
                if (jin) __builtin_nontemporal_store(out, dh + rb + jl);
                if (!EXT) {
                    fmin_a = fmin(fmin_a, ok ? av : (T)INFINITY);
                    fmax_a = fmax(fmax_a, ok ? av : -(T)INFINITY);
                }
                n_all += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(ok));
                n_below += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(below));
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(cand);
                if (mask) {
                    const int c = __popcll(mask);
                    int pos0 = 0;
                    if (lane == 0) pos0 = atomicAdd(&s_held, c);
                    pos0 = __builtin_amdgcn_readfirstlane(pos0);
                    if (cand) stage[pos0 + __popcll(mask & ((1ull << lane) - 1ull))] = out;
                    n_in += (uint32_t)c;
                }
            }
            // (r is uniform over the workgroup: every wave walks the same rows) room for NKL_ROWS more rows must remain
            if (((r + 1) % NKL_ROWS) == 0 && r + 1 < nrow) block_flush(NKL_CAP - 2 * NKL_ROWS * 256);
        }
    }
    block_flush(0);
    K kmin = (fmin_a <= fmax_a) ? key_of(fmin_a) : ~(K)0;
    K kmax = (fmin_a <= fmax_a) ? key_of(fmax_a) : (K)0;
    for (int off = 32; off > 0; off >>= 1) {
        const K a = k_shfl_down(kmin, off), b = k_shfl_down(kmax, off);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    // one set of global atomics per workgroup
    if (lane == 0) {
        s_red[wave][0] = (uint64_t)kmin; s_red[wave][1] = (uint64_t)kmax;
        s_red[wave][2] = n_all; s_red[wave][3] = n_below; s_red[wave][4] = n_in;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t mn = ~(uint64_t)0, mx = 0, c0 = 0, c1 = 0, c2 = 0;
        for (int w = 0; w < 4; ++w) {
            const uint64_t a = (s_red[w][0] == (uint64_t)(~(K)0)) ? ~(uint64_t)0 : s_red[w][0];
            mn = a < mn ? a : mn;
            mx = s_red[w][1] > mx ? s_red[w][1] : mx;
            c0 += s_red[w][2]; c1 += s_red[w][3]; c2 += s_red[w][4];
        }
        if (!EXT && mn != ~(uint64_t)0) k_atomic_min(&stats->asp_min, mn);
        if (!EXT && mx != 0) k_atomic_max(&stats->asp_max, mx);
        if (c0) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[0]), (unsigned long long)c0);
        if (c1) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[1]), (unsigned long long)c1);
        if (c2) atomicAdd(reinterpret_cast<unsigned long long*>(&counters[2]), (unsigned long long)c2);
    }
}

// The same dh as an element source of the bracketed selection's SAMPLE pass (select_run.h): evaluated on the sampled lines
// only, before the dh raster exists, so that nk_dh_count_kernel can count against the bracket while it computes dh.
template <typename T> struct NkDhSource {
    typedef typename KeyT<T>::type K;
    const T* ref; const T* tba; const uint8_t* valid; const T* aspect;
    T* dh;
    NkGeom g;
    int64_t q0;       // local index of element 0 of the pass (this rank's first own pixel)
    double invW;
    DhStats* stats;
    struct Raw { T a00, a01, a10, a11, rv, av; BiTap t; int64_t q; uint8_t vd; };
    struct Acc { K kmin = ~(K)0, kmax = 0; };
    static size_t lds_bytes(int) { return 0; }
    static constexpr bool HAS_LEAN = false;
    __device__ __forceinline__ void setup(unsigned char*, int) {}
    __device__ __forceinline__ void fetch(int64_t p, Raw& r) const {
        r.q = q0 + p;
        int64_t li, j;
        row_col(r.q, g.W, invW, li, j);
        r.t = bi_locate(g, li + g.roff, j);
        const BiVals<T> tv = bi_load<T>(tba, r.t);
        r.a00 = tv.a00; r.a01 = tv.a01; r.a10 = tv.a10; r.a11 = tv.a11;
        r.rv = ref[r.q];
        r.av = aspect[r.q];
        r.vd = valid[r.q];
    }
    __device__ __forceinline__ void blank(Raw& r) const { r.q = -1; }
    template <bool ACC> __device__ __forceinline__ bool eval(const Raw& r, int, T& v, uint16_t& b, Acc& acc) const {
        b = 0;
        v = (T)NAN;
        if (r.q < 0) return false;
        T val;
        const bool ok = bi_value<T>(g, tba, r.t, r.a00, r.a01, r.a10, r.a11, val) && r.vd;
        T out = t_sub(r.rv, val);
        const bool keep = ok && t_finite(out);
        if (!keep) out = (T)NAN;
        if (ACC) {
            dh[r.q] = out;
            if (keep) {
                const K ka = key_of(r.av);
                acc.kmin = ka < acc.kmin ? ka : acc.kmin;
                acc.kmax = ka > acc.kmax ? ka : acc.kmax;
            }
        }
        v = out;
        return keep;
    }
    __device__ __forceinline__ void finish(Acc& acc) const {
        K kmin = acc.kmin, kmax = acc.kmax;
        for (int off = 32; off > 0; off >>= 1) {
            const K a = k_shfl_down(kmin, off), c = k_shfl_down(kmax, off);
            kmin = a < kmin ? a : kmin;
            kmax = c > kmax ? c : kmax;
        }
        if ((threadIdx.x & 63) == 0) {
            if (kmin != ~(K)0) k_atomic_min(&stats->asp_min, (uint64_t)kmin);
            if (kmax != 0) k_atomic_max(&stats->asp_max, (uint64_t)kmax);
        }
    }
};

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
// = 128 bins, 0xFF = "no bin" -- the one-pass step touches 13 instead of 14 bytes per pixel, the bin pass of the two-pass route 9
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
    const T* vshift_p;  // device scalar (written by nk_vshift_edges_kernel, or uploaded by the host on the plain route)
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
// known exactly (selection among the dh candidates, as before), nk_resolve_kernel evaluates the candidates' y in the
// reference's arithmetic, counts those below / inside [lo_b, hi_b] and hands the inside ones to the same exact selection as
// before: rank (k - certainly below - candidates below) among them.  All counting is integer; every rank claim is checked by
// the counts (bracket_given_kernel); a miss or an overflow anywhere sends the step to the two-pass route of round 3.
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
    if (ext_survivors[0] == 0 || ext_survivors[1] == 0) ctr[3] = 1ull;   // min / max aspect unknown on this route
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
    // round 5, on the side (two launches less): v^ and delta from the dh sample's bracket, which nk_vhat_kernel used to derive -- every
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
        T v, d;
        if (!(pr.dlo <= pr.dhi) || !nk_vhat_of<T>(1u, lo, hi, v, d)) ctr[3] = 1ull;
        *vhat = v;
        *delta = d;
    }
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
template <typename T, int RULE>   // (RULE 2 = rules 2 / 3 through the bad-bit mask: six more registers -> one workgroup per CU fewer instead of spills)
__global__ __launch_bounds__(256, (RULE == 2 ? XD_NKZ_LB - 1 : XD_NKZ_LB)) void nk_fused_kernel(const T* __restrict__ ref, const T* __restrict__ tba, const T* __restrict__ slope_tan,
                                                       const nk_bin_t* __restrict__ bcache, NkGeom g, int64_t row0, int64_t row1, int64_t nbuf,
                                                       int nb, int copies, const typename KeyT<T>::type* __restrict__ klo_p,
                                                       const typename KeyT<T>::type* __restrict__ khi_p, const T* vhat_p, const T* delta_p,
                                                       const typename KeyT<T>::type* __restrict__ klo_y,
                                                       const typename KeyT<T>::type* __restrict__ khi_y, uint64_t* cnt_d /* [3] */,
                                                       uint64_t* cls_y /* [3][nb]: above, below, candidates */, T* cd_vals, int64_t cd_cap,
                                                       T* cy_d, T* cy_st, uint16_t* cy_b, int64_t cy_cap,
                                                       unsigned long long* ctr /* [1] dh candidates, [5] y candidates, [2] overflow */,
                                                       double* sums /* [5] */, const uint64_t* __restrict__ badbits = nullptr, int64_t bad_wpr = 0) {
    typedef typename KeyT<T>::type K;
    constexpr int NKZ_CAP = NkzCap<T>::v;
    constexpr int SEG = NKZ_CAP / 4;      // staging slots of ONE wave: waves reserve in their own segment with a scalar counter --
                                          // no LDS atomic with return and its round trip on the path of every row (bin candidates are
                                          // ~7 % of the pixels: practically every row of every wave holds some)
    // (a look at the staging buffers every NKZ_ROWS rows, a flush when a segment is half full; a wave that would overrun its segment
    // before the next look -- more than half of its pixels candidates over NKZ_ROWS rows: not a raster this route is for -- raises
    // the overflow flag and the step falls through to the two-pass route)
    static_assert(SEG >= 2 * 64, "a segment holds at least two rows of candidates");
    // candidates of the median of dh are ~0.7 % of the pixels (half a pixel per wave and row): their segments are small and a wave
    // that would overrun its segment between two looks -- more than half of its pixels inside the bracket of the median: a raster
    // of (nearly) one dh value -- raises the overflow flag, i.e. hands the step to the two-pass route, which is built for that
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
    if (nrow <= 0) return;  // (uniform over the workgroup)
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
    auto tap_row = [&](int k) -> uint32_t { return (uint32_t)((k > k_base ? k : k_base) - k_base) * wbytes; };   // (scalar)
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
    if (threadIdx.x < 5) atomicAdd(&sums[threadIdx.x], s_sum[0][threadIdx.x] + s_sum[1][threadIdx.x] + s_sum[2][threadIdx.x] + s_sum[3][threadIdx.x]);
    for (int k = threadIdx.x; k < 3 * nb; k += blockDim.x) {
        unsigned long long t = 0;
        for (int q = 0; q < copies; ++q) t += c[q * cs + k];
        if (t) atomicAdd(reinterpret_cast<unsigned long long*>(&cls_y[k]), t);
    }
}




// ---- round 5: the bin candidates partitioned by bin; ONE workgroup per bin selects its exact median ----------------------------
// After nk_resolve_kernel the exact selection among the candidates of the 72 bins used to run as select_enqueue: three more digit
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
                else ctr[2] = 1ull;   // (cannot happen for consistent counters: the step then takes the two-pass route)
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
    if (s_pick[2] > (unsigned long long)CAPK) {   // (a bucket beyond the LDS buffer -- ties en masse: the two-pass route takes the step)
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
// in LDS and writes vshift the way nk_fz_vshift_kernel does.  (Gather and final in one launch -- the last workgroup to finish, by a
// ticket, doing the final part -- needs the keys, plain stores of many workgroups, published by device-scope fences: on this
// multi-XCD part a fence writes back the issuing XCD's whole L2, 38-55 us for 64 workgroups, measured; a kernel boundary is cheaper.)
// Same integers, same keys: the result is the generic selection's bit for bit (GPU tests: every route agrees).
constexpr int DSEL_BUCKETS = 4096;
constexpr int DSEL_CAP = 8192;        // keys of the chosen bucket (more -- ties en masse -- send the step to the two-pass route)
constexpr int DSEL_HDR_WORDS = 4;     // 64-bit words: [0] ~(smallest key IN the bucket), [1] keys appended, [2] ~(smallest key above the bucket), [3] largest key in the bucket; then the histogram
// Two forms of the monotone bucket map.  VALUE buckets (floor((v - lo) * 4096 / (hi - lo)) in float64) for brackets that straddle
// zero or span many binades -- there the float KEYS are spread exponentially and most values would sit in a few key buckets.
// KEY buckets ((key - klo) >> s, s the smallest shift that brings the bracket under 4096 buckets) for brackets inside at most three
// binades of one sign -- there keys are evenly dense, and this is where ties concentrate: a well-aligned pair (the state every fit
// converges to) has dh = offset + small noise, a bracket a few hundred float32 values wide holding a million candidates.  With
// s = 0 a bucket IS a key: the histogram alone gives the selected key, its duplicates and its successor -- nothing is gathered,
// whatever the multiplicity (`exact`).  (Found by bench.py's whole-fit leg, whose later iterations fell to the two-pass route with
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
                                                                     uint32_t* hist /* [DSEL_BUCKETS], zeroed */) {
    typedef typename KeyT<T>::type K;
    __shared__ uint32_t h[DSEL_BUCKETS];
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
    if (s_pick[2] == ~0ull) {   // no bucket found (inconsistent histogram): the two-pass route
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
    auto hand_back = [&](T vs) {   // what nk_fz_vshift_kernel writes
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
    if (big) {   // more values than the key buffer takes: one key many times over, or the two-pass route
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
    if (tid == 0) {   // (the arithmetic of nk_fz_vshift_kernel)
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
// With a reduction hook (row blocks over ranks, one process per GPU) the step used to take the two-pass route: ~25 small all-reduces and
// two data passes.  The one-pass step needs, besides its data pass over THIS rank's rows, global counters and global order statistics
// among candidates that are spread over the ranks.  Everything that crosses ranks is an all-reduce of 8-byte words through the hook
// (sums), TEN per step:
//   1-3   the dh sample's dual bracket selection: one histogram per digit        (select_enqueue, as on the two-pass route); the first
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
// from reduced data and are identical on every rank, so all ranks fall through to the two-pass route together or not at all.
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
static __global__ __launch_bounds__(256) void nk_mr_counts_pack_kernel(const uint64_t* cnt_d, const uint64_t* cls_y, const double* sums, int nb, int rank,
                                                                        int world, uint64_t* red, uint64_t* cls_loc) {
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

// per bin: total / below / inside in the layout bracket_given_kernel reads, from the classes of the pass and of the candidates

}  // namespace xd

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
    unsigned char* fz_pack = nullptr;   // ... the step's small results, gathered for one device-to-host copy
    size_t fz_pack_bytes = 0;
    std::vector<unsigned char> fz_host;
    int64_t n_onepass = 0, n_twopass = 0, n_plain = 0;   // steps answered by each route (xdemhip_nk_route_counts)
    // one-pass route: the sample brackets of the NEXT step are 2^-fz_narrow as wide as the rule for fully correlated sample lines
    // asks (select.h: sel_bracket_halfwidth); set from how far off the bracket centres the wanted ranks lay in the steps so far
    int fz_narrow = 0, fz_narrow_cap = 2;
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

// Device-side epilogue of the fused global-median stage: vertical shift (np.nanmedian of dh, from the counters and the
// candidate selection -- the arithmetic of run_select_bracketed's host epilogue and of median_from) and SciPy's bin edges
// from the min / max aspect, so that the per-bin stage can be queued without a host round trip.
template <typename T>
__global__ void nk_vshift_edges_kernel(const uint64_t* cnt /* total, below, inside */, const SelState<typename KeyT<T>::type>* st,
                                       const uint64_t* succ, const typename KeyT<T>::type* klo, const uint32_t* rbs_p,
                                       const unsigned long long* flags, const DhStats* stats, int nb, unsigned char* info, T* edges,
                                       const unsigned long long* ext_survivors = nullptr) {
    typedef typename KeyT<T>::type K;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint64_t total = cnt[0], lt = cnt[1];
    const int rbs = (int)*rbs_p;
    T vs = (T)NAN;
    if (total) {
        const K prefix = (K)((K)(st[0].prefix >> rbs) + klo[0]);
        const uint64_t n_le = st[0].n_le + lt;
        const T lo = val_of(prefix);
        if (total & 1) vs = lo;
        else {
            const uint64_t k2 = total / 2;
            T hi = lo;
            if (!(n_le > k2)) hi = val_of((K)((K)((K)succ[0] >> rbs) + klo[0]));
            vs = (T)((T)(lo + hi) / (T)2);
        }
    }
    *reinterpret_cast<T*>(info) = vs;
    *reinterpret_cast<uint64_t*>(info + 8) = total;
    // bit 2: EXT route and no listed pixel of a list kept a finite dh (with finite dh at all): min / max aspect unknown
    const bool ext_miss = ext_survivors && total && (ext_survivors[0] == 0 || ext_survivors[1] == 0);
    *reinterpret_cast<uint64_t*>(info + 16) = (uint64_t)((flags[2] != 0) | ((flags[3] != 0) << 1) | ((ext_miss ? 1 : 0) << 2));
    *reinterpret_cast<double*>(info + 24) = (double)vs;
    make_edges_into<T>((double)val_of((K)stats->asp_min), (double)val_of((K)stats->asp_max), nb, edges);
}

// Stage 1+2 of a step, fused: sample of dh on the sampled lines (computed on the fly) -> bracket of its median -> ONE pass
// that computes dh for every own pixel, writes it, tracks min / max aspect and counts / compacts the bracket's candidates ->
// exact selection among the candidates -> vshift and bin edges on the device.  Returns *queued = false when the bracketed
// route does not apply (small grid, plain mode, no workspace): the caller then runs the separate passes.
template <typename T>
int nk_global_fused(xdemhip_nk_plan* P, const NkGeom& g, int64_t q0, int64_t n, int nb, bool* queued) {
    typedef typename KeyT<T>::type K;
    xdemhip_ctx* ctx = P->ctx;
    SelWorkspace* ws = &P->ws;
    unsigned char* scratch = static_cast<unsigned char*>(P->scratch);
    *queued = false;
    const bool plain = ctx->selection_mode == 1 || ctx->selection_mode == 2 || !ws->d_small || ws->es != sizeof(T) ||
                       n < SEL_BRACKET_MIN_N || (n / 24 + 4096) > ws->s_cap;
    if (ctx->allreduce) {  // sharded data: every rank must take the same route
        uint64_t can = plain ? 0 : 1;
        if (ctx->allreduce(&can, 1, XDEMHIP_RED_MIN_U64, ctx->allreduce_user) != 0) return xd_fail(ctx, XDEMHIP_EHIP, "all-reduce hook failed");
        if (!can) return XDEMHIP_OK;
    } else if (plain) {
        return XDEMHIP_OK;
    }
    DhStats* d_stats = reinterpret_cast<DhStats*>(scratch + OFF_STATS);
    NkDhSource<T> src{static_cast<const T*>(P->ref), static_cast<const T*>(P->tba), P->valid, static_cast<const T*>(P->aspect),
                      static_cast<T*>(P->dh), g, q0, 1.0 / (double)P->W, d_stats};
    uint64_t* d_ctr = ws->d_small;
    unsigned long long* d_flags = reinterpret_cast<unsigned long long*>(d_ctr);
    K* d_klo = reinterpret_cast<K*>(ws->d_small + 8);
    K* d_khi = reinterpret_cast<K*>(ws->d_small + 8 + ws->nb_max);
    uint64_t* d_given = ws->d_small + 8 + 2 * ws->nb_max;
    uint64_t* d_cnt = ws->d_small + 8 + 3 * ws->nb_max;
    uint32_t* d_rbs = reinterpret_cast<uint32_t*>(ws->d_small + 4);
    const SelState<K>* d_st = reinterpret_cast<const SelState<K>*>(scratch + OFF_STATE);
    XD_HIP_CHECK(ctx, hipMemsetAsync(ws->d_small, 0, (size_t)(8 + 6 * ws->nb_max) * 8, ctx->stream));
    const size_t lds_stage = (size_t)SEL_STAGE_CAP * (sizeof(T) + 2) + 16;
    int rc = set_big_lds(ctx, sample_lines_kernel<T, NkDhSource<T>>, lds_stage);
    if (rc) return rc;
    hipLaunchKernelGGL((sample_lines_kernel<T, NkDhSource<T>>), dim3(grid_for(ctx, n / 64 + 1, HIST_THREADS, 2)), dim3(HIST_THREADS), lds_stage,
                       ctx->stream, src, n, 1, static_cast<T*>(ws->s_vals), ws->s_bins, d_flags, ws->s_cap);
    XD_HIP_CHECK(ctx, hipGetLastError());
    const int64_t m_est = n / 48 + 1;
    constexpr int BR_PASSES = 3;
    const K low_mask = (K)(((K)1 << (8 * (KeyT<T>::passes - BR_PASSES))) - 1);
    // both ends of the bracket in ONE selection over the sample (two states, every element offered to both)
    rc = select_enqueue<T>(ctx, static_cast<const T*>(ws->s_vals), nullptr, ws->s_cap, m_est, d_flags + 0, 1, scratch, SEL_BRACKET_DUAL, nullptr,
                           BR_PASSES, false);
    if (rc) return rc;
    hipLaunchKernelGGL((bracket_finish_kernel<K>), dim3(1), dim3(64), 0, ctx->stream, d_st, 1, 0, low_mask, d_klo, d_khi, d_rbs);
    XD_HIP_CHECK(ctx, hipGetLastError());
    // the one pass: dh for every own pixel (written), min / max aspect, counters, candidates
    bool ext_used = false;
    if (P->row1 > P->row0) {
        dim3 grid = grid2d(ctx, P->W, P->row1 - P->row0);
        const int64_t rows = P->row1 - P->row0;
        if ((rows + grid.y - 1) / grid.y > NK_CHUNK_MAX) grid.y = (unsigned)((rows + NK_CHUNK_MAX - 1) / NK_CHUNK_MAX);
        const bool ext = P->ext_ok && !ctx->allreduce && (g.rule <= 1 || P->badbits);
        if (ext) {   // min / max aspect from the listed extreme-aspect pixels; the dh pass then reads neither mask nor aspect
            XD_HIP_CHECK(ctx, hipMemsetAsync(P->ext_cnt + 2, 0, 16, ctx->stream));
            hipLaunchKernelGGL((nk_ext_eval_kernel<T>), dim3(EXT_CAP / 256, 2), dim3(256), 0, ctx->stream, static_cast<const T*>(P->ref_m),
                               static_cast<const T*>(P->tba), static_cast<const T*>(P->aspect), g, P->ext_idx, P->ext_cnt, d_stats, P->ext_cnt + 2);
            XD_HIP_CHECK(ctx, hipGetLastError());
        }
        ext_used = ext;
#define XD_NK_LEAN(RULE)                                                                                                            \
    do {                                                                                                                            \
        if (ext)                                                                                                                    \
            hipLaunchKernelGGL((nk_dh_count_lean_kernel<T, RULE, true>), grid, dim3(256), 0, ctx->stream, static_cast<const T*>(P->ref_m), \
                               static_cast<const T*>(P->tba), P->valid, static_cast<const T*>(P->aspect), g, static_cast<T*>(P->dh), d_stats, \
                               P->row0, P->row1, P->nbuf, d_klo, d_khi, d_cnt, static_cast<T*>(ws->c_vals), d_flags, ws->c_cap,    \
                               P->badbits, P->bad_wpr);                                                                            \
        else                                                                                                                        \
            hipLaunchKernelGGL((nk_dh_count_lean_kernel<T, RULE>), grid, dim3(256), 0, ctx->stream, static_cast<const T*>(P->ref),  \
                               static_cast<const T*>(P->tba), P->valid, static_cast<const T*>(P->aspect), g, static_cast<T*>(P->dh), d_stats, \
                               P->row0, P->row1, P->nbuf, d_klo, d_khi, d_cnt, static_cast<T*>(ws->c_vals), d_flags, ws->c_cap,    \
                               P->badbits, P->bad_wpr);                                                                            \
    } while (0)
        if (g.rule == 0) XD_NK_LEAN(0);
        else if (g.rule == 1) XD_NK_LEAN(1);
        else if (P->badbits) XD_NK_LEAN(2);   // rules 2 / 3 through the bad-bit mask of the plan (whole-raster plans, with or without a reduction hook)
        else
            hipLaunchKernelGGL((nk_dh_count_kernel<T>), grid, dim3(256), 0, ctx->stream, static_cast<const T*>(P->ref),
                               static_cast<const T*>(P->tba), P->valid, static_cast<const T*>(P->aspect), g, static_cast<T*>(P->dh),
                               d_stats, P->row0, P->row1, d_klo, d_khi, d_cnt, static_cast<T*>(ws->c_vals), d_flags, ws->c_cap);
#undef XD_NK_LEAN
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    rc = xd_allreduce_device(ctx, d_cnt, 3, XDEMHIP_RED_SUM_U64);
    if (rc) return rc;
    rc = xd_allreduce_device(ctx, d_ctr + 2, 1, XDEMHIP_RED_SUM_U64);
    if (rc) return rc;
    rc = xd_allreduce_device(ctx, &d_stats->asp_min, 1, XDEMHIP_RED_MIN_U64);
    if (rc) return rc;
    rc = xd_allreduce_device(ctx, &d_stats->asp_max, 1, XDEMHIP_RED_MAX_U64);
    if (rc) return rc;
    hipLaunchKernelGGL(bracket_given_kernel, dim3(1), dim3(64), 0, ctx->stream, d_cnt, 1, d_given, d_flags);
    XD_HIP_CHECK(ctx, hipGetLastError());
    rc = select_enqueue<T>(ctx, static_cast<const T*>(ws->c_vals), nullptr, ws->c_cap, n / 32 + 1, d_flags + 1, 1, scratch, SEL_GIVEN, d_given,
                           0, true, d_klo, d_rbs);
    if (rc) return rc;
    hipLaunchKernelGGL((nk_vshift_edges_kernel<T>), dim3(1), dim3(64), 0, ctx->stream, d_cnt, d_st,
                       reinterpret_cast<const uint64_t*>(scratch + off_succ(1)), d_klo, d_rbs, d_flags, d_stats, nb, scratch + OFF_INFO,
                       reinterpret_cast<T*>(scratch), ext_used ? P->ext_cnt + 2 : nullptr);
    XD_HIP_CHECK(ctx, hipGetLastError());
    *queued = true;
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

// Stages 1 + 2 of a step: dh at the shifted position and its exact nanmedian.  On return the device holds dh, the vertical
// shift (OFF_INFO), the bin edges (scratch start; the plan's custom edges if set) and, at OFF_INFO + 8 / + 16, the valid
// count and the bracket flags.  *fused tells whether everything was only queued (fused route: flags must be checked after
// the next synchronisation) or computed through the host (plain route: flags are 0).
template <typename T>
int nk_stage_a(xdemhip_nk_plan* P, const NkGeom& g, int64_t q0, int64_t n, int nb, bool force_plain, bool* fused) {
    typedef typename KeyT<T>::type K;
    xdemhip_ctx* ctx = P->ctx;
    unsigned char* base = static_cast<unsigned char*>(P->scratch);
    DhStats* d_stats = reinterpret_cast<DhStats*>(base + OFF_STATS);
    T* d_edges = reinterpret_cast<T*>(base);
    DhStats hs0;
    hipLaunchKernelGGL(nk_stats_init_kernel, dim3(1), dim3(64), 0, ctx->stream, d_stats);
    XD_HIP_CHECK(ctx, hipGetLastError());
    *fused = false;
    int rc = XDEMHIP_OK;
    if (!force_plain) {
        rc = nk_global_fused<T>(P, g, q0, n, nb, fused);
        if (rc) return rc;
    }
    if (!*fused) {
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
// then runs the two-pass route of round 3, which needs nothing from here.
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
    unsigned long long* ctr = reinterpret_cast<unsigned long long*>(fz);   // [1] dh candidates, [2] overflow, [3] miss, [5] y candidates
    uint32_t* rbs_d = reinterpret_cast<uint32_t*>(fz + 8);
    uint32_t* rbs_y = reinterpret_cast<uint32_t*>(fz + 9);
    uint64_t* cnt_d = fz + 10;
    uint64_t* given_d = fz + 13;
    K* klo_d = reinterpret_cast<K*>(fz + 14);
    K* khi_d = reinterpret_cast<K*>(fz + 15);
    T* d_vhat = reinterpret_cast<T*>(fz + 16);
    T* d_delta = reinterpret_cast<T*>(fz + 17);
    double* d_sums = reinterpret_cast<double*>(fz + 18);
    K* klo_y = reinterpret_cast<K*>(fz + 24);
    K* khi_y = reinterpret_cast<K*>(fz + 24 + nbm);
    uint64_t* given_y = fz + 24 + 2 * nbm;
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
    hipLaunchKernelGGL((nk_sample_dh_kernel<T>), dim3(grid_for(ctx, n_slots, 256, 8)), dim3(256), 0, ctx->stream, ref_m, tba, g, q0, n,
                       1.0 / (double)P->W, n_slots, s_v, select_reset_plan<K>(scratch, 1, SEL_BRACKET_DUAL));
    XD_HIP_CHECK(ctx, hipGetLastError());
    if (predict_d) {
        // the dh sample is still taken (the y^ sample is formed from it), its selection is not: bracket from the previous median
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
    hipLaunchKernelGGL((nk_sample_y_kernel<T>), dim3(grid_for(ctx, n_slots, 256, 8)), dim3(256), 0, ctx->stream, s_v, ws->s_bins, st_all + q0,
                       P->bcache + q0, n, n_slots, d_vhat, klo_d, khi_d, d_vhat, d_delta, ctr, select_reset_plan<K>(scratch, nb, SEL_BRACKET_DUAL));
    XD_HIP_CHECK(ctx, hipGetLastError());
    rc = select_enqueue<T>(ctx, s_v, nb == 1 ? nullptr : ws->s_bins, n_slots, n_slots, nullptr, nb, scratch, SEL_BRACKET_DUAL, nullptr, BR_PASSES,
                           false, nullptr, nullptr, false, narrow, klo_y, khi_y, rbs_y, low_mask, &fused, true);
    if (rc) return rc;
    if (!fused) hipLaunchKernelGGL((bracket_finish_kernel<K>), dim3(1), dim3(64), 0, ctx->stream, d_st, nb, 0, low_mask, klo_y, khi_y, rbs_y);
    XD_HIP_CHECK(ctx, hipGetLastError());
    }   // (sampled brackets)
    const int nbb = (nb + 63) / 64;
    // 4. the one pass
    {
        dim3 grid = grid2d(ctx, P->W, rows);
        if ((rows + grid.y - 1) / grid.y > NKZ_CHUNK_MAX) grid.y = (unsigned)((rows + NKZ_CHUNK_MAX - 1) / NKZ_CHUNK_MAX);
        int copies = (4608) / (nb * 12);   // (static 26 KB + this: five workgroups of 256 threads per CU)
        copies = copies < 1 ? 1 : (copies > 16 ? 16 : copies);
        const size_t lds = (size_t)nb * sizeof(FzPair<T>) + (size_t)((3 * nb) | 1) * 4 * (size_t)copies + 64 * 4;
        T* cy_d = static_cast<T*>(ws->c_vals);
#define XD_NK_FZ(RULE)                                                                                                               \
    hipLaunchKernelGGL((nk_fused_kernel<T, RULE>), grid, dim3(256), lds, ctx->stream, ref_m, tba, st_all, P->bcache, g, P->row0, P->row1,  \
                       P->nbuf, nb, copies, klo_d, khi_d, d_vhat, d_delta, klo_y, khi_y, cnt_d, cls_y, static_cast<T*>(P->cd_vals),   \
                       P->cd_cap, cy_d, static_cast<T*>(P->c_st), ws->c_bins, ws->c_cap, ctr, d_sums, P->badbits, P->bad_wpr)
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
        hipLaunchKernelGGL(nk_mr_counts_pack_kernel, dim3(1), dim3(256), 0, ctx->stream, cnt_d, cls_y, d_sums, nb, rank, world, P->mr_a, cls_loc);
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
                           klo_d, khi_d, dsel + 2 * DSEL_HDR_WORDS);
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
    if (h_ctr[2] != 0 || h_ctr[3] != 0) {   // overflow / a bracket missed / no extreme-aspect survivor: the two-pass route
        if (narrow > 0) { P->fz_narrow_cap = narrow - 1; P->fz_narrow = 0; }   // (narrowed brackets may be what missed: not that narrow again)
        P->pr_have = false;
        return XDEMHIP_OK;
    }
    if (!predict) {
        // How centred were the brackets?  |wanted rank - centre| in half widths, scaled to the full rule.  Lines of a sample that
        // are not fully correlated (the rule's worst case) leave the ranks within a small fraction of it: the next step then takes
        // brackets half or a quarter as wide -- fewer candidates staged, resolved and selected from (measured on C3: 1.88 -> 1.73
        // -> 1.62 ms per step); a miss costs that step the two-pass route and caps the narrowing (exact either way).
        auto off_of = [&](uint64_t tot, uint64_t lt, uint64_t in) {
            return fabs(((double)(tot - 1) * 0.5 - (double)lt) - 0.5 * (double)in) / (0.5 * (double)in);
        };
        // (offsets are measured in half widths of the bracket that was used; all statistics are kept in units of the FULL rule)
        const double unit = 1.0 / (double)(1 << narrow);
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
        // 1 / 17, the rule being 6 sigma of 8-element lines that are fully correlated): the next brackets keep >= 7 sigma and
        // >= 2.2 x the worst offset ever seen, in steps of halving
        int next = 0;
        if (P->fz_offn >= 24) {
            const double sigma = sqrt(P->fz_off2 / (double)P->fz_offn);
            for (int k = 2; k >= 1 && next == 0; --k)
                if (7.0 * sigma <= 1.0 / (double)(1 << k) && 2.2 * P->fz_worst <= 1.0 / (double)(1 << k)) next = k;
        }
        next = next > P->fz_narrow_cap ? P->fz_narrow_cap : next;
        if (ctx->nk_narrow >= 0) next = ctx->nk_narrow;   // option "nk_narrow": -1 = this rule, 0 / 1 / 2 fixed
        if (getenv("XDEMHIP_DEBUG"))
            fprintf(stderr, "[xdemhip] one-pass step: brackets 2^-%d, offsets rms %.3f worst %.3f of the full half width (%lld brackets) -> next 2^-%d\n", narrow,
                    P->fz_offn ? sqrt(P->fz_off2 / (double)P->fz_offn) : 0.0, P->fz_worst, (long long)P->fz_offn, next);
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
            P->pr_w.assign(nb, 0.0);
            for (int b = 0; b < nb; ++b)
                if (counts[b] > 0 && khi[b] >= klo[b]) P->pr_w[b] = 0.5 * ((double)val_of(khi[b]) - (double)val_of(klo[b]));
            if (!predict_d) P->pr_wd = h_kd[1] >= h_kd[0] ? 0.5 * ((double)val_of(h_kd[1]) - (double)val_of(h_kd[0])) : 0.0;
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
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool fused = false;
        int rc = nk_stage_a<T>(P, g, q0, n, nb, attempt == 1, &fused);
        if (rc) return rc;
        // stage 3 + 4, queued right behind stage 1 + 2 on the fused route (vshift and the edges are read from the device)
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
        if (fused && (flags & 4)) P->ext_ok = false;  // no listed extreme-aspect pixel kept a finite dh: this plan reads the aspect again
        if (fused && flags != 0) continue;  // a bracket of the global median missed / overflowed: again on the plain route
        if (fused) ++P->n_twopass; else ++P->n_plain;
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
    return xd_fail(ctx, XDEMHIP_EHIP, "Nuth-Kaab step: selection failed on both routes");
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
    // EXT route of the dh pass (large single-GPU plans): masked copy of the reference DEM + the lists of extreme-aspect pixels;
    // without the memory for it the plan simply keeps the route that reads mask and aspect
    if ((int64_t)n >= SEL_BRACKET_MIN_N && ctx->nk_ext != 0) {
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
    // plan keeps the two-pass route
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
    if (P->own_inputs) { (void)hipFree(P->ref); (void)hipFree(P->tba); if (P->inlier) (void)hipFree(P->inlier); }
    void* bufs[] = {P->slope_tan, P->aspect, P->dh, P->y, P->valid, P->bins, P->bcache, P->scratch, P->ref_m, P->ext_idx, P->ext_cnt,
                    P->fz, P->cd_vals, P->c_st, P->fz_pack, P->badbits, P->mr_a, P->mr_b, P->mr_small};
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

int xdemhip_nk_predict_counts(xdemhip_nk_plan* P, int64_t* predicted, int64_t* predicted_dh_only, int64_t* missed) {
    if (!P) return XDEMHIP_EINVAL;
    if (predicted) *predicted = P->n_predicted;
    if (predicted_dh_only) *predicted_dh_only = P->n_predicted_d;
    if (missed) *missed = P->n_predict_miss;
    return XDEMHIP_OK;
}

int xdemhip_nk_route_counts(xdemhip_nk_plan* P, int64_t* onepass, int64_t* twopass, int64_t* plain) {
    if (!P) return XDEMHIP_EINVAL;
    if (onepass) *onepass = P->n_onepass;
    if (twopass) *twopass = P->n_twopass;
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
