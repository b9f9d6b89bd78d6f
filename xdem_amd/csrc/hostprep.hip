// hostprep.hip -- HOST-side preparation of the variogram path in native code (no device code in this file): the equidistant ring
// sampler of raster-sampled variograms and the Morton-ordered copies of the pair blocks.  The reference does this work in Python /
// NumPy inside scikit-gstat's RasterEquidistantMetricSpace (un-vendored; restated in xdem_amd/spatialstats.py, whose NumPy forms
// remain the specification and the fall-back); at BASELINE's C5 sizes (100 runs x 14 draws of 2e4 - 2e5 pixels per variogram) the
// NumPy forms cost 5 s per variogram against 0.15 s of pair kernels, and Python threads do not scale on them.  Here the runs are
// spread over std::threads; every (run, ring) draw has its own counter-based random stream, so the result does not depend on the
// number of threads.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/xdemhip.h"

namespace {

// splitmix64: the stream of a (seed, run, ring) triple
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    // uniform integer in [0, n) without modulo bias (Lemire's multiply-shift with rejection)
    uint64_t below(uint64_t n) {
        uint64_t x = next();
        __uint128_t m = (__uint128_t)x * (__uint128_t)n;
        uint64_t l = (uint64_t)m;
        if (l < n) {
            const uint64_t t = (0 - n) % n;
            while (l < t) { x = next(); m = (__uint128_t)x * (__uint128_t)n; l = (uint64_t)m; }
        }
        return (uint64_t)(m >> 64);
    }
};
inline uint64_t mix3(uint64_t seed, uint64_t a, uint64_t b) {
    Rng r(seed ^ (a * 0xD6E8FEB86659FD93ull) ^ (b * 0xCA5A826395121157ull + 0x632BE59BD9B4E019ull));
    r.next();
    return r.next();
}

// The ring lo <= d < hi around pixel (cx, cy) as two column spans per raster row, taken one pixel generous (the exact distance
// test decides): the closed form of xdem_amd/spatialstats.py: _draw_ring_pixels.
struct Spans {
    int64_t y0 = 0;
    std::vector<int64_t> la, nl, ra, nr, cum;   // per row: left span start / length, right span start / length, running total
    int64_t total = 0;
};
void make_spans(int64_t ny, int64_t nx, int64_t cx, int64_t cy, double lo, double hi, double gsd, Spans& s) {
    const int64_t reach = (int64_t)floor(hi / gsd) + 1;
    const int64_t y0 = std::max<int64_t>(0, cy - reach), y1 = std::min<int64_t>(ny - 1, cy + reach);
    s.y0 = y0;
    const int64_t rows = y1 >= y0 ? y1 - y0 + 1 : 0;
    s.la.resize(rows); s.nl.resize(rows); s.ra.resize(rows); s.nr.resize(rows); s.cum.resize(rows);
    const double ho = (hi / gsd) * (hi / gsd), hi2 = (lo / gsd) * (lo / gsd);
    int64_t run = 0;
    for (int64_t r = 0; r < rows; ++r) {
        const double dy = (double)(y0 + r - cy), dy2 = dy * dy;
        const int64_t wo = (int64_t)floor(sqrt(std::max(ho - dy2, 0.0))) + 1;
        const int64_t wi = std::max<int64_t>((int64_t)ceil(sqrt(std::max(hi2 - dy2, 0.0))) - 1, 0);
        const int64_t la = std::max<int64_t>(cx - wo, 0), lb = std::min<int64_t>(cx - wi, nx - 1);
        const int64_t ra = std::max<int64_t>(cx + std::max<int64_t>(wi, 1), 0), rb = std::min<int64_t>(cx + wo, nx - 1);
        s.la[r] = la; s.nl[r] = std::max<int64_t>(lb - la + 1, 0);
        s.ra[r] = ra; s.nr[r] = std::max<int64_t>(rb - ra + 1, 0);
        run += s.nl[r] + s.nr[r];
        s.cum[r] = run;
    }
    s.total = run;
}

struct Draw {
    int64_t ny, nx, cx, cy;
    double lo, hi, gsd;
    const uint8_t* valid;   // [ny][nx] or null
    bool member(int64_t ix, int64_t iy) const {
        const double dx = (double)(ix - cx) * gsd, dyv = (double)(iy - cy) * gsd;
        const double d = sqrt(dx * dx + dyv * dyv);   // (the NumPy form's arithmetic: squares, sum, square root in float64)
        return d >= lo && d < hi && (!valid || valid[iy * nx + ix] != 0);
    }
};

// Scratch of one worker thread, kept between its jobs (fresh megabyte-sized vectors per job would go through mmap / munmap every time,
// which serialises the threads on the process's memory-map lock)
struct Work {
    Spans s;
    std::vector<int64_t> table;   // open-addressing set, -1 = free; emptied after use through `used`
    std::vector<size_t> used;
    std::vector<int64_t> all;     // members of a ring that is enumerated
};

// Up to `samples` distinct member pixels of the ring, uniformly without replacement, in random order; returns how many.
int64_t draw_ring(const Draw& D, int64_t samples, uint64_t seed, int64_t* out, Work& W) {
    Spans& s = W.s;
    make_spans(D.ny, D.nx, D.cx, D.cy, D.lo, D.hi, D.gsd, s);
    if (s.total == 0 || samples <= 0) return 0;
    Rng rng(seed);
    const int64_t rows = (int64_t)s.cum.size();
    auto pixel_of = [&](int64_t k, int64_t& ix, int64_t& iy) {
        const int64_t r = std::upper_bound(s.cum.begin(), s.cum.end(), k) - s.cum.begin();
        const int64_t o = k - (s.cum[r] - s.nl[r] - s.nr[r]);
        ix = o < s.nl[r] ? s.la[r] + o : s.ra[r] + (o - s.nl[r]);
        iy = s.y0 + r;
    };
    auto enumerate_all = [&]() -> int64_t {
        std::vector<int64_t>& all = W.all;
        all.clear();
        for (int64_t r = 0; r < rows; ++r) {
            const int64_t iy = s.y0 + r;
            for (int64_t o = 0; o < s.nl[r]; ++o) if (D.member(s.la[r] + o, iy)) all.push_back(iy * D.nx + s.la[r] + o);
            for (int64_t o = 0; o < s.nr[r]; ++o) if (D.member(s.ra[r] + o, iy)) all.push_back(iy * D.nx + s.ra[r] + o);
        }
        const int64_t n = (int64_t)all.size();
        if (n <= samples) {   // the whole ring, in raster order (as the NumPy form returns it)
            std::copy(all.begin(), all.end(), out);
            return n;
        }
        for (int64_t i = 0; i < samples; ++i) {   // partial Fisher-Yates: a uniform sample in random order
            const int64_t j = i + (int64_t)rng.below((uint64_t)(n - i));
            std::swap(all[(size_t)i], all[(size_t)j]);
            out[i] = all[(size_t)i];
        }
        return samples;
    };
    if (s.total <= std::max<int64_t>(1 << 16, 4 * samples)) return enumerate_all();
    // rejection over the spans with an open-addressing set of the accepted pixels: the first `samples` DISTINCT members of a
    // sequence of independent uniform draws are a uniform sample without replacement, in random order
    size_t cap = 1;
    while (cap < (size_t)(2 * samples + 16)) cap <<= 1;
    std::vector<int64_t>& table = W.table;
    if (table.size() != cap) table.assign(cap, -1);
    W.used.clear();
    struct Cleaner { Work& w; ~Cleaner() { for (size_t h : w.used) w.table[h] = -1; } } cleaner{W};
    int64_t have = 0, drawn = 0, accepted = 0;
    const int64_t look = std::max<int64_t>(4096, samples / 2);
    while (have < samples) {
        int64_t ix, iy;
        pixel_of((int64_t)rng.below((uint64_t)s.total), ix, iy);
        ++drawn;
        if (D.member(ix, iy)) {
            ++accepted;
            const int64_t p = iy * D.nx + ix;
            size_t h = (size_t)(((uint64_t)p * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
            while (table[h] != -1 && table[h] != p) h = (h + 1) & (cap - 1);
            if (table[h] == -1) { table[h] = p; W.used.push_back(h); out[have++] = p; }
        }
        // about as many (valid) ring pixels as wanted, or fewer: take them all (checked now and then)
        if (drawn % look == 0 && (double)accepted * ((double)s.total / (double)drawn) < 1.5 * (double)samples) return enumerate_all();
        if (drawn > 64 * (samples + 1024) && have < samples) return enumerate_all();
    }
    return samples;
}

// f(i, worker scratch) for i in [0, n) on `threads` threads (a shared counter hands the jobs out)
template <class W, class F> void parallel_for(int64_t n, int threads, F f) {
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n));
    if (threads == 1) { W w; for (int64_t i = 0; i < n; ++i) f(i, w); return; }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&]() { W w; for (int64_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i, w); });
    for (auto& th : pool) th.join();
}

struct NoScratch {};

inline uint32_t spread16(uint32_t v) {
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

}  // namespace

extern "C" {

// (declared in include/xdemhip.h)
int xdemhip_host_ring_sample(const uint8_t* valid, int64_t ny, int64_t nx, double gsd, int64_t runs, const int64_t* cx, const int64_t* cy,
                             int n_rings, const double* ring_lo, const double* ring_hi /* [n_rings] each */, int64_t samples, uint64_t seed,
                             int threads, int64_t* out_idx /* [runs][n_rings][samples] */, int64_t* out_count /* [runs][n_rings] */) {
    if (ny < 1 || nx < 1 || !(gsd > 0) || runs < 0 || n_rings < 1 || !ring_lo || !ring_hi || samples < 1 || !out_idx || !out_count ||
        (runs > 0 && (!cx || !cy)))
        return XDEMHIP_EINVAL;
    for (int k = 0; k < n_rings; ++k) if (!(ring_hi[k] >= ring_lo[k]) || !(ring_lo[k] >= 0)) return XDEMHIP_EINVAL;
    for (int64_t r = 0; r < runs; ++r) if (cx[r] < 0 || cx[r] >= nx || cy[r] < 0 || cy[r] >= ny) return XDEMHIP_EINVAL;
    // one job per (run, ring): the outer rings of a run cost 10x its inner ones, and a pool over the pairs balances that
    parallel_for<Work>(runs * n_rings, threads, [&](int64_t job, Work& W) {
        const int64_t run = job / n_rings;
        const int ring = (int)(job % n_rings);
        Draw D{ny, nx, cx[run], cy[run], ring_lo[ring], ring_hi[ring], gsd, valid};
        int64_t* o = out_idx + (run * n_rings + ring) * samples;
        const int64_t n = draw_ring(D, samples, mix3(seed, (uint64_t)run, (uint64_t)ring), o, W);
        out_count[run * n_rings + ring] = n;
        for (int64_t i = n; i < samples; ++i) o[i] = -1;
    });
    return XDEMHIP_OK;
}

// Values and coordinates of the sampled pixels of `n_blocks` point sets (flat pixel indexes idx[off[b] .. off[b + 1])) in the order
// given -- and, if sx_out is not null, once more with every set permuted into Morton order of its own bounding box (16 bits per axis:
// the order of xdem_amd/spatialstats.py: _morton_order).  dtype: XDEMHIP_F32 / XDEMHIP_F64 values.  The gather reads a raster of
// gigabytes at random: the loads are prefetched a few dozen elements ahead (one cache miss in flight per element otherwise).
int xdemhip_host_gather_points(const void* values, int dtype, int64_t nx, double gsd, int n_blocks, const int64_t* off, const int64_t* idx,
                               int threads, double* x_out, double* y_out, void* v_out, double* sx_out, double* sy_out, void* sv_out) {
    if (!values || (dtype != XDEMHIP_F32 && dtype != XDEMHIP_F64) || nx < 1 || n_blocks < 0 || !off || !idx || !x_out || !y_out || !v_out)
        return XDEMHIP_EINVAL;
    if (sx_out && (!sy_out || !sv_out)) return XDEMHIP_EINVAL;
    const size_t es = dtype == XDEMHIP_F32 ? 4 : 8;
    struct Scratch { std::vector<uint64_t> keyed; };
    parallel_for<Scratch>(n_blocks, threads, [&](int64_t b, Scratch& S) {
        const int64_t o = off[b], n = off[b + 1] - off[b];
        const int64_t* id = idx + o;
        constexpr int64_t AHEAD = 32;
        for (int64_t i = 0; i < n; ++i) {
            if (i + AHEAD < n) __builtin_prefetch(static_cast<const char*>(values) + (size_t)id[i + AHEAD] * es, 0, 0);
            const int64_t p = id[i];
            x_out[o + i] = (double)(p % nx) * gsd;
            y_out[o + i] = (double)(p / nx) * gsd;
            if (dtype == XDEMHIP_F32) static_cast<float*>(v_out)[o + i] = static_cast<const float*>(values)[p];
            else static_cast<double*>(v_out)[o + i] = static_cast<const double*>(values)[p];
        }
        if (!sx_out) return;
        std::vector<uint64_t>& keyed = S.keyed;
        keyed.resize((size_t)n);
        if (n >= 3) {
            int64_t x0 = INT64_MAX, x1 = INT64_MIN, y0 = INT64_MAX, y1 = INT64_MIN;
            for (int64_t i = 0; i < n; ++i) {
                const int64_t ix = id[i] % nx, iy = id[i] / nx;
                x0 = std::min(x0, ix); x1 = std::max(x1, ix); y0 = std::min(y0, iy); y1 = std::max(y1, iy);
            }
            const double sx = x1 > x0 ? 65535.0 / ((double)(x1 - x0) * gsd) : 0.0, sy = y1 > y0 ? 65535.0 / ((double)(y1 - y0) * gsd) : 0.0;
            for (int64_t i = 0; i < n; ++i) {
                const int64_t ix = id[i] % nx, iy = id[i] / nx;
                const uint32_t qx = (uint32_t)(((double)(ix - x0) * gsd) * sx), qy = (uint32_t)(((double)(iy - y0) * gsd) * sy);
                keyed[(size_t)i] = ((uint64_t)(spread16(qx) | (spread16(qy) << 1)) << 32) | (uint64_t)(uint32_t)i;   // (n < 2^32)
            }
            std::sort(keyed.begin(), keyed.end());
        } else {
            for (int64_t i = 0; i < n; ++i) keyed[(size_t)i] = (uint64_t)i;
        }
        for (int64_t i = 0; i < n; ++i) {
            const int64_t j = o + (int64_t)(keyed[(size_t)i] & 0xFFFFFFFFull);
            sx_out[o + i] = x_out[j];
            sy_out[o + i] = y_out[j];
            if (dtype == XDEMHIP_F32) static_cast<float*>(sv_out)[o + i] = static_cast<const float*>(v_out)[j];
            else static_cast<double*>(sv_out)[o + i] = static_cast<const double*>(v_out)[j];
        }
    });
    return XDEMHIP_OK;
}

// (declared in include/xdemhip.h)
int xdemhip_host_count_finite(const void* values, int dtype, int64_t n, int threads, int64_t* n_finite, uint8_t* valid_out) {
    if ((!values && n > 0) || (dtype != XDEMHIP_F32 && dtype != XDEMHIP_F64) || n < 0 || !n_finite) return XDEMHIP_EINVAL;
    constexpr int64_t PIECE = (int64_t)1 << 22;
    const int64_t pieces = (n + PIECE - 1) / PIECE;
    std::atomic<int64_t> total(0);
    parallel_for<NoScratch>(pieces, threads, [&](int64_t k, NoScratch&) {
        const int64_t a = k * PIECE, b = std::min(n, a + PIECE);
        int64_t c = 0;
        // (finite <=> the exponent bits are not all ones: integer tests, which the compiler vectorises)
        if (dtype == XDEMHIP_F32) {
            const uint32_t* v = static_cast<const uint32_t*>(values);
            if (valid_out) for (int64_t i = a; i < b; ++i) { const int f = (v[i] & 0x7F800000u) != 0x7F800000u; valid_out[i] = (uint8_t)f; c += f; }
            else for (int64_t i = a; i < b; ++i) c += (v[i] & 0x7F800000u) != 0x7F800000u;
        } else {
            const uint64_t* v = static_cast<const uint64_t*>(values);
            if (valid_out) for (int64_t i = a; i < b; ++i) { const int f = (v[i] & 0x7FF0000000000000ull) != 0x7FF0000000000000ull; valid_out[i] = (uint8_t)f; c += f; }
            else for (int64_t i = a; i < b; ++i) c += (v[i] & 0x7FF0000000000000ull) != 0x7FF0000000000000ull;
        }
        total.fetch_add(c);
    });
    *n_finite = total.load();
    return XDEMHIP_OK;
}

}  // extern "C"
