// common.h -- context object and helpers shared by the C-ABI translation units of libxdemhip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

#include "../../include/xdemhip.h"

struct xdemhip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // stream work is enqueued on (own_stream or the caller's)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_copy = nullptr;
    static constexpr int MAX_COPY_THREADS = 16;
    hipStream_t copy_streams[MAX_COPY_THREADS] = {};  // one per copy thread of the host-buffer path
    int host_copy_threads = 8;                        // option "host_copy_threads"
    void* stage_in[2] = {nullptr, nullptr};            // pinned staging of the host-buffer terrain path (double-buffered), kept between calls
    void* stage_out[2] = {nullptr, nullptr};
    size_t stage_in_bytes = 0, stage_out_bytes = 0;
    int pairs_launch_cap = 0;                         // option "pairs_launch_cap": workgroups per pair-kernel launch (0 = 2^31 / NT)
    bool timed = false;
    int num_cu = 256;
    xdemhip_allreduce_fn allreduce = nullptr;  // multi-GPU hook (null: single process)
    void* allreduce_user = nullptr;
    xdemhip_allreduce_device_fn allreduce_dev = nullptr;  // device-side form of the hook (device arrays stay on the device)
    void* allreduce_dev_user = nullptr;
    int64_t n_red_host = 0, n_red_dev = 0;     // reductions that went through the host / the device hook
    int rank = 0, world = 0;                   // xdemhip_set_rank: this process's place among the ranks the hooks reduce over (world 0: not told)
    int nk_fused_dist = 1;   // option "nk_fused_dist": 1 partitioned Nuth-Kaab plans (reduction hook + xdemhip_set_rank) take the one-pass step too: 5-10 all-reduces per step (default), 0 the plain route
    int host_chunk_rows = 0; // option "host_chunk_rows": rows per chunk of host-buffer terrain calls (0 = from the budget); the mp_config tile size
    int host_chunk_mb = 0;   // device budget (MiB) of one row chunk of host-buffer terrain calls; 0 = default
    int terrain_store = 0;   // option "terrain_store": 0 direct stores (default), 1 staged 1 KiB row stores where possible (measured slower)
    int terrain_rows = 0;    // option "terrain_rows": tile height of the fused terrain kernel (0 automatic, 16, 24, 32)
    int nk_narrow = -1;      // option "nk_narrow": sample brackets of the one-pass Nuth-Kaab step 2^-k as wide as the rule (-1 = adaptive: from the offsets measured in earlier steps)
    int vario_runs = 1;      // option "vario_runs": run-length counting pass of the bracketed Dowd selection when a Morton-ordered copy is linked (0 = per-pair counters)
    int vario_deff = 0;      // option "vario_deff": design effect assumed for the pair samples of the bracketed Dowd selection (0 = built-in rule)
    int vario_sort = 1;      // option "vario_sort": the Python side uploads the points of a pair block in Morton order (run-length accumulation of the pair kernels)
    int nk_predict = 1;      // option "nk_predict": 1 a settled one-pass Nuth-Kaab step takes its brackets from the previous step's exact medians moved by the model (no sample kernels, no digit passes over samples; default), 0 every step samples
    int nk_fused = 1;        // option "nk_fused": 1 the Nuth-Kaab step of large single-GPU plans is ONE data pass (14 B/pixel: dh, its median's counting and the aspect-bin counting against sample brackets with per-pixel margins; default), 0 the two passes of round 3
    int terrain_stream = 1;  // option "terrain_stream": 1 streaming strips for the raster interior where they apply (default), 0 tiles only; 128 / 256 / 512 = band height
    int terrain_sync = 0;    // option "terrain_sync": workgroup barrier every N output rows of the direct-store march (0 none; N a power of two)
    int terrain_order = 0;   // option "terrain_order": 0 one band of tiles per XCD (default), 1 natural order (XCDs interleave along a tile row); strips only: 2 permuted, 3 column-major (measurement)
    int terrain_window_lds = 1;  // option "terrain_window_lds": 1 LDS-tiled window kernel for window sizes != 3 (default), 0 the per-pixel form (its check)
    int terrain_ring_wait = 0;  // option "terrain_ring_wait": 1 = the streaming strips drain every VMEM operation before reading a refilled ring block (test switch for the counted wait)
    int terrain_nonfinite = 0;  // option "terrain_nonfinite": 0 an output is NaN iff its full window holds a non-finite value (SciPy engine, default), 1 the Numba engine's rule: +-Inf pixels go through the arithmetic (terrain_nonfinite.hip)
    int* nf_flag = nullptr;  // device word of that path: "the rows at hand hold an infinite pixel"
    int terrain_math = 2;    // option "terrain_math": float32 rasters: 2 lean tail for the specialised attribute sets (default), 0 mixed-precision tail of round 2, 1 float64 tail everywhere
    int vario_grid = 1;      // option "vario_grid": 1 = integer-lattice pair kernels for raster-sampled points (default), 0 = always float64 coordinates
    int vario_edge = 0;      // option "vario_edge": lag classes 0 = [e_{k-1}, e_k) (default), 1 = (e_{k-1}, e_k]
    int vario_diff = 0;      // option "vario_diff": |dv| formed 0 = in the value dtype (default), 1 = in float64 (float32 values are widened)
    int nk_nan_rule = 0;     // option "nk_nan_rule": nodata spreading of the bilinear taps (nuthkaab.hip): 0 4tap, 1 weighted, 2 dilate3x3, 3 dilate_cross
    int selection_mode = 0;  // 0 auto (bracketed for large inputs), 1 plain digit passes only, 2 degenerate brackets (tests the
                             // fallback), 3 bracketed whatever the per-bin sample size (2 and 3: test switches)
    // Deferred device-to-host results (xd_d2h / xd_sync below): small result blocks land in one pinned staging buffer with
    // truly asynchronous copies and are handed to their destinations at the next xd_sync -- a copy into pageable memory
    // would stall the host once per block (tens of microseconds each, a dozen blocks per Nuth-Kaab step).
    unsigned char* pin = nullptr;
    size_t pin_cap = 0, pin_used = 0;
    struct Pending { void* dst; size_t off, bytes; };
    std::vector<Pending> pending;
    std::string err;
};

#define XD_HIP_CHECK(ctx, expr)                                                                   \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            char _b[512];                                                                         \
            snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (ctx)->err = _b;                                                                      \
            return XDEMHIP_EHIP;                                                                  \
        }                                                                                         \
    } while (0)

inline int xd_fail(xdemhip_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// Queue a small device-to-host copy whose destination is filled by the next xd_sync(ctx) (falls back to a plain copy when
// the staging buffer is full).  `dst` must stay alive until that xd_sync; xd_drop_pending forgets undelivered blocks (entry
// points call it on their way in and out, so an error return never leaves a dangling destination behind).
inline void xd_drop_pending(xdemhip_ctx* ctx) {
    ctx->pending.clear();
    ctx->pin_used = 0;
}
inline int xd_d2h(xdemhip_ctx* ctx, void* dst, const void* dsrc, size_t bytes) {
    if (!ctx->pin) {
        if (hipHostMalloc(reinterpret_cast<void**>(&ctx->pin), 256 * 1024, hipHostMallocDefault) == hipSuccess) ctx->pin_cap = 256 * 1024;
        else ctx->pin = nullptr;
    }
    const size_t need = (bytes + 15) & ~(size_t)15;
    if (!ctx->pin || ctx->pin_used + need > ctx->pin_cap) {
        XD_HIP_CHECK(ctx, hipMemcpyAsync(dst, dsrc, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return XDEMHIP_OK;
    }
    if (hipMemcpyAsync(ctx->pin + ctx->pin_used, dsrc, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
        xd_drop_pending(ctx);  // the operation is abandoned: nothing queued before may be delivered into dead destinations later
        return xd_fail(ctx, XDEMHIP_EHIP, "hipMemcpyAsync (deferred device-to-host copy) failed");
    }
    ctx->pending.push_back({dst, ctx->pin_used, bytes});
    ctx->pin_used += need;
    return XDEMHIP_OK;
}
// The same for results a KERNEL writes itself: a region of the pinned staging buffer (device-writable: pinned host memory is mapped)
// that the next xd_sync hands to `dst` -- saves the copy engine's launch behind the kernel.  nullptr: no room, use xd_d2h.
inline unsigned char* xd_pin_claim(xdemhip_ctx* ctx, void* dst, size_t bytes) {
    if (!ctx->pin) {
        if (hipHostMalloc(reinterpret_cast<void**>(&ctx->pin), 256 * 1024, hipHostMallocDefault) == hipSuccess) ctx->pin_cap = 256 * 1024;
        else { ctx->pin = nullptr; (void)hipGetLastError(); }
    }
    const size_t need = (bytes + 15) & ~(size_t)15;
    if (!ctx->pin || ctx->pin_used + need > ctx->pin_cap) return nullptr;
    unsigned char* at = ctx->pin + ctx->pin_used;
    ctx->pending.push_back({dst, ctx->pin_used, bytes});
    ctx->pin_used += need;
    return at;
}
// Pages of a caller's HOST output buffer made resident before a large device-to-host copy lands in them: a fresh np.empty is
// untouched virtual memory, and a copy that faults its pages in one by one runs at ~10 GB/s instead of the link's ~55.  `threads`
// threads write every page's first byte back to itself (contents kept); returns the threads for the caller to join -- the faults
// overlap whatever the device is still doing for this call.
struct XdPrefault {
    std::vector<std::thread> workers;
    void join() { for (auto& w : workers) if (w.joinable()) w.join(); workers.clear(); }
    ~XdPrefault() { join(); }
};
inline void xd_prefault_start(XdPrefault& pf, void* dst, size_t bytes, int threads) {
    if (!dst || bytes < ((size_t)64 << 20)) return;
    threads = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
    const size_t per = (bytes + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; ++t)
        pf.workers.emplace_back([=]() {
            volatile unsigned char* p = static_cast<volatile unsigned char*>(dst);
            const size_t lo = per * (size_t)t, hi = lo + per < bytes ? lo + per : bytes;
            for (size_t o = lo; o < hi; o += 4096) p[o] = p[o];
        });
}
inline int xd_sync(xdemhip_ctx* ctx) {
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        xd_drop_pending(ctx);
        return xd_fail(ctx, XDEMHIP_EHIP, std::string("hipStreamSynchronize failed: ") + hipGetErrorString(e));
    }
    for (const auto& p : ctx->pending) memcpy(p.dst, ctx->pin + p.off, p.bytes);
    xd_drop_pending(ctx);
    return XDEMHIP_OK;
}
struct XdFetchScope {  // RAII for C entry points that queue deferred copies
    xdemhip_ctx* c;
    explicit XdFetchScope(xdemhip_ctx* ctx) : c(ctx) { if (c) xd_drop_pending(c); }
    ~XdFetchScope() { if (c) xd_drop_pending(c); }
};

// All-reduce a small device array over the ranks through the caller's hook (no-op without a hook): the array is
// staged to the host, combined by the hook (torch.distributed / RCCL or gloo on the Python side) and copied back.
inline int xd_allreduce_device(xdemhip_ctx* ctx, void* dptr, int64_t count, int kind) {
    if (!ctx->allreduce || count <= 0) return XDEMHIP_OK;
    if (ctx->allreduce_dev) {  // enqueued on the context's stream by the caller's hook: no staging, no host synchronisation
        ++ctx->n_red_dev;
        if (ctx->allreduce_dev(dptr, count, kind, static_cast<void*>(ctx->stream), ctx->allreduce_dev_user) != 0)
            return xd_fail(ctx, XDEMHIP_EHIP, "device all-reduce hook failed");
        return XDEMHIP_OK;
    }
    ++ctx->n_red_host;
    std::string buf((size_t)count * 8, '\0');
    hipError_t e = hipMemcpyAsync(&buf[0], dptr, buf.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return xd_fail(ctx, XDEMHIP_EHIP, "all-reduce staging (D2H) failed");
    if (ctx->allreduce(&buf[0], count, kind, ctx->allreduce_user) != 0) return xd_fail(ctx, XDEMHIP_EHIP, "all-reduce hook failed");
    e = hipMemcpyAsync(dptr, &buf[0], buf.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return xd_fail(ctx, XDEMHIP_EHIP, "all-reduce staging (H2D) failed");
    return XDEMHIP_OK;
}

// A small HOST array through the host hook: the route agreements, made once per plan (like the per-step agreements of the selection
// routes they are not counted by xdemhip_reduction_calls, which counts the data reductions of the steps).
inline int xd_allreduce_host(xdemhip_ctx* ctx, void* hptr, int64_t count, int kind) {
    if (!ctx->allreduce || count <= 0) return XDEMHIP_OK;
    if (ctx->allreduce(hptr, count, kind, ctx->allreduce_user) != 0) return xd_fail(ctx, XDEMHIP_EHIP, "all-reduce hook failed");
    return XDEMHIP_OK;
}

// Launchers implemented in the kernel translation units.
namespace xd {
struct TerrainLaunch {
    const void* dem;     // device pointer to buffer row 0
    int dem_dtype, out_dtype;
    int64_t H, W, row_stride, halo_top, halo_bottom;
    double resolution;
    int surface_fit, curv_method, tri_method, window_size, degrees;
    int hs_unclipped;    // bit 1 of xdemhip_terrain's `degrees`: hillshade as the reference's ENGINE returns it (no clip to [0, 255])
    uint32_t attr_mask;
    double hs_alt, hs_az, hs_z;
    void* planes[XDEMHIP_ATTR_COUNT];  // device pointers by attribute bit (null when not requested)
};
int launch_terrain(xdemhip_ctx* ctx, const TerrainLaunch& L);
int launch_terrain_nonfinite(xdemhip_ctx* ctx, const TerrainLaunch& L);  // terrain_nonfinite.hip
// rugosity / fractal roughness (window_extra.hip); plane indexes of the two attributes
constexpr int P_RUGOSITY_IDX = 13, P_FRACTAL_IDX = 14;
int launch_window_extra(xdemhip_ctx* ctx, const TerrainLaunch& L);
int fractal_constants(int w, int* qs, double* x, double* m_x, double* ss_xx);
}  // namespace xd
