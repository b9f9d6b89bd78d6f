// capi.hip -- extern "C" entry points of libxdemhip.so (declared in include/xdemhip.h): context
// management, argument validation, host<->device staging.  Kernels live in terrain.hip / nuthkaab.hip /
// variogram.hip.
#include <math.h>
#include <string.h>

#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

#define XDEMHIP_VERSION_NUM 100

extern "C" {

int xdemhip_version(void) { return XDEMHIP_VERSION_NUM; }

int xdemhip_create(int device_id, xdemhip_ctx** out_ctx) {
    if (!out_ctx) return XDEMHIP_EINVAL;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return XDEMHIP_ENODEV;
    if (device_id < 0 || device_id >= n) return XDEMHIP_ENODEV;
    xdemhip_ctx* c = new xdemhip_ctx();
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) { delete c; return XDEMHIP_EHIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return XDEMHIP_EHIP; }
    c->stream = c->own_stream;
    for (int t = 0; t < xdemhip_ctx::MAX_COPY_THREADS; ++t)
        if (hipStreamCreateWithFlags(&c->copy_streams[t], hipStreamNonBlocking) != hipSuccess) { delete c; return XDEMHIP_EHIP; }
    if (hipEventCreate(&c->ev_start) != hipSuccess || hipEventCreate(&c->ev_stop) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return XDEMHIP_EHIP;
    }
    *out_ctx = c;
    return XDEMHIP_OK;
}

void xdemhip_destroy(xdemhip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->ev_copy) (void)hipEventDestroy(ctx->ev_copy);
    for (int t = 0; t < xdemhip_ctx::MAX_COPY_THREADS; ++t)
        if (ctx->copy_streams[t]) (void)hipStreamDestroy(ctx->copy_streams[t]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->nf_flag) (void)hipFree(ctx->nf_flag);
    for (int q = 0; q < 2; ++q) {
        if (ctx->stage_in[q]) (void)hipHostFree(ctx->stage_in[q]);
        if (ctx->stage_out[q]) (void)hipHostFree(ctx->stage_out[q]);
    }
    delete ctx;
}

const char* xdemhip_last_error(const xdemhip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int xdemhip_set_stream(xdemhip_ctx* ctx, void* hip_stream) {
    if (!ctx) return XDEMHIP_EINVAL;
    // NULL is a stream too: HIP's default stream, which is what torch.cuda.current_stream().cuda_stream reports (0) unless the
    // caller switched streams -- work must be ordered with the caller's kernels on it, not parked on a private stream
    ctx->stream = (hip_stream == XDEMHIP_OWN_STREAM) ? ctx->own_stream : static_cast<hipStream_t>(hip_stream);
    return XDEMHIP_OK;
}

int xdemhip_set_allreduce(xdemhip_ctx* ctx, xdemhip_allreduce_fn fn, void* user) {
    if (!ctx) return XDEMHIP_EINVAL;
    ctx->allreduce = fn;
    ctx->allreduce_user = user;
    return XDEMHIP_OK;
}

int xdemhip_set_allreduce_device(xdemhip_ctx* ctx, xdemhip_allreduce_device_fn fn, void* user) {
    if (!ctx) return XDEMHIP_EINVAL;
    ctx->allreduce_dev = fn;
    ctx->allreduce_dev_user = user;
    return XDEMHIP_OK;
}

int xdemhip_set_rank(xdemhip_ctx* ctx, int rank, int world) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (world < 0 || (world > 0 && (rank < 0 || rank >= world))) return xd_fail(ctx, XDEMHIP_EINVAL, "rank must lie in [0, world)");
    ctx->rank = world > 0 ? rank : 0;
    ctx->world = world;
    return XDEMHIP_OK;
}

int xdemhip_reduction_calls(xdemhip_ctx* ctx, int64_t* host_calls, int64_t* device_calls) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (host_calls) *host_calls = ctx->n_red_host;
    if (device_calls) *device_calls = ctx->n_red_dev;
    return XDEMHIP_OK;
}

namespace {
int device_alloc_once(xdemhip_ctx* ctx, size_t bytes, bool contiguous, void** ptr, int* got_contiguous) {
    *ptr = nullptr;
    if (got_contiguous) *got_contiguous = 0;
    if (contiguous) {
        if (hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocContiguous) == hipSuccess && *ptr) {
            if (got_contiguous) *got_contiguous = 1;
            return XDEMHIP_OK;
        }
        (void)hipGetLastError();   // no single piece of that size: an ordinary allocation will do
        *ptr = nullptr;
    }
    if (hipMalloc(ptr, bytes) != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc failed");
    }
    return XDEMHIP_OK;
}
}  // namespace

namespace {
// XDEMHIP_ALLOC_CHUNKED / _SCATTERED: one virtual range backed by separately created physical pieces (HIP virtual memory
// management); SCATTERED maps the pieces into the range in a pseudo-random order, so that neighbouring pieces of the range are not
// neighbours in device memory even when the driver hands the pieces out of one contiguous block
struct ChunkedAlloc { void* base = nullptr; size_t size = 0, piece = 0; std::vector<hipMemGenericAllocationHandle_t> pieces; std::vector<size_t> slot; };
std::mutex g_chunk_mu;
std::vector<ChunkedAlloc> g_chunked;

// Virtual ranges are NOT handed back for reuse when their pieces go: on this stack (ROCm 7.2, gfx950) a range that is freed, reserved
// again at the same address and mapped onto other physical pieces can be read and written through translations of its previous
// mapping -- planes of a 46400^2 raster came back with tens of MiB of other memory's contents in 6 of 14 repeats, and in none of 14 with
// the addresses kept (tools/vmm_stale_probe.py, profiles/r04_vmm_stale_probe.txt).  Freed ranges stay reserved (address space only, no
// memory) in a first-in first-out quarantine and are returned to the driver once more than XDEMHIP_VMM_QUARANTINE_TB (default 32 TiB
// of the 128 TiB address space) wait there -- by then thousands of other ranges have been mapped and used since.
struct QuarantinedRange { void* base; size_t size; };
std::mutex g_quarantine_mu;
std::vector<QuarantinedRange> g_quarantine;
size_t g_quarantine_bytes = 0;

void quarantine_range(void* base, size_t size) {
    // (scaled in double BEFORE the cast: a fractional value such as 0.5 must not truncate to 0 and switch the quarantine off)
    static const size_t limit = (size_t)((getenv("XDEMHIP_VMM_QUARANTINE_TB") ? atof(getenv("XDEMHIP_VMM_QUARANTINE_TB")) : 32.0) * (double)((size_t)1 << 40));
    std::lock_guard<std::mutex> lock(g_quarantine_mu);
    g_quarantine.push_back({base, size});
    g_quarantine_bytes += size;
    size_t n_free = 0;
    while (g_quarantine_bytes > limit && n_free < g_quarantine.size()) {
        (void)hipMemAddressFree(g_quarantine[n_free].base, g_quarantine[n_free].size);
        g_quarantine_bytes -= g_quarantine[n_free].size;
        ++n_free;
    }
    if (n_free) g_quarantine.erase(g_quarantine.begin(), g_quarantine.begin() + (long)n_free);
}

void chunked_release(ChunkedAlloc& c, size_t mapped_pieces) {
    for (size_t i = 0; i < c.pieces.size(); ++i) {
        if (i < mapped_pieces) (void)hipMemUnmap(static_cast<char*>(c.base) + c.slot[i] * c.piece, c.piece);
        (void)hipMemRelease(c.pieces[i]);
    }
    if (c.base) quarantine_range(c.base, c.size);
}

int device_alloc_chunked(xdemhip_ctx* ctx, size_t bytes, size_t piece, bool shuffle, void** ptr) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) {
        (void)hipGetLastError();
        return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "virtual memory management is not available");
    }
    ChunkedAlloc c;
    c.piece = ((piece + gran - 1) / gran) * gran;
    const size_t n = (bytes + c.piece - 1) / c.piece;
    c.size = n * c.piece;
    if (hipMemAddressReserve(&c.base, c.size, 0, nullptr, 0) != hipSuccess || !c.base) {
        (void)hipGetLastError();
        // out of address space with ranges waiting in the quarantine: hand the OLDEST back, one at a time, until the reserve
        // succeeds -- the most recently freed ranges (the ones whose stale translations were measured) stay quarantined
        c.base = nullptr;
        std::lock_guard<std::mutex> lock(g_quarantine_mu);
        while (!c.base && !g_quarantine.empty()) {
            (void)hipMemAddressFree(g_quarantine.front().base, g_quarantine.front().size);
            g_quarantine_bytes -= g_quarantine.front().size;
            g_quarantine.erase(g_quarantine.begin());
            if (hipMemAddressReserve(&c.base, c.size, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); c.base = nullptr; }
        }
        if (!c.base) return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMemAddressReserve failed");
    }
    c.slot.resize(n);
    for (size_t i = 0; i < n; ++i) c.slot[i] = i;
    if (shuffle) {   // Fisher-Yates with a fixed linear congruential generator: the same layout every time
        uint64_t st = 0x9E3779B97F4A7C15ull;
        for (size_t i = n; i > 1; --i) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const size_t k = (size_t)((st >> 33) % i);
            std::swap(c.slot[i - 1], c.slot[k]);
        }
    }
    size_t mapped = 0;
    bool ok = true;
    for (size_t i = 0; i < n && ok; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, c.piece, &prop, 0) != hipSuccess) { ok = false; break; }
        c.pieces.push_back(h);
        if (hipMemMap(static_cast<char*>(c.base) + c.slot[i] * c.piece, c.piece, 0, h, 0) != hipSuccess) { ok = false; break; }
        ++mapped;
    }
    if (ok) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = hipMemSetAccess(c.base, c.size, &acc, 1) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        chunked_release(c, mapped);
        return xd_fail(ctx, XDEMHIP_ENOMEM, "chunked device allocation failed");
    }
    *ptr = c.base;
    std::lock_guard<std::mutex> lock(g_chunk_mu);
    g_chunked.push_back(std::move(c));
    return XDEMHIP_OK;
}

bool device_free_chunked(void* p) {   // true: `p` was a chunked allocation (now released)
    ChunkedAlloc c;
    {
        std::lock_guard<std::mutex> lock(g_chunk_mu);
        size_t i = 0;
        for (; i < g_chunked.size(); ++i)
            if (g_chunked[i].base == p) break;
        if (i == g_chunked.size()) return false;
        c = std::move(g_chunked[i]);
        g_chunked.erase(g_chunked.begin() + (long)i);
    }
    (void)hipDeviceSynchronize();
    chunked_release(c, c.pieces.size());
    return true;
}
}  // namespace

int xdemhip_device_alloc(xdemhip_ctx* ctx, size_t bytes, int flags, void** ptr, int* got_contiguous) {
    if (!ctx || !ptr || bytes == 0) return ctx ? xd_fail(ctx, XDEMHIP_EINVAL, "bad argument") : XDEMHIP_EINVAL;
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (flags & (XDEMHIP_ALLOC_CHUNKED | XDEMHIP_ALLOC_SCATTERED)) {
        *ptr = nullptr;
        if (got_contiguous) *got_contiguous = 0;
        const bool scattered = (flags & XDEMHIP_ALLOC_SCATTERED) != 0;
        // (round 6: 32 MiB pieces.  On boxes whose ordinary allocations are FAST placements larger pieces are faster -- fewer TLB misses:
        //  13.05 / 12.93 / 12.80 ms for 8 / 32 / 128 MiB against 12.87 on torch.empty planes --, on a box whose ordinary allocations are
        //  slow ones (14.7 ms) 8 / 32 / 128 MiB ran 13.0 / 12.8-13.0 / 12.8-13.2: profiles/r06_piece_probe.txt)
        size_t piece_mb = scattered ? 32 : 64;
        if (const char* e = getenv("XDEMHIP_SCATTER_PIECE_MB")) {   // (measurements: tools/piece_probe.py)
            const long v = atol(e);
            if (scattered && v >= 1 && v <= 4096) piece_mb = (size_t)v;
        }
        return device_alloc_chunked(ctx, bytes, piece_mb << 20, scattered, ptr);
    }
    const bool contiguous = (flags & XDEMHIP_ALLOC_CONTIGUOUS) != 0;
    if (flags & XDEMHIP_ALLOC_RECYCLED) {
        // a first allocation of this size is mapped, touched and released, and the request is served again: the second allocation
        // reuses the virtual range whose page tables now exist (see include/xdemhip.h)
        void* first = nullptr;
        int rc = device_alloc_once(ctx, bytes, contiguous, &first, nullptr);
        if (rc) return rc;
        const size_t touch = bytes < ((size_t)64 << 20) ? bytes : ((size_t)64 << 20);
        (void)hipMemsetAsync(first, 0, touch, ctx->stream);
        (void)hipMemsetAsync(static_cast<char*>(first) + (bytes - touch), 0, touch, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        XD_HIP_CHECK(ctx, hipFree(first));
    }
    return device_alloc_once(ctx, bytes, contiguous, ptr, got_contiguous);
}

int xdemhip_device_free(xdemhip_ctx* ctx, void* ptr) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (!ptr) return XDEMHIP_OK;
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (device_free_chunked(ptr)) return XDEMHIP_OK;
    XD_HIP_CHECK(ctx, hipFree(ptr));
    return XDEMHIP_OK;
}

int xdemhip_synchronize(xdemhip_ctx* ctx) {
    if (!ctx) return XDEMHIP_EINVAL;
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return XDEMHIP_OK;
}

// Options of the PRODUCT (include/xdemhip.h): what a caller of the C-ABI may want to set.  Everything that only selects between
// internal routes that must agree -- for the route-agreement tests and for measurements -- is a TEST SWITCH
// (xdemhip_set_test_switch, include/xdemhip_test.h); switches of measurement builds exist under -DXD_EXPERIMENT only.
namespace {
struct XdOpt { const char* name; int lo, hi; int xdemhip_ctx::*field; const char* what; };
const XdOpt kOptions[] = {
    {"host_chunk_mb", 0, 1 << 30, &xdemhip_ctx::host_chunk_mb, "host_chunk_mb must be >= 0"},
    {"host_chunk_rows", 0, 1 << 30, &xdemhip_ctx::host_chunk_rows, "host_chunk_rows must be >= 0"},
    {"pairs_launch_cap", 0, 1 << 30, &xdemhip_ctx::pairs_launch_cap, "pairs_launch_cap must be >= 0"},
    {"terrain_nonfinite", 0, 1, &xdemhip_ctx::terrain_nonfinite, "terrain_nonfinite: 0 window rule (SciPy engine), 1 arithmetic rule (Numba engine)"},
    {"terrain_math", 0, 2, &xdemhip_ctx::terrain_math, "terrain_math: 0 mixed precision, 1 float64, 2 lean"},
    {"vario_edge", 0, 1, &xdemhip_ctx::vario_edge, "vario_edge: 0 or 1"},
    {"vario_diff", 0, 1, &xdemhip_ctx::vario_diff, "vario_diff: 0 or 1"},
    {"nk_nan_rule", 0, 3, &xdemhip_ctx::nk_nan_rule, "nk_nan_rule: 0 4tap, 1 weighted, 2 dilate3x3, 3 dilate_cross"},
    {"selection", 0, 3, &xdemhip_ctx::selection_mode, "selection: 0 auto, 1 plain, 2 degenerate brackets, 3 bracketed"},
    {"nk_fused", 0, 1, &xdemhip_ctx::nk_fused, "nk_fused: 0 or 1"},
    {"nk_fused_dist", 0, 1, &xdemhip_ctx::nk_fused_dist, "nk_fused_dist: 0 or 1"},
    {"nk_predict", 0, 1, &xdemhip_ctx::nk_predict, "nk_predict: 0 or 1"},
};
const XdOpt kTestSwitches[] = {
    {"terrain_stream", 0, 512, &xdemhip_ctx::terrain_stream, "terrain_stream: 0, 1, 2, 3, 128, 256 or 512"},
    {"terrain_order", 0, 3, &xdemhip_ctx::terrain_order, "terrain_order: 0 XCD bands, 1 natural order, 2 permuted strips, 3 column-major strips"},
    {"terrain_ring_wait", 0, 1, &xdemhip_ctx::terrain_ring_wait, "terrain_ring_wait: 0 counted wait, 1 vmcnt(0)"},
    {"terrain_window_lds", 0, 1, &xdemhip_ctx::terrain_window_lds, "terrain_window_lds: 0 or 1"},
    {"nk_narrow", -1, 2, &xdemhip_ctx::nk_narrow, "nk_narrow: -1 (adaptive), 0, 1 or 2"},
    {"vario_grid", 0, 1, &xdemhip_ctx::vario_grid, "vario_grid: 0 or 1"},
    {"vario_runs", 0, 1, &xdemhip_ctx::vario_runs, "vario_runs: 0 or 1"},
    {"vario_sort", 0, 1, &xdemhip_ctx::vario_sort, "vario_sort: 0 or 1"},
#ifdef XD_EXPERIMENT   // measurement builds (csrc/Makefile: libxdemhip_exp*.so) only
    {"terrain_store", 0, 1, &xdemhip_ctx::terrain_store, "terrain_store: 0 direct, 1 staged row stores"},
    {"terrain_rows", 0, 1000, &xdemhip_ctx::terrain_rows, "terrain_rows: 0, 16, 24, 32 or an occupancy experiment >= 100"},
    {"terrain_sync", 0, 32, &xdemhip_ctx::terrain_sync, "terrain_sync: 0 or a power of two <= 32"},
    {"vario_deff", 0, 1 << 20, &xdemhip_ctx::vario_deff, "vario_deff: 0 .. 2^20"},
#endif
};
int xd_set_from(xdemhip_ctx* ctx, const XdOpt* tab, size_t n, const char* name, int value, bool* found) {
    *found = false;
    for (size_t k = 0; k < n; ++k)
        if (const XdOpt& o = tab[k]; strcmp(o.name, name) == 0) {
            *found = true;
            if (value < o.lo || value > o.hi) return xd_fail(ctx, XDEMHIP_EINVAL, o.what);
            ctx->*(o.field) = value;
            return XDEMHIP_OK;
        }
    return XDEMHIP_OK;
}
}  // namespace

int xdemhip_set_option(xdemhip_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return XDEMHIP_EINVAL;
    if (strcmp(name, "host_release") == 0) {  // free the pinned staging buffers of the host-buffer terrain path (they come back with the next call)
        for (int q = 0; q < 2; ++q) {
            if (ctx->stage_in[q]) (void)hipHostFree(ctx->stage_in[q]);
            if (ctx->stage_out[q]) (void)hipHostFree(ctx->stage_out[q]);
            ctx->stage_in[q] = ctx->stage_out[q] = nullptr;
        }
        ctx->stage_in_bytes = ctx->stage_out_bytes = 0;
        return XDEMHIP_OK;
    }
    if (strcmp(name, "host_copy_threads") == 0) {  // threads (one stream each) moving host-buffer rasters over PCIe (0 = default 8)
        if (value < 0 || value > xdemhip_ctx::MAX_COPY_THREADS) return xd_fail(ctx, XDEMHIP_EINVAL, "host_copy_threads must be 0..16");
        ctx->host_copy_threads = value ? value : 8;
        return XDEMHIP_OK;
    }
    bool found = false;
    const int rc = xd_set_from(ctx, kOptions, sizeof kOptions / sizeof kOptions[0], name, value, &found);
    if (found) return rc;
    for (const XdOpt& o : kTestSwitches)
        if (strcmp(o.name, name) == 0) return xd_fail(ctx, XDEMHIP_EINVAL, std::string(name) + " is a test switch, not an option: xdemhip_set_test_switch (include/xdemhip_test.h)");
    return xd_fail(ctx, XDEMHIP_EINVAL, std::string("unknown option: ") + name);
}

int xdemhip_set_test_switch(xdemhip_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return XDEMHIP_EINVAL;
    if (strcmp(name, "terrain_stream") == 0 && !(value == 0 || value == 1 || value == 2 || value == 3 || value == 128 || value == 256 || value == 512))
        return xd_fail(ctx, XDEMHIP_EINVAL, "terrain_stream: 0, 1, 2, 3, 128, 256 or 512");
    if (strcmp(name, "terrain_sync") == 0 && (value & (value - 1))) return xd_fail(ctx, XDEMHIP_EINVAL, "terrain_sync: 0 or a power of two <= 32");
    bool found = false;
    const int rc = xd_set_from(ctx, kTestSwitches, sizeof kTestSwitches / sizeof kTestSwitches[0], name, value, &found);
    if (found) return rc;
    return xd_fail(ctx, XDEMHIP_EINVAL, std::string("unknown test switch: ") + name);
}

int xdemhip_last_kernel_ms(xdemhip_ctx* ctx, float* ms) {
    if (!ctx || !ms) return XDEMHIP_EINVAL;
    if (!ctx->timed) return xd_fail(ctx, XDEMHIP_EINVAL, "no timed launch on this context yet");
    XD_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_stop));
    XD_HIP_CHECK(ctx, hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
    return XDEMHIP_OK;
}

// One wave that sleeps `sleeps` x 64 x 127 shader clocks and reads both device counters around it: s_memtime (clock64: counts at the
// SHADER clock on gfx950 -- 24.0 ticks per tick of the other one on an idle 2.4 GHz part, tools/clock_probe.hip) and s_memrealtime
// (wall_clock64: constant 100 MHz).  Launched on a side stream next to a heavy kernel it tells that kernel's effective clock.
static __global__ void xd_clock_probe_kernel(uint64_t* out, int sleeps) {
    const uint64_t t0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    const uint64_t t1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = t1 - t0; }
}

int xdemhip_clock_probe(xdemhip_ctx* ctx, void* hip_stream, int sleeps, uint64_t* out_device) {
    if (!ctx || !out_device || sleeps < 1) return XDEMHIP_EINVAL;
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(xd_clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(hip_stream), out_device, sleeps);
    XD_HIP_CHECK(ctx, hipGetLastError());
    return XDEMHIP_OK;
}

int xdemhip_fractal_constants(int window_size, int max_q, int* q, double* log_q, double* mean_log_q, double* ss_xx) {
    if (window_size < 3 || (window_size & 1) == 0 || window_size > 1023 || !q || !log_q || !mean_log_q || !ss_xx)
        return XDEMHIP_EINVAL;
    int qs[24];
    double x[24];
    const int n = xd::fractal_constants(window_size, qs, x, mean_log_q, ss_xx);
    if (n > max_q) return XDEMHIP_EINVAL;
    for (int i = 0; i < n; ++i) { q[i] = qs[i]; log_q[i] = x[i]; }
    return n;
}

static int popcount32(uint32_t v) { return __builtin_popcount(v); }

int xdemhip_terrain(xdemhip_ctx* ctx, const void* dem, int dem_dtype, int64_t H, int64_t W, int64_t row_stride,
                    int64_t halo_top, int64_t halo_bottom, double resolution, int surface_fit, int curv_method,
                    uint32_t attr_mask, int tri_method, int window_size, double hs_alt, double hs_az, double hs_z,
                    int degrees, int out_dtype, void* const* out_planes, int memspace) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (!dem || !out_planes) return xd_fail(ctx, XDEMHIP_EINVAL, "null buffer");
    if (H <= 0 || W <= 0 || row_stride < W || halo_top < 0 || halo_bottom < 0)
        return xd_fail(ctx, XDEMHIP_EINVAL, "bad raster geometry");
    if ((dem_dtype != XDEMHIP_F32 && dem_dtype != XDEMHIP_F64) || (out_dtype != XDEMHIP_F32 && out_dtype != XDEMHIP_F64))
        return xd_fail(ctx, XDEMHIP_EINVAL, "dtype must be XDEMHIP_F32 or XDEMHIP_F64");
    if (surface_fit < 0 || surface_fit > 2) return xd_fail(ctx, XDEMHIP_EINVAL, "unknown surface_fit");
    if (curv_method < 0 || curv_method > 1) return xd_fail(ctx, XDEMHIP_EINVAL, "unknown curv_method");
    if (tri_method < 0 || tri_method > 1) return xd_fail(ctx, XDEMHIP_EINVAL, "unknown tri_method");
    if (attr_mask == 0 || (attr_mask >> XDEMHIP_ATTR_COUNT)) return xd_fail(ctx, XDEMHIP_EINVAL, "bad attr_mask");
    const uint32_t curv_bits = attr_mask & 0x3f8u;
    if (surface_fit == XDEMHIP_FIT_HORN && curv_bits)
        return xd_fail(ctx, XDEMHIP_EINVAL, "'Horn' surface fit cannot be used to calculate curvatures");
    if ((attr_mask & 0x5c00u) && (window_size < 3 || (window_size & 1) == 0 || window_size > 1023))
        return xd_fail(ctx, XDEMHIP_EINVAL, "window_size must be odd and >= 3");
    const uint32_t needs_res = 0x3ffu | XDEMHIP_ATTR_RUGOSITY;  // surface fit + rugosity (terrain.py:352-367)
    if ((attr_mask & needs_res) && !(resolution > 0.0) ) return xd_fail(ctx, XDEMHIP_EINVAL, "resolution must be > 0");
    if (memspace != XDEMHIP_HOST && memspace != XDEMHIP_DEVICE) return xd_fail(ctx, XDEMHIP_EINVAL, "bad memspace");

    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int n_planes = popcount32(attr_mask);
    for (int i = 0; i < n_planes; ++i)
        if (!out_planes[i]) return xd_fail(ctx, XDEMHIP_EINVAL, "null output plane");

    xd::TerrainLaunch L;
    memset(&L, 0, sizeof L);
    L.dem_dtype = dem_dtype; L.out_dtype = out_dtype;
    L.H = H; L.W = W; L.row_stride = row_stride; L.halo_top = halo_top; L.halo_bottom = halo_bottom;
    L.resolution = (attr_mask & needs_res) ? resolution : 1.0;
    L.surface_fit = surface_fit; L.curv_method = curv_method; L.tri_method = tri_method;
    L.window_size = window_size; L.degrees = degrees & 1; L.hs_unclipped = (degrees >> 1) & 1; L.attr_mask = attr_mask;
    L.hs_alt = hs_alt; L.hs_az = hs_az; L.hs_z = hs_z;

    const size_t in_es = dem_dtype == XDEMHIP_F32 ? 4 : 8, out_es = out_dtype == XDEMHIP_F32 ? 4 : 8;

    if (memspace == XDEMHIP_DEVICE) {
        L.dem = dem;
        for (int bit = 0, i = 0; bit < XDEMHIP_ATTR_COUNT; ++bit)
            if (attr_mask & (1u << bit)) L.planes[bit] = out_planes[i++];
        XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
        int rc = xd::launch_terrain(ctx, L);
        XD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
        ctx->timed = (rc == XDEMHIP_OK);
        return rc;
    }

    // Host buffers: the raster streams through the device in ROW CHUNKS with the overlap the requested attributes need (the GPU
    // analogue of the reference's map_overlap tiles, xdem/terrain/terrain.py:412-466): device memory stays bounded however
    // large the host raster is, and a chunk is computed exactly as the same rows of the whole raster (halo_top / halo_bottom
    // of the kernels).  Round 3: a double-buffered pipeline over PINNED staging buffers owned by the context --
    //   copy threads: caller's rows -> pinned in[k & 1]      |  H2D (copy stream 0)  |  kernels (context stream)
    //   D2H (copy stream 1) -> pinned out[k & 1]             |  copy threads: pinned out -> caller's planes
    // with chunk k's transfers and kernels running while the threads move chunk k - 1's planes into the caller's (pageable,
    // often never-touched) arrays: the PCIe link stays busy in both directions instead of idling behind a pageable copy.
    int depth = 0;  // overlap rows per side
    if (attr_mask & 0x3ffu) depth = surface_fit == XDEMHIP_FIT_FLORINSKY ? 2 : 1;
    if ((attr_mask & 0x5c00u) && window_size / 2 > depth) depth = window_size / 2;
    if ((attr_mask & XDEMHIP_ATTR_RUGOSITY) && depth < 1) depth = 1;
    // chunk size: ~256 MiB of output planes per chunk by default (deep enough to hide latencies, small enough to pipeline),
    // capped by option "host_chunk_mb" (the device / pinned budget of one chunk, input + planes)
    const size_t row_bytes = (size_t)W * (in_es + out_es * (size_t)n_planes);
    const size_t budget = (size_t)(ctx->host_chunk_mb > 0 ? ctx->host_chunk_mb : 288) << 20;
    int64_t chunk = (int64_t)(budget / (row_bytes ? row_bytes : 1)) - 2 * depth;
    // the caller's tile size (mp_config.chunk_size of the reference's tiled call) can only SHRINK the chunks: a tile of the whole raster's
    // height -- or a moderate one on a very wide raster -- would otherwise put input + all planes of the raster on the device and
    // fail with ENOMEM where the untiled call streams
    if (ctx->host_chunk_rows > 0 && ctx->host_chunk_rows < chunk) chunk = ctx->host_chunk_rows;
    if (chunk < 64) chunk = 64;  // (a few rows of a very wide raster may exceed the budget; still correct)
    if (chunk > H) chunk = H;
    size_t in_bytes = 0, plane_bytes = 0, out_bytes = 0;
    auto size_chunk = [&]() {
        in_bytes = (size_t)(chunk + 2 * depth) * (size_t)W * in_es;
        plane_bytes = (size_t)chunk * (size_t)W * out_es;
        out_bytes = plane_bytes * (size_t)n_planes;
    };
    size_chunk();
    void* d_in[2] = {nullptr, nullptr};
    void* d_out[2] = {nullptr, nullptr};
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    int rc = XDEMHIP_OK;
    auto cleanup = [&]() {
        for (int q = 0; q < 2; ++q) {
            if (d_in[q]) (void)hipFree(d_in[q]);
            if (d_out[q]) (void)hipFree(d_out[q]);
            if (ev_in[q]) (void)hipEventDestroy(ev_in[q]);
            if (ev_comp[q]) (void)hipEventDestroy(ev_comp[q]);
            if (ev_out[q]) (void)hipEventDestroy(ev_out[q]);
        }
    };
    // pinned staging, kept by the context between calls (pinning hundreds of MiB costs more than moving them).  The budget sizes
    // 2 x (input + planes) of pinned host memory AND the same of device memory; where the host cannot pin that much the chunk is
    // halved until it can (down to 64 rows) instead of failing the call.  Option "host_release" frees the staging buffers.
    if (ctx->stage_in_bytes < in_bytes || ctx->stage_out_bytes < out_bytes) {
        auto drop_staging = [&]() {
            for (int q = 0; q < 2; ++q) {
                if (ctx->stage_in[q]) (void)hipHostFree(ctx->stage_in[q]);
                if (ctx->stage_out[q]) (void)hipHostFree(ctx->stage_out[q]);
                ctx->stage_in[q] = ctx->stage_out[q] = nullptr;
            }
            ctx->stage_in_bytes = ctx->stage_out_bytes = 0;
        };
        for (;;) {
            drop_staging();
            bool ok = true;
            for (int q = 0; q < 2 && ok; ++q)
                ok = hipHostMalloc(&ctx->stage_in[q], in_bytes, hipHostMallocDefault) == hipSuccess &&
                     hipHostMalloc(&ctx->stage_out[q], out_bytes, hipHostMallocDefault) == hipSuccess;
            if (ok) break;
            (void)hipGetLastError();
            drop_staging();
            if (chunk <= 64) return xd_fail(ctx, XDEMHIP_ENOMEM, "hipHostMalloc(staging) failed even for 64-row chunks");
            chunk = chunk / 2 < 64 ? 64 : chunk / 2;
            size_chunk();
        }
        ctx->stage_in_bytes = in_bytes;
        ctx->stage_out_bytes = out_bytes;
    }
    for (int q = 0; q < 2; ++q) {
        if (hipMalloc(&d_in[q], in_bytes) != hipSuccess || hipMalloc(&d_out[q], out_bytes) != hipSuccess ||
            hipEventCreateWithFlags(&ev_in[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_comp[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_out[q], hipEventDisableTiming) != hipSuccess) {
            cleanup();
            return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc / hipEventCreate (host path) failed");
        }
    }
    L.row_stride = W;
    const int nthreads = ctx->host_copy_threads;
    hipStream_t s_h2d = ctx->copy_streams[0], s_d2h = ctx->copy_streams[1];
    // `nthreads` threads, each copying its share of the byte ranges handed to it (caller's memory is pageable; fresh output
    // pages fault on first touch: both want many threads)
    struct Span { char* dst; const char* src; size_t bytes; };
    auto parallel_copy = [&](const std::vector<Span>& spans) {
        size_t total = 0;
        for (const Span& sp : spans) total += sp.bytes;
        if (total < ((size_t)4 << 20) || nthreads <= 1) {
            for (const Span& sp : spans) memcpy(sp.dst, sp.src, sp.bytes);
            return;
        }
        std::vector<std::thread> workers;
        const size_t per = (total + (size_t)nthreads - 1) / (size_t)nthreads;
        for (int t = 0; t < nthreads; ++t)
            workers.emplace_back([&, t]() {
                size_t lo = per * (size_t)t, hi = lo + per < total ? lo + per : total, pos = 0;
                for (const Span& sp : spans) {
                    const size_t a = pos > lo ? pos : lo, b = pos + sp.bytes < hi ? pos + sp.bytes : hi;
                    if (b > a) memcpy(sp.dst + (a - pos), sp.src + (a - pos), b - a);
                    pos += sp.bytes;
                    if (pos >= hi) break;
                }
            });
        for (auto& w : workers) w.join();
    };
    struct ChunkGeom { int64_t r0, r1, top, bot; };
    std::vector<ChunkGeom> chunks;
    for (int64_t r0 = 0; r0 < H; r0 += chunk) {
        ChunkGeom g;
        g.r0 = r0; g.r1 = (r0 + chunk < H) ? r0 + chunk : H;
        // rows of the caller's buffer (which itself may carry halo rows) available above / below this chunk
        g.top = (r0 + halo_top < depth) ? r0 + halo_top : depth;
        g.bot = (H + halo_bottom - g.r1 < depth) ? H + halo_bottom - g.r1 : depth;
        chunks.push_back(g);
    }
    const int nchunks = (int)chunks.size();
    auto drain = [&](int k) -> int {   // chunk k's planes: pinned staging -> caller's arrays (after its D2H has landed)
        const ChunkGeom& g = chunks[k];
        const int q = k & 1;
        if (hipEventSynchronize(ev_out[q]) != hipSuccess) return xd_fail(ctx, XDEMHIP_EHIP, "D2H copy failed");
        std::vector<Span> spans;
        const size_t nb = (size_t)(g.r1 - g.r0) * (size_t)W * out_es;
        for (int i = 0; i < n_planes; ++i)
            spans.push_back({static_cast<char*>(out_planes[i]) + (size_t)g.r0 * (size_t)W * out_es,
                             static_cast<const char*>(ctx->stage_out[q]) + (size_t)i * plane_bytes, nb});
        parallel_copy(spans);
        return XDEMHIP_OK;
    };
    for (int k = 0; k < nchunks && rc == XDEMHIP_OK; ++k) {
        const ChunkGeom& g = chunks[k];
        const int q = k & 1;
        const int64_t in_rows = g.top + (g.r1 - g.r0) + g.bot;
        // (buffers q are free: chunk k - 2 was drained two iterations ago, which followed its kernels and copies)
        {   // caller's rows -> pinned in[q] (row stride squeezed out), then H2D
            const char* src = static_cast<const char*>(dem) + (size_t)(g.r0 + halo_top - g.top) * (size_t)row_stride * in_es;
            std::vector<Span> spans;
            if (row_stride == W) spans.push_back({static_cast<char*>(ctx->stage_in[q]), src, (size_t)in_rows * (size_t)W * in_es});
            else
                for (int64_t r = 0; r < in_rows; ++r)
                    spans.push_back({static_cast<char*>(ctx->stage_in[q]) + (size_t)r * (size_t)W * in_es,
                                     src + (size_t)r * (size_t)row_stride * in_es, (size_t)W * in_es});
            parallel_copy(spans);
        }
        if (hipMemcpyAsync(d_in[q], ctx->stage_in[q], (size_t)in_rows * (size_t)W * in_es, hipMemcpyHostToDevice, s_h2d) != hipSuccess ||
            hipEventRecord(ev_in[q], s_h2d) != hipSuccess || hipStreamWaitEvent(ctx->stream, ev_in[q], 0) != hipSuccess) {
            rc = xd_fail(ctx, XDEMHIP_EHIP, "H2D copy failed");
            break;
        }
        L.dem = d_in[q];
        for (int bit = 0, i = 0; bit < XDEMHIP_ATTR_COUNT; ++bit)
            if (attr_mask & (1u << bit)) L.planes[bit] = static_cast<char*>(d_out[q]) + (size_t)(i++) * plane_bytes;
        L.H = g.r1 - g.r0; L.halo_top = g.top; L.halo_bottom = g.bot;
        if (k == 0) (void)hipEventRecord(ctx->ev_start, ctx->stream);
        rc = xd::launch_terrain(ctx, L);
        if (rc != XDEMHIP_OK) break;
        if (k == nchunks - 1) (void)hipEventRecord(ctx->ev_stop, ctx->stream);
        const size_t nb = (size_t)(g.r1 - g.r0) * (size_t)W * out_es;
        hipError_t e = hipEventRecord(ev_comp[q], ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(s_d2h, ev_comp[q], 0);
        for (int i = 0; i < n_planes && e == hipSuccess; ++i)   // (a short last chunk leaves gaps between its planes: one copy per plane)
            e = hipMemcpyAsync(static_cast<char*>(ctx->stage_out[q]) + (size_t)i * plane_bytes,
                               static_cast<const char*>(d_out[q]) + (size_t)i * plane_bytes, nb, hipMemcpyDeviceToHost, s_d2h);
        if (e == hipSuccess) e = hipEventRecord(ev_out[q], s_d2h);
        if (e != hipSuccess) { rc = xd_fail(ctx, XDEMHIP_EHIP, "D2H copy failed"); break; }
        // while the device works on chunk k, move chunk k - 1's planes into the caller's arrays
        if (k > 0) rc = drain(k - 1);
    }
    if (rc == XDEMHIP_OK) rc = drain(nchunks - 1);
    if (rc != XDEMHIP_OK) { (void)hipStreamSynchronize(s_h2d); (void)hipStreamSynchronize(s_d2h); }
    ctx->timed = (rc == XDEMHIP_OK);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e2 != hipSuccess && rc == XDEMHIP_OK) rc = xd_fail(ctx, XDEMHIP_EHIP, std::string("kernel failed: ") + hipGetErrorString(e2));
    cleanup();
    return rc;
}

}  // extern "C"
