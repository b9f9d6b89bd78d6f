// capi.hip -- extern "C" entry points of libxdemhip.so (declared in include/xdemhip.h): context
// management, argument validation, host<->device staging.  Kernels live in terrain.hip / nuthkaab.hip /
// variogram.hip.
#include <math.h>
#include <string.h>

#include <functional>
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

int xdemhip_reduction_calls(xdemhip_ctx* ctx, int64_t* host_calls, int64_t* device_calls) {
    if (!ctx) return XDEMHIP_EINVAL;
    if (host_calls) *host_calls = ctx->n_red_host;
    if (device_calls) *device_calls = ctx->n_red_dev;
    return XDEMHIP_OK;
}

int xdemhip_synchronize(xdemhip_ctx* ctx) {
    if (!ctx) return XDEMHIP_EINVAL;
    XD_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    XD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return XDEMHIP_OK;
}

int xdemhip_set_option(xdemhip_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return XDEMHIP_EINVAL;
    if (std::string(name) == "host_chunk_mb") {  // device budget of one row chunk of host-buffer terrain calls (0 = default 8 GiB)
        if (value < 0) return xd_fail(ctx, XDEMHIP_EINVAL, "host_chunk_mb must be >= 0");
        ctx->host_chunk_mb = value;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "host_copy_threads") {  // threads (one stream each) moving host-buffer rasters over PCIe (0 = default 8)
        if (value < 0 || value > xdemhip_ctx::MAX_COPY_THREADS) return xd_fail(ctx, XDEMHIP_EINVAL, "host_copy_threads must be 0..16");
        ctx->host_copy_threads = value ? value : 8;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "pairs_launch_cap") {  // test switch: split the pair passes into launches of at most this many workgroups
        if (value < 0) return xd_fail(ctx, XDEMHIP_EINVAL, "pairs_launch_cap must be >= 0");
        ctx->pairs_launch_cap = value;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "terrain_store") {
        if (value < 0 || value > 1) return xd_fail(ctx, XDEMHIP_EINVAL, "terrain_store: 0 direct, 1 staged row stores");
        ctx->terrain_store = value;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "terrain_rows") {
                // (values >= 100 select occupancy experiments of measurement builds and are ignored by the shipped library)
        if (value != 0 && value != 16 && value != 24 && value != 32 && value < 100) return xd_fail(ctx, XDEMHIP_EINVAL, "terrain_rows: 0, 16, 24 or 32");
        ctx->terrain_rows = value;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "nk_ext") {
        if (value < 0 || value > 1) return xd_fail(ctx, XDEMHIP_EINVAL, "nk_ext: 0 or 1");
        ctx->nk_ext = value;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "terrain_stream") {
        if (value != 0 && value != 1 && value != 2 && value != 3 && value != 128 && value != 256 && value != 512) return xd_fail(ctx, XDEMHIP_EINVAL, "terrain_stream: 0, 1, 128, 256 or 512");
        ctx->terrain_stream = value;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "terrain_sync") {
        if (value < 0 || value > 32 || (value & (value - 1))) return xd_fail(ctx, XDEMHIP_EINVAL, "terrain_sync: 0 or a power of two <= 32");
        ctx->terrain_sync = value;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "terrain_order") {
        if (value < 0 || value > 1) return xd_fail(ctx, XDEMHIP_EINVAL, "terrain_order: 0 XCD bands, 1 natural order");
        ctx->terrain_order = value;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "terrain_math") {
        if (value < 0 || value > 2) return xd_fail(ctx, XDEMHIP_EINVAL, "terrain_math: 0 mixed precision, 1 float64, 2 lean");
        ctx->terrain_math = value;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "vario_grid" || std::string(name) == "vario_edge" || std::string(name) == "vario_diff") {
        if (value < 0 || value > 1) return xd_fail(ctx, XDEMHIP_EINVAL, std::string(name) + ": 0 or 1");
        (std::string(name) == "vario_grid" ? ctx->vario_grid : std::string(name) == "vario_edge" ? ctx->vario_edge : ctx->vario_diff) = value;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "nk_nan_rule") {
        if (value < 0 || value > 3) return xd_fail(ctx, XDEMHIP_EINVAL, "nk_nan_rule: 0 4tap, 1 weighted, 2 dilate3x3, 3 dilate_cross");
        ctx->nk_nan_rule = value;
        return XDEMHIP_OK;
    }
    if (std::string(name) == "selection") {
        if (value < 0 || value > 3) return xd_fail(ctx, XDEMHIP_EINVAL, "selection: 0 auto, 1 plain, 2 degenerate brackets, 3 bracketed");
        ctx->selection_mode = value;
        return XDEMHIP_OK;
    }
    return xd_fail(ctx, XDEMHIP_EINVAL, std::string("unknown option: ") + name);
}

int xdemhip_last_kernel_ms(xdemhip_ctx* ctx, float* ms) {
    if (!ctx || !ms) return XDEMHIP_EINVAL;
    if (!ctx->timed) return xd_fail(ctx, XDEMHIP_EINVAL, "no timed launch on this context yet");
    XD_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_stop));
    XD_HIP_CHECK(ctx, hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
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
    L.window_size = window_size; L.degrees = degrees; L.attr_mask = attr_mask;
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
    // analogue of the reference's map_overlap tiles, xdem/terrain/terrain.py:412-466): device memory stays bounded by the
    // chunk budget however large the host raster is, and a chunk is computed exactly as the same rows of the whole raster
    // (halo_top / halo_bottom of the kernels).
    int depth = 0;  // overlap rows per side
    if (attr_mask & 0x3ffu) depth = surface_fit == XDEMHIP_FIT_FLORINSKY ? 2 : 1;
    if ((attr_mask & 0x5c00u) && window_size / 2 > depth) depth = window_size / 2;
    if ((attr_mask & XDEMHIP_ATTR_RUGOSITY) && depth < 1) depth = 1;
    const size_t budget = (size_t)(ctx->host_chunk_mb > 0 ? ctx->host_chunk_mb : 8192) << 20;
    const size_t row_bytes = (size_t)W * (in_es + out_es * (size_t)n_planes);
    int64_t chunk = (int64_t)(budget / (row_bytes ? row_bytes : 1)) - 2 * depth;
    if (chunk < 64) chunk = 64;  // (a few rows of a very wide raster may exceed the budget; still correct)
    if (chunk > H) chunk = H;
    void* d_dem = nullptr;
    std::vector<void*> d_out(n_planes, nullptr);
    int rc = XDEMHIP_OK;
    auto cleanup = [&]() {
        if (d_dem) (void)hipFree(d_dem);
        for (void* p : d_out)
            if (p) (void)hipFree(p);
    };
    const size_t in_bytes = (size_t)(chunk + 2 * depth) * (size_t)W * in_es;
    const size_t plane_bytes = (size_t)chunk * (size_t)W * out_es;
    if (hipMalloc(&d_dem, in_bytes) != hipSuccess) { cleanup(); return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc(dem) failed"); }
    for (int i = 0; i < n_planes; ++i)
        if (hipMalloc(&d_out[i], plane_bytes) != hipSuccess) { cleanup(); return xd_fail(ctx, XDEMHIP_ENOMEM, "hipMalloc(plane) failed"); }
    L.dem = d_dem;
    L.row_stride = W;
    for (int bit = 0, i = 0; bit < XDEMHIP_ATTR_COUNT; ++bit)
        if (attr_mask & (1u << bit)) L.planes[bit] = d_out[i++];
    bool first = true;
    for (int64_t r0 = 0; r0 < H && rc == XDEMHIP_OK; r0 += chunk) {
        const int64_t r1 = (r0 + chunk < H) ? r0 + chunk : H;
        // rows of the caller's buffer (which itself may carry halo rows) available above / below this chunk
        const int64_t top = (r0 + halo_top < depth) ? r0 + halo_top : depth;
        const int64_t bot = (H + halo_bottom - r1 < depth) ? H + halo_bottom - r1 : depth;
        const char* src = static_cast<const char*>(dem) + (size_t)(r0 + halo_top - top) * (size_t)row_stride * in_es;
        // Pageable host memory is staged by the runtime on the calling thread (and fresh output pages fault on first touch):
        // `nthreads` threads, each with its own stream and its share of the rows, keep more of the PCIe link busy than one.
        const int nthreads = ctx->host_copy_threads;
        const int64_t in_rows = top + (r1 - r0) + bot;
        auto run_threads = [&](const std::function<int(int, hipStream_t)>& body) -> bool {
            std::vector<std::thread> workers;
            std::vector<int> wrc(nthreads, 0);
            for (int t = 0; t < nthreads; ++t)
                workers.emplace_back([&, t]() {
                    if (hipSetDevice(ctx->device) != hipSuccess) { wrc[t] = 1; return; }
                    wrc[t] = body(t, ctx->copy_streams[t]);
                });
            for (auto& w : workers) w.join();
            for (int t = 0; t < nthreads; ++t)
                if (wrc[t]) return false;
            return true;
        };
        // (the previous chunk's kernels and copies have completed: every thread synchronised its stream)
        const bool up = run_threads([&](int t, hipStream_t st) -> int {
            const int64_t a = in_rows * t / nthreads, b = in_rows * (t + 1) / nthreads;
            if (b <= a) return 0;
            if (hipMemcpy2DAsync(static_cast<char*>(d_dem) + (size_t)a * (size_t)W * in_es, (size_t)W * in_es,
                                 src + (size_t)a * (size_t)row_stride * in_es, (size_t)row_stride * in_es, (size_t)W * in_es,
                                 (size_t)(b - a), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
            return hipStreamSynchronize(st) != hipSuccess;
        });
        if (!up) { rc = xd_fail(ctx, XDEMHIP_EHIP, "H2D copy failed"); break; }
        L.H = r1 - r0; L.halo_top = top; L.halo_bottom = bot;
        if (first) (void)hipEventRecord(ctx->ev_start, ctx->stream);
        rc = xd::launch_terrain(ctx, L);
        if (rc != XDEMHIP_OK) break;
        if (r1 == H) (void)hipEventRecord(ctx->ev_stop, ctx->stream);
        first = false;
        hipEvent_t done_ev = ctx->ev_copy;
        (void)hipEventRecord(done_ev, ctx->stream);
        // D2H: tasks = (plane, row slice); `slices` slices per plane so that every thread has work whatever the plane count
        const int slices = (nthreads + n_planes - 1) / n_planes > 1 ? (nthreads + n_planes - 1) / n_planes : 2;
        const int n_tasks = n_planes * slices;
        const int64_t rows = r1 - r0;
        const bool down = run_threads([&](int t, hipStream_t st) -> int {
            if (hipStreamWaitEvent(st, done_ev, 0) != hipSuccess) return 1;
            for (int k = t; k < n_tasks; k += nthreads) {
                const int i = k / slices, sl = k % slices;
                const int64_t a = rows * sl / slices, b = rows * (sl + 1) / slices;
                if (b <= a) continue;
                if (hipMemcpyAsync(static_cast<char*>(out_planes[i]) + (size_t)(r0 + a) * (size_t)W * out_es,
                                   static_cast<char*>(d_out[i]) + (size_t)a * (size_t)W * out_es, (size_t)(b - a) * (size_t)W * out_es,
                                   hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
            }
            return hipStreamSynchronize(st) != hipSuccess;
        });
        if (!down) rc = xd_fail(ctx, XDEMHIP_EHIP, "D2H copy failed");
    }
    ctx->timed = (rc == XDEMHIP_OK);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e2 != hipSuccess && rc == XDEMHIP_OK) rc = xd_fail(ctx, XDEMHIP_EHIP, std::string("kernel failed: ") + hipGetErrorString(e2));
    cleanup();
    return rc;
}

}  // extern "C"
