/* xdemhip.h -- C-ABI of libxdemhip.so: MI355X (gfx950) kernels for xDEM's three dense-array hot paths.
 *
 * The reference (GlacioHack/xdem) has no FFI: its operator boundary for these paths is a set of private
 * Python engine functions that take plain ndarrays.  Every entry point below cites
 * the reference function it stands in for; the ctypes binding a maintainer would add is in INTEGRATION.md
 * and is what xdem_amd/_lib.py implements.
 *
 * Conventions
 *  - every function returns 0 on success or a negative XDEMHIP_E* code; nothing throws; the text of the
 *    last error of a context is available through xdemhip_last_error();
 *  - the caller owns every buffer; the library owns only the opaque context (one per GPU / per process);
 *  - `memspace` says where the caller's buffers live: XDEMHIP_HOST (the library stages H2D/D2H itself) or
 *    XDEMHIP_DEVICE (raw device pointers, e.g. torch tensors' data_ptr(); work is enqueued on the context
 *    stream and the call returns without synchronising);
 *  - rasters are row-major, element (row, col) at base[row * row_stride + col]; all sizes are 64-bit.
 */
#ifndef XDEMHIP_H
#define XDEMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Threading contract: the library is re-entrant PER CONTEXT.  A context owns one stream, one scratch state and one queue of
 * deferred result copies; calls on the same context (and on plans / pair sets created from it) must not overlap in time --
 * drive a context from one thread at a time.  Different contexts may be used from different threads concurrently. */
typedef struct xdemhip_ctx xdemhip_ctx;

enum { XDEMHIP_OK = 0, XDEMHIP_EINVAL = -1, XDEMHIP_ENODEV = -2, XDEMHIP_EHIP = -3, XDEMHIP_ENOMEM = -4,
       XDEMHIP_EUNSUPPORTED = -5 };
enum { XDEMHIP_F32 = 0, XDEMHIP_F64 = 1 };
enum { XDEMHIP_HOST = 0, XDEMHIP_DEVICE = 1 };

/* surface_fit ids: xdem/terrain/surfit.py:1240 ; curvature method ids: surfit.py:1244 ; TRI: window.py:964 */
enum { XDEMHIP_FIT_HORN = 0, XDEMHIP_FIT_ZEVENBERGTHORNE = 1, XDEMHIP_FIT_FLORINSKY = 2 };
enum { XDEMHIP_CURV_GEOMETRIC = 0, XDEMHIP_CURV_DIRECTIONAL = 1 };
enum { XDEMHIP_TRI_RILEY = 0, XDEMHIP_TRI_WILSON = 1 };

/* Attribute bits.  Bits 0-9 follow the reference's fixed attribute order (surfit.py:407-418), bits 10-11
 * are the two windowed indexes of window.py:752-758 that are on the hot path, bit 12 the first "next" one. */
enum {
    XDEMHIP_ATTR_SLOPE = 1u << 0,
    XDEMHIP_ATTR_ASPECT = 1u << 1,
    XDEMHIP_ATTR_HILLSHADE = 1u << 2,
    XDEMHIP_ATTR_CURVATURE = 1u << 3,
    XDEMHIP_ATTR_PROFILE_CURVATURE = 1u << 4,
    XDEMHIP_ATTR_TANGENTIAL_CURVATURE = 1u << 5,
    XDEMHIP_ATTR_PLANFORM_CURVATURE = 1u << 6,
    XDEMHIP_ATTR_FLOWLINE_CURVATURE = 1u << 7,
    XDEMHIP_ATTR_MAX_CURVATURE = 1u << 8,
    XDEMHIP_ATTR_MIN_CURVATURE = 1u << 9,
    XDEMHIP_ATTR_TPI = 1u << 10,
    XDEMHIP_ATTR_TRI = 1u << 11,
    XDEMHIP_ATTR_ROUGHNESS = 1u << 12, /* max - min of the window (window.py:261-308) */
    XDEMHIP_ATTR_RUGOSITY = 1u << 13,  /* 8f-2: Jenness surface-area ratio, always 3x3, needs resolution (window.py:466-563) */
    XDEMHIP_ATTR_FRACTAL_ROUGHNESS = 1u << 14, /* 8f-2: box-counting dimension over window_size (window.py:316-401);
                                                   the reference runs it in a second engine call with its own
                                                   window_size_fractal (terrain.py:619-630): do the same here */
    XDEMHIP_ATTR_COUNT = 15
};

/* ---- context ------------------------------------------------------------------------------------- */
int xdemhip_version(void);
/* One context per process and GPU (the multi-GPU layer is one process per GPU, torch.distributed/RCCL). */
int xdemhip_create(int device_id, xdemhip_ctx** out_ctx);
void xdemhip_destroy(xdemhip_ctx* ctx);
const char* xdemhip_last_error(const xdemhip_ctx* ctx);
/* Enqueue on the caller's hipStream_t, e.g. torch.cuda.current_stream().cuda_stream -- NULL (0) is HIP's default stream,
 * which is what torch reports for its default stream: device-resident calls are then ordered with the caller's own kernels.
 * XDEMHIP_OWN_STREAM returns to the context's private non-blocking stream (the initial setting; host-buffer calls synchronise
 * it themselves). */
#define XDEMHIP_OWN_STREAM ((void*)(intptr_t)-1)
int xdemhip_set_stream(xdemhip_ctx* ctx, void* hip_stream);
int xdemhip_synchronize(xdemhip_ctx* ctx);
/* Device memory with a chosen PHYSICAL backing, for resident planes.  The streaming terrain kernel keeps
 * ~55 row streams going at once; in physically contiguous memory -- XDEMHIP_ALLOC_CONTIGUOUS, or an ordinary allocation on a box
 * whose free memory is one block -- they collide in the memory channels (14.4-15.6 ms for the 40000^2 set), in memory whose
 * pieces are scattered they do not (12.7-13.3 ms).  XDEMHIP_ALLOC_SCATTERED is the form to use: one virtual range over 32 MiB
 * physical pieces mapped in a fixed pseudo-random order (HIP virtual memory management).  `flags` = 0: hipMalloc.
 * XDEMHIP_ALLOC_CONTIGUOUS (hipExtMallocWithFlags / hipDeviceMallocContiguous; *got_contiguous -- optional -- tells whether the
 * driver had one piece), XDEMHIP_ALLOC_RECYCLED (allocate, touch, free, allocate again), XDEMHIP_ALLOC_CHUNKED (64 MiB pieces in
 * order) exist for measurements.  xdemhip_device_free takes all of them. */
enum { XDEMHIP_ALLOC_CONTIGUOUS = 1, XDEMHIP_ALLOC_RECYCLED = 2, XDEMHIP_ALLOC_CHUNKED = 4, XDEMHIP_ALLOC_SCATTERED = 8 };
int xdemhip_device_alloc(xdemhip_ctx* ctx, size_t bytes, int flags, void** ptr, int* got_contiguous);
int xdemhip_device_free(xdemhip_ctx* ctx, void* ptr);
/* Timing of the work enqueued by the last call on the context stream, measured with hipEvents recorded on
 * that stream around the kernel launch(es); returns milliseconds in *ms (synchronises the stop event). */
int xdemhip_last_kernel_ms(xdemhip_ctx* ctx, float* ms);
/* Diagnostics: the shader clock WHILE other work runs.  Enqueues on `hip_stream` (a stream of the caller's, not the context's: the
 * point is to run next to the context's kernels) one wave that sleeps `sleeps` x 8128 shader clocks (about 3.4 us each at 2.4 GHz)
 * and writes to out_device[0..1] (device memory, 16 bytes) the ticks of the shader-clock counter and of the constant 100 MHz
 * counter it saw go by: clock in MHz = 100 x out[0] / out[1].  No profiler, no SMI, nothing synchronises.  (bench.py launches one
 * per timed step: the same binary runs the terrain launch at 12.7 ms on one box and 13.9 ms on the next, and only the clock under
 * load tells a power-managed part from a slow placement of the planes.) */
int xdemhip_clock_probe(xdemhip_ctx* ctx, void* hip_stream, int sleeps, uint64_t* out_device);
/* OPTIONS of a context (fourteen names; every one of them has a GPU test).  Switches that only choose between internal routes
 * which must agree -- for the route-agreement tests and for measurements -- are not options: include/xdemhip_test.h.
 *  Host-buffer terrain calls
 *   "host_chunk_mb"     device-memory budget (MiB) of one row chunk (0 = default 288): host rasters of any size stream through the
 *                       GPU in row chunks with the overlap the attributes need.
 *   "host_chunk_rows"   rows per chunk where that is FEWER than the budget gives (0 = from the budget; at least 64 are taken) -- what
 *                       the reference's tiled call takes from `mp_config.chunk_size` (xdem/terrain/terrain.py:412-466); chunked and
 *                       one-pass results are bit-identical.
 *   "host_copy_threads" threads (one HIP stream each) that move host-buffer rasters over PCIe, rows split among them (0 = default 8,
 *                       at most 16).
 *   "host_release"      (any value) frees the pinned staging buffers the context keeps between such calls.
 *  Terrain
 *   "terrain_math"      precision recipe of float32 -> float32 launches: 2 (default) lean tail -- float64 where terms cancel, float32
 *                       scale factors; 0 mixed tail -- float64 everywhere but the arcsines; 1 float64 attribute math (what every other
 *                       dtype pair always runs).  All three meet the 1e-6 bar of the reference fixtures.
 *   "terrain_nonfinite" which of the reference's two rules decides what +-Inf pixels do to the surface-fit attributes -- 0 (default) the
 *                       SciPy engine's: an output is NaN iff its full window holds a non-finite value (surfit.py:1185-1192); 1 the
 *                       Numba engine's (surfit.py:948-1088, 1270-1303): no mask, the float64 loop over every tap and the formulas decide
 *                       (0 x Inf and Inf - Inf give NaN, other infinite windows give slope 90 deg etc.).  NaN pixels and raster edges
 *                       behave alike under both.
 *  Conventions of the third-party packages that nothing readable offline pins (oracle/pin_thirdparty.py decides them where the
 *  packages are importable; xdem_amd/thirdparty_decision.json then sets every context's defaults)
 *   "nk_nan_rule"       how nodata spreads through the bilinear taps of the Nuth-Kaab step / translation resample (geoutils'
 *                       _interp_points): 0 "4tap" (default; NaN if any of the four taps is non-finite or outside, zero weights included --
 *                       except a zero-weight tap beyond the last row / column, so a node exactly on the upper edge keeps its value), 1
 *                       "weighted" (zero-weight taps ignored everywhere), 2 "dilate3x3" (NaN if the 3 x 3 neighbourhood of the nearest
 *                       pixel holds a non-finite value), 3 "dilate_cross" (the same with the 4-connected cross: SciPy's default
 *                       binary-dilation structure).  Read when a plan is created / a resample is launched.
 *   "vario_edge"        lag classes of the pair kernels, 0 = [e_{k-1}, e_k) (default), 1 = (e_{k-1}, e_k].
 *   "vario_diff"        |dv| formed 0 = in the value dtype (default), 1 = in float64 (float32 values widened at xdemhip_pairs_create).
 *  Exact order statistics and the Nuth-Kaab step (results are identical under every value; tests hold the routes against each other)
 *   "selection"         how the exact medians (nanmedian of dh, aspect-bin and nd_binning medians, NMAD, Dowd) are selected -- 0 (default)
 *                       bracketed for large inputs: brackets from a ~1/64 sample, one counting + compaction pass, exact selection among
 *                       the candidates, plain radix passes if a bracket misses or the input is too small per bin; 1 plain 8-bit radix
 *                       passes only (the fall-back, and the check of the other); 2 degenerate brackets (exercises the fall-back); 3
 *                       bracketed even where the per-bin sample is small.
 *   "nk_fused"          1 (default) the Nuth-Kaab step of large plans is ONE data pass (13 B/pixel), 0 the plain route (stored dh and selections
 *                       over it: what a one-pass step hands over to when a bracket misses or a buffer overflows).
 *   "nk_fused_dist"     1 (default) partitioned plans (reduction hook + xdemhip_set_rank) take the one-pass step too, 0 the plain route.
 *   "nk_predict"        1 (default) a settled one-pass step takes its brackets from the previous step's exact medians (see
 *                       xdemhip_nk_predict_counts), 0 every step samples.
 *   "pairs_launch_cap"  workgroups per launch of the variogram pair passes (0 = default 2^31 / workgroup size, the most a HIP dispatch
 *                       holds; a pass over more tiles goes out as several launches -- a small value exercises that path). */
int xdemhip_set_option(xdemhip_ctx* ctx, const char* name, int value);

/* Multi-GPU hook for the accumulator-style paths (Nuth-Kaab reductions): one process per GPU, every rank works on its
 * share and calls the same entry points; wherever a global reduction is needed the library hands a small HOST array of
 * `count` 8-byte elements to `fn`, which must combine it in place over all ranks (e.g. torch.distributed.all_reduce over
 * RCCL) and return 0.  kind: 0 = sum of uint64, 1 = sum of float64, 2 = min of uint64, 3 = max of uint64.  Integer
 * histograms / counts make the reductions exact and order-independent.  fn == NULL restores single-process behaviour. */
enum { XDEMHIP_RED_SUM_U64 = 0, XDEMHIP_RED_SUM_F64 = 1, XDEMHIP_RED_MIN_U64 = 2, XDEMHIP_RED_MAX_U64 = 3 };
typedef int (*xdemhip_allreduce_fn)(void* host_array, int64_t count, int kind, void* user);
int xdemhip_set_allreduce(xdemhip_ctx* ctx, xdemhip_allreduce_fn fn, void* user);
/* Device-side form of the same hook (round 3): the library's per-pass reductions (integer histograms, counters, min / max
 * keys) live in DEVICE memory; with this hook installed next to the host one they are handed over as they are -- `fn` gets
 * the device pointer and the hipStream_t (as void*) the library's work is queued on, must ENQUEUE the in-place all-reduce so
 * that it is ordered after the work already on that stream and before whatever is queued next (RCCL: ncclAllReduce on that
 * stream, or on another one bracketed by events), and returns without waiting: no D2H / H2D staging, no host
 * synchronisation per reduction.  The host hook stays in use for the few host-side scalars (route agreements).  NULL
 * removes it (every reduction is then staged through the host hook).  xdemhip_reduction_calls reports how many reductions
 * went through each hook since the context was created. */
typedef int (*xdemhip_allreduce_device_fn)(void* device_array, int64_t count, int kind, void* hip_stream, void* user);
int xdemhip_set_allreduce_device(xdemhip_ctx* ctx, xdemhip_allreduce_device_fn fn, void* user);
int xdemhip_reduction_calls(xdemhip_ctx* ctx, int64_t* host_calls, int64_t* device_calls);
/* ---- host-side preparation of the variogram path in native code (no context, no GPU: plain C++ on `threads` host threads) ----------
 * xdem's default sampler of raster variograms is scikit-gstat's RasterEquidistantMetricSpace (xdem/spatialstats.py:1185-1260; un-vendored):
 * per run a random centre, a centre disk and rings whose radii grow by sqrt(2), up to `samples` pixels of each, every centre-disk pixel
 * paired with every ring pixel.  xdemhip_host_ring_sample draws, for `runs` centres (cx, cy: pixel column / row) and `n_rings` rings
 * ring_lo[k] <= d < ring_hi[k] (distances in units of gsd x pixels, float64 arithmetic as NumPy forms them), up to `samples` distinct
 * pixels of every ring uniformly without replacement -- valid[iy * nx + ix] != 0 only, if `valid` is given -- as flat indexes iy * nx + ix
 * into out_idx[run][ring][0 .. out_count[run][ring]) (padded with -1; a ring that holds no more than `samples` pixels is returned whole,
 * in raster order, any other in random order).  Every (run, ring) has its own random stream derived from `seed`: the result does not
 * depend on `threads`.  Returns XDEMHIP_OK / XDEMHIP_EINVAL. */
int xdemhip_host_ring_sample(const uint8_t* valid, int64_t ny, int64_t nx, double gsd, int64_t runs, const int64_t* cx, const int64_t* cy,
                             int n_rings, const double* ring_lo, const double* ring_hi, int64_t samples, uint64_t seed, int threads,
                             int64_t* out_idx, int64_t* out_count);
/* Coordinates (x = ix gsd, y = iy gsd, float64) and values (`dtype` XDEMHIP_F32 / XDEMHIP_F64, gathered from the row-major raster
 * `values` with `nx` columns) of `n_blocks` point sets given as flat pixel indexes idx[off[b] .. off[b + 1]), in the order given (x_out,
 * y_out, v_out) and -- if sx_out is not NULL -- once more with every set permuted into Z-order over its own bounding box (sx_out, sy_out,
 * sv_out: the slot order the pair kernels' run-length accumulation wants). */
int xdemhip_host_gather_points(const void* values, int dtype, int64_t nx, double gsd, int n_blocks, const int64_t* off, const int64_t* idx,
                               int threads, double* x_out, double* y_out, void* v_out, double* sx_out, double* sy_out, void* sv_out);
/* np.isfinite(values) over a host array of gigabytes on `threads` threads (the first thing sample_empirical_variogram does with its
 * raster, xdem/spatialstats.py:1404-1410: the NaN filter of its values): the number of finite elements, and -- if valid_out is not
 * NULL -- the 0 / 1 mask itself (a caller whose raster turns out to be all finite never needs one). */
int xdemhip_host_count_finite(const void* values, int dtype, int64_t n, int threads, int64_t* n_finite, uint8_t* valid_out);
/* This process's place among the ranks the hooks reduce over (round 5).  The hooks only combine; a few exchanges of the one-pass
 * Nuth-Kaab step on partitioned plans need every rank's contribution SEPARATELY (per-rank histogram rows, per-rank slices of a small
 * key list): they travel as sum all-reduces in which every rank fills its own slot and adds zeros to the others', for which the
 * library must know `rank` in [0, `world`).  world = 0 (default) = not told: partitioned plans keep the plain route.  Ranks
 * must be numbered the same way on every process of the group (torch.distributed's group rank does). */
int xdemhip_set_rank(xdemhip_ctx* ctx, int rank, int world);

/* ---- path 1: terrain stencil engine --------------------------------------------------------------
 * Replaces  _get_surface_attributes(dem, resolution, surface_attributes, out_dtype, surface_fit,
 *           curv_method, engine, hillshade_*)                       xdem/terrain/surfit.py:1197-1305
 *      and  _get_windowed_indexes(dem, window_size, windowed_indexes, resolution, out_dtype, tri_method,
 *           engine) for TPI / TRI                                   xdem/terrain/window.py:926-1002
 *      plus the unit conversion / clip of _get_terrain_attribute     xdem/terrain/terrain.py:586-596
 * in ONE fused pass: one DEM read, one write per requested attribute.
 *
 *  dem          first row of the buffer.  Output row r (0 <= r < H) is buffer row (halo_top + r); the
 *               buffer holds halo_top + H + halo_bottom rows.  Halo rows are neighbour data of a
 *               row-block partition (multi-GPU); rows beyond them count as outside the raster (NaN), as
 *               do columns outside [0, W).  Single raster: halo_top = halo_bottom = 0.
 *  attr_mask    OR of XDEMHIP_ATTR_*; out_planes[k] receives the k-th SET bit in ascending bit order,
 *               each an (H, W) plane with row stride W in out_dtype.
 *  degrees      bit 0: slope / aspect in degrees (terrain.py:586-591), else radians.  Bit 1 (value 2, or 3 with degrees): the
 *               hillshade plane as the reference's ENGINE returns it (surfit.py:609-622), WITHOUT the caller's clip to [0, 255]
 *               (terrain.py:594-596) that every other call fuses -- for bindings at the engine boundary
 *               (_get_surface_attributes); such a launch takes the float64 attribute tail.
 *  window_size  odd window of TPI / TRI / roughness (reference default 3) and of fractal roughness (reference default
 *               13 through its own window_size_fractal: request that attribute in a call of its own); rugosity is 3x3.
 * NaN / +-Inf in the DEM are nodata: an output pixel is NaN iff its full window (3x3 Horn/ZT, 5x5
 * Florinsky; window_size for TPI/TRI) holds a non-finite value or leaves the raster (surfit.py:1185-1192).
 */
int xdemhip_terrain(xdemhip_ctx* ctx, const void* dem, int dem_dtype, int64_t H, int64_t W, int64_t row_stride,
                    int64_t halo_top, int64_t halo_bottom, double resolution, int surface_fit, int curv_method,
                    uint32_t attr_mask, int tri_method, int window_size, double hillshade_altitude_deg,
                    double hillshade_azimuth_deg, double hillshade_z_factor, int degrees, int out_dtype,
                    void* const* out_planes, int memspace);

/* Host-only helper (no GPU needed): the regression abscissae of fractal roughness for a window size, with NumPy's
 * float16 arithmetic reproduced (xdem/terrain/window.py:362-393: the divisors q of window_size//2 are a uint8 array, so
 * np.log / np.mean / SS_xx are float16).  Fills q[], log_q[] (float16 values as double) and returns the number of
 * divisors (<= max_q), or a negative status. */
int xdemhip_fractal_constants(int window_size, int max_q, int* q, double* log_q, double* mean_log_q, double* ss_xx);

/* Texture shading, the frequency-domain attribute of get_terrain_attribute: replaces
 *   _texture_shading_fft(dem, alpha)   xdem/terrain/freq.py:63-148   (called at xdem/terrain/terrain.py:637-644)
 * non-finite pixels are filled with the NaN-ignoring mean, the raster is padded symmetrically to the next 2/3/5/7-smooth
 * size, multiplied by |f|^alpha in the frequency domain (DC zeroed when alpha > 0) and cropped; invalid pixels -> NaN.
 * 0 <= alpha <= 2.  The FFT itself is hipFFT's (loaded on first use), in the DEM's precision like scipy.fft. */
int xdemhip_texture_shading(xdemhip_ctx* ctx, const void* dem, int dem_dtype, int64_t H, int64_t W, double alpha, int out_dtype,
                            void* out, int memspace);

/* ---- path 2: Nuth & Kaab (2011) inner loop ------------------------------------------------------------
 * Replaces the array work of  nuth_kaab(ref_elev, tba_elev, inlier_mask, transform, ...)   xdem/coreg/affine.py:539-609
 * for two rasters on the same grid, i.e. what NuthKaab._fit_rst_rst reaches (affine.py:2458-2522):
 *
 *  xdemhip_nk_create   once per fit: slope tangent / aspect from np.gradient (affine.py:433-438), zero slopes -> NaN
 *                      (affine.py:578-579), valid = inlier & finite(ref, tba, slope_tan, aspect) (base.py:650-661);
 *                      *n_valid = number of valid pixels (the reference's `subsample_final` for subsample == 1).
 *  xdemhip_nk_step     one _nuth_kaab_iteration_step (affine.py:477-536) up to, not including, the 72-point curve
 *                      fit: for the current offsets (georeferenced units, east / north)
 *                        dh      = ref - bilinear(tba)(row - shift_y/res_y, col + shift_x/res_x)     [convention: DESIGN.md]
 *                        vshift  = np.nanmedian(dh)                          (exact, radix selection)
 *                        y       = (dh - vshift) / slope_tan,  y_mean / y_std = np.nanmean / np.nanstd (for p0)
 *                        edges[n_bins+1], counts[n_bins], medians[n_bins] = scipy.stats.binned_statistic(aspect, y,
 *                                  np.nanmedian, bins=n_bins) with 'count'      (xdem/spatialstats.py:143-157)
 *                      Empty bins give NaN medians.  Returns XDEMHIP_EINVAL ("The subsample contains no more valid
 *                      values.") when no pixel survives, like affine.py:510-515.
 *  xdemhip_binned_median  the binning alone on caller-supplied 1-D (x, y) host arrays (nd_binning, 1 variable).
 * Host code fits a*cos(b - x) + c to (bin mids, medians) with scipy.optimize.curve_fit exactly as base.py:1038-1045.
 */
typedef struct xdemhip_nk_plan xdemhip_nk_plan;
int xdemhip_nk_create(xdemhip_ctx* ctx, const void* ref, const void* tba, const uint8_t* inlier_mask_or_null, int dtype,
                      int64_t H, int64_t W, int memspace, xdemhip_nk_plan** out_plan, int64_t* n_valid);
int xdemhip_nk_step(xdemhip_nk_plan* plan, double shift_x, double shift_y, double res_x, double res_y, int n_bins,
                    double* vshift, int64_t* n_valid, double* y_mean, double* y_std, double* edges, int64_t* counts,
                    double* medians);
/* Multi-GPU, partitioned (the production layout; structural ancestor: xdem/coreg/blockwise.py:174): this rank holds only
 * raster rows [row_begin - halo_top, row_end + halo_bottom) of ref / tba / inlier mask -- its own row block plus halo rows
 * copied from the neighbours (xdem_amd.dist.RowBlock exchanges them over RCCL send / recv).  The gradient needs ONE halo
 * row; the bilinear taps of a step need floor(|shift_y / res_y|) + 1 more, so the halo bounds the vertical shift a fit may
 * reach: xdemhip_nk_step returns XDEMHIP_EINVAL ("halo too small ...") when a step would leave it, and the caller
 * re-creates the plan with a deeper halo.  Install the all-reduce hook (and xdemhip_set_rank) BEFORE this call: n_valid is the
 * global count, and every reduction of a step (histograms, counters, min / max keys, successor keys) goes through the hook. */
int xdemhip_nk_create_block(xdemhip_ctx* ctx, const void* ref_block, const void* tba_block, const uint8_t* inlier_block_or_null,
                            int dtype, int64_t H, int64_t W, int64_t row_begin, int64_t row_end, int64_t halo_top,
                            int64_t halo_bottom, int memspace, xdemhip_nk_plan** out_plan, int64_t* n_valid);
/* Multi-GPU, replicated (every rank holds the full ref / tba): restrict this rank's work to raster rows
 * [row_begin, row_end); the streaming passes and their histograms are sharded by row block and combined through the
 * all-reduce hook.  Call right after xdemhip_nk_create on every rank; n_valid then returns the global count. */
int xdemhip_nk_set_rows(xdemhip_nk_plan* plan, int64_t row_begin, int64_t row_end, int64_t* n_valid);
/* Debug / test access: copy the auxiliary rasters back to host buffers (any pointer may be NULL). */
/* Statistic of the aspect bins (NuthKaab(bin_statistic=...), xdem/coreg/affine.py:2404): XDEMHIP_BINSTAT_MEDIAN (default,
 * np.nanmedian -- exact selection) or XDEMHIP_BINSTAT_MEAN (np.nanmean: per-bin float64 sums and counts in one pass, sum-
 * reducible across GPUs; equals NumPy's float32 pairwise mean to rounding, not bit for bit).  xdemhip_nk_step then returns
 * the bin means in `medians`. */
/* NuthKaab(bin_before_fit=False) (the mode of the reference's own synthetic tests, tests/test_coreg/test_affine.py:163-239):
 * the step without binning.  curve_fit then runs on every valid point (xdem/coreg/base.py:975-989); the model
 * a cos(b - x) + c is linear in (a cos b, a sin b, c), so its optimum follows from ten float64 sums over the points --
 * sums = [n, S cos x, S sin x, S cos^2, S sin^2, S cos sin, S y, S y cos, S y sin, S y^2] with x = aspect, y = (dh - vshift) /
 * slope_tan widened to float64 like curve_fit widens its inputs -- which is what this entry returns next to vshift,
 * n_valid and the p0 ingredients (np.nanmean / np.nanstd of y). */
int xdemhip_nk_step_fit(xdemhip_nk_plan* plan, double shift_x, double shift_y, double res_x, double res_y, double* vshift,
                        int64_t* n_valid, double* y_mean, double* y_std, double* sums /* [10] */);
/* NuthKaab(bin_statistic=<a callable other than np.nanmedian / np.nanmean>) (xdem/coreg/affine.py:2404; nd_binning hands the callable to
 * scipy.stats.binned_statistic, xdem/spatialstats.py:143-157): host code cannot run on the GPU, so this entry returns what the callable is
 * applied to -- y = (dh - vshift) / slope_tan in the DEM dtype and the aspect-bin id (uint16; 0xFFFF = no bin) of every pixel of the raster
 * in raster order, NaN / 0xFFFF where the pixel has no dh at this shift -- next to vshift, the valid count, np.nanmean / np.nanstd of y and
 * the n_bins + 1 bin edges.  `y_out` holds H x W values of the plan's dtype, `bins_out` H x W uint16, both in `memspace`.  Whole-raster
 * plans without a reduction hook only (a partitioned plan holds a part of every bin: XDEMHIP_EINVAL); the step takes the plain route. */
int xdemhip_nk_step_values(xdemhip_nk_plan* plan, double shift_x, double shift_y, double res_x, double res_y, int n_bins, double* vshift,
                           int64_t* n_valid, double* y_mean, double* y_std, double* edges /* [n_bins + 1] */, void* y_out,
                           uint16_t* bins_out, int memspace);
/* Explicit aspect-bin edges (NuthKaab(bin_sizes={"aspect": edges}), the array form of scipy.stats.binned_statistic's `bins`):
 * n_edges increasing values (2 .. 129: at most 128 bins, one histogram sweep); xdemhip_nk_step must then be called with
 * n_bins = n_edges - 1 (its output arrays are sized by n_bins; any other value is XDEMHIP_EINVAL).  `decimal` = SciPy's
 * `int(-log10(min edge spacing)) + 6` for these edges in the sample dtype (its rule for samples at or beyond the rightmost
 * edge, _binned_statistic.py:_bin_numbers).  n_edges = 0 restores SciPy's automatic edges. */
int xdemhip_nk_set_bin_edges(xdemhip_nk_plan* plan, const double* edges, int n_edges, int decimal);
#define XDEMHIP_BINSTAT_MEDIAN 0
#define XDEMHIP_BINSTAT_MEAN 1
int xdemhip_nk_set_statistic(xdemhip_nk_plan* plan, int bin_stat);
int xdemhip_nk_get_aux(xdemhip_nk_plan* plan, void* slope_tan, void* aspect, uint8_t* valid);
/* The random subsample of the valid pixels (replaces the host round trip of _get_subsample_on_valid_mask, xdem/coreg/base.py:577-617,
 * whose draw is `rng.choice(np.flatnonzero(valid), k, replace=False)` = flatnonzero(valid)[rng.choice(n_valid, k, replace=False)]): the
 * caller draws `k` distinct RANKS in [0, n_valid) -- positions among the plan's valid pixels in raster order -- and the plan keeps
 * exactly those pixels as inliers (the valid mask never travels to the host; rasters and aux variables stay where they are; a second
 * call draws again among the pixels still valid).  `ranks`: int64, host or device (`memspace`).  Whole-raster plans of one process
 * only (XDEMHIP_EINVAL otherwise: partitioned plans pass the drawn mask as their inlier mask).  n_valid returns the new count (= k). */
int xdemhip_nk_subsample(xdemhip_nk_plan* plan, const int64_t* ranks, int64_t k, int memspace, int64_t* n_valid);
/* How the steps of this plan were answered so far (either pointer may be NULL).  Two routes since round 6:
 *   ONE-PASS  one data pass of 13 B/pixel -- the shifted elevation difference, the counting for its exact median and the
 *             aspect-bin counting against brackets with per-pixel margins (large plans, median statistic, context option
 *             "nk_fused" = 1, the default); around it per-bin candidate segments with one workgroup per bin, a value-bucket
 *             selection of the median of dh, sample passes that advance their own selection states: 24 launches per step with
 *             sampled brackets, 14 with predicted ones (option "nk_predict");
 *   PLAIN     dh written by a generic kernel that reads mask and aspect, then the selections of select_run.h over the stored
 *             arrays (bracketed where the size pays, plain digit passes otherwise): small rasters, the mean statistic, the
 *             un-binned fit, "nk_fused" = 0, and the fall-back of a one-pass step whose brackets missed or overflowed.
 * (The queued two-pass route of rounds 2-5 is retired: no plan needed it.)  Results are identical on both routes (integer
 * counts, exact selections); nanmean / nanstd of y -- the p0 of the curve fit -- agree to 2e-6 of the spread on the one-pass
 * route (float32 partial sums, the accuracy class of the reference's own float32 np.nanmean).
 * PARTITIONED plans (reduction hook installed, xdemhip_set_rank told, context option "nk_fused_dist" = 1, the default) take the
 * one-pass step as well: one data pass over the rank's own rows and ten all-reduces per sampled step, five per predicted one
 * (nuthkaab.hip: "the ONE-PASS step on PARTITIONED plans"), all of them enqueued through the device hook where it is installed;
 * every rank returns the same integers as a single-GPU fit of the whole rasters. */
int xdemhip_nk_route_counts(xdemhip_nk_plan* plan, int64_t* onepass, int64_t* plain);
/* Round 6: how many of the one-pass steps took PREDICTED brackets -- the previous step's exact medians moved by the Nuth-Kaab
 * model for the change of the shift, instead of brackets from a fresh 1/64 sample: no sample kernels and no digit passes over
 * samples on a settled fit (context option "nk_predict", default 1; 0 = every step samples) --, how many predicted only the bracket
 * of the median of dh (steps that still move too far for the bins: the dh sample's three digit passes are skipped, the bins'
 * brackets are sampled), and how many predictions missed (such a step is run again with sampled brackets: the integers returned
 * are the same either way). */
int xdemhip_nk_predict_counts(xdemhip_nk_plan* plan, int64_t* predicted, int64_t* predicted_dh_only, int64_t* missed);
void xdemhip_nk_destroy(xdemhip_nk_plan* plan);
int xdemhip_binned_median(xdemhip_ctx* ctx, const void* x, const void* y, int dtype, int64_t n, int n_bins, double* edges,
                          int64_t* counts, double* medians);

/* "Next" row f4: the dense step of the patches method -- mean_filter_nan(img, kernel_size, kernel_shape)
 * (xdem/spatialstats.py:2597-2655: two scipy.ndimage.convolve calls with a ones / circular uint8 kernel, mode="constant",
 * cval=nan).  kernel_shape 0 = square, 1 = circular (_create_circular_mask, spatialstats.py:880-904).  Outputs are float64
 * (H, W) like the reference's: mean of the finite pixels under the kernel and their number; a window with a kernel pixel
 * outside the raster gives (NaN, 0) as SciPy's NaN border value does.  *n_kernel_px = np.count_nonzero(kernel).  Kernels of
 * more than 127 pixels (kernel_size > 11 square, > 13 circular) return XDEMHIP_EUNSUPPORTED: the reference counts in int8,
 * which wraps there. */
int xdemhip_mean_filter_nan(xdemhip_ctx* ctx, const void* img, int dtype, int64_t H, int64_t W, int kernel_size, int kernel_shape,
                            double* mean_out, double* nvalid_out, int* n_kernel_px, int memspace);

/* Row a5 as the PUBLIC function: xdem.spatialstats.convolution(imgs, filters, method)
 * (xdem/spatialstats.py:2558-2594; called by the reference's surface fit, surfit.py:1107, and by its tests,
 * tests/test_terrain/test_surfit.py:542-563, 612): `n_img` images (H x W, `dtype` float32 / float64, contiguous) against
 * `n_f` filters (M1 x M2, float64, HOST memory whatever `memspace` says) -> out float64 (n_img, n_f, H, W).
 * method 0 = "scipy": scipy.ndimage.convolve(img, filter, mode="constant", cval=nan) per pair (_scipy_convolution,
 * spatialstats.py:2512-2525) -- true convolution, weights with |w| <= DBL_EPSILON skipped, NaN beyond the border, double
 * accumulation in SciPy's tap order, rounded to the image dtype; method 1 = "numba": the loop of _numba_convolution
 * (spatialstats.py:2528-2555) on the NaN-padded images of 2582-2585 -- correlation over every tap, unrounded float64, last
 * row / column left 0 for an even filter size.  Bit for bit in both (tests/test_convolution_gpu.py). */
int xdemhip_convolution(xdemhip_ctx* ctx, const void* imgs, int dtype, int64_t n_img, int64_t H, int64_t W, const double* filters,
                        int n_f, int M1, int M2, int method, double* out, int memspace);

/* "Next" row f3, consumer side: xdem.spatialstats.get_perbin_nd_binning(df, list_var, list_var_names,
 * statistic, min_count) (xdem/spatialstats.py:425-527; used by the bias corrections, xdem/coreg/biascorr.py:302) as ONE
 * lookup per pixel instead of a mask per bin.  `vars[k]` = variable k (n values, `var_dtypes[k]` float32 / float64, in
 * `memspace`); host tables: variable k's `n_intervals[k]` sorted unique intervals [left, right) concatenated in `left` /
 * `right` (float64, already rounded to the dtype NumPy compares in), `table` / `pass` over the Cartesian product of the
 * intervals in itertools.product order (last variable fastest): the bin's statistic and 1 = write it (count > min_count),
 * 0 = leave NaN, 2 = the DataFrame has no such row.  `disjoint` != 0: the intervals of every variable are pairwise disjoint
 * (nd_binning's output) -- one containing interval per variable; 0: overlapping intervals, the product is walked per pixel in
 * upstream's order and the last bin that writes wins.  out float64 (n); *n_missing = pixels lying in a bin of kind 2 (upstream
 * raises IndexError there, and so does the Python side). */
int xdemhip_perbin_lookup(xdemhip_ctx* ctx, const void* const* vars, const int* var_dtypes, int n_var, int64_t n,
                          const int* n_intervals, const double* left, const double* right, const double* table,
                          const unsigned char* pass, int disjoint, double* out, int64_t* n_missing, int memspace);

/* "Next" row f1: the step right after a Nuth-Kaab fit -- resampling the translated DEM back onto
 * its own grid, i.e. _reproject_horizontal_shift_samecrs(raster_arr, src_transform, dst_transform)
 * (xdem/coreg/base.py:1615-1655) as used by Coreg.apply for pure translations:
 *     out(r, c) = bilinear(src)(r + shift_row_px, c + shift_col_px) + dz      (same tap convention as xdemhip_nk_step) */
int xdemhip_shift_bilinear(xdemhip_ctx* ctx, const void* src, int dtype, int64_t H, int64_t W, double shift_row_px,
                           double shift_col_px, double dz, void* out, int memspace);

/* ---- path 3: empirical variogram, pairwise lag binning -----------------------------------------------
 * Replaces the pairwise work sample_empirical_variogram delegates to scikit-gstat:
 *   skg.Variogram(coordinates, values, bin_func=<right edges>, maxlag=, estimator=)      xdem/spatialstats.py:1091  (pdist)
 *   skg.Variogram(RasterEquidistantMetricSpace / ProbabalisticMetricSpace, values, ...)  xdem/spatialstats.py:1247-1255 (cdist)
 *   -> V.get_empirical(bin_center=False), V.bin_count
 * A "pair set" is a list of blocks; block r pairs every point A_r[i] with every point B_r[j] (b_off != NULL), or
 * all i < j inside A_r (b_off == NULL).  Points are SoA (x, y float64; value float32/float64), blocks concatenated,
 * a_off / b_off hold n_blocks + 1 offsets.  Lag class k: right_edges[k-1] <= d < right_edges[k] (edges[-1] := 0);
 * d >= right_edges[n_bins-1] is dropped.  Pairs with a NaN value difference are dropped.
 *
 *  xdemhip_pairs_sums   kind 0: sums[k] = sum |dv|^2 (Matheron), kind 1: sum sqrt|dv| (Cressie-Hawkins); counts[k]
 *  xdemhip_pairs_hist   one radix-select pass for the exact per-class median of |dv| (Dowd): histogram (n_bins x 256,
 *                       uint64) of key bits [shift, shift+8) among pairs whose higher key bits equal prefix[k]
 *                       (first != 0: all pairs).  Keys are order-preserving integer images of |dv|: its IEEE bits
 *                       shifted left by one (the sign bit of |dv| is always 0), 32 bit for float32 values, 64 bit for
 *                       float64.  The host advances the selection; integer
 *                       histograms / counts can be summed over GPUs (all-reduce) before doing so.
 *  xdemhip_pairs_succ   succ[k] = smallest key > key[k] in class k (all-ones if none): upper median of even classes.
 */
typedef struct xdemhip_pairs xdemhip_pairs;
int xdemhip_pairs_create(xdemhip_ctx* ctx, int n_blocks, const int64_t* a_off, const double* ax, const double* ay,
                         const void* av, const int64_t* b_off, const double* bx, const double* by, const void* bv,
                         int val_dtype, const double* right_edges, int n_bins, int memspace, xdemhip_pairs** out,
                         int64_t* n_pairs);
int xdemhip_pairs_sums(xdemhip_pairs* pairs, int kind, double* sums, int64_t* counts);
int xdemhip_pairs_hist(xdemhip_pairs* pairs, int shift, int first, const uint64_t* prefix, uint64_t* hist);
int xdemhip_pairs_succ(xdemhip_pairs* pairs, const uint64_t* key, uint64_t* succ);
/* Exact per-class median of |dv| (np.median semantics: mean of the two middle values in the value dtype for even classes,
 * NaN for empty ones) and the class counts, selection state kept on the device: Dowd's estimator is
 * 2.198 * median^2 / 2.  Large pair sets (>= 4e9 pairs) are bracketed first -- digit passes over a 1/64 sample of the
 * (A tile x B tile) units, ONE pass over all pairs that counts and compacts the candidates, exact selection among them --
 * with the plain 4 (float32) / 8 (float64) digit passes + successor pass as the fall-back and for small sets.  Histograms,
 * counters and successor keys go through the xdemhip_set_allreduce hook when the pair blocks are sharded over GPUs. */
int xdemhip_pairs_medians(xdemhip_pairs* pairs, int64_t* counts, double* medians);
/* Tell `pairs` that `sorted` holds THE SAME blocks (same sizes, value dtype, edges, context) with the points of every block
 * in an order in which neighbouring slots are neighbouring points (Morton order): the one pass of xdemhip_pairs_medians that
 * visits every pair -- counting against the brackets and compacting the candidates -- then reads `sorted`'s points and
 * counts run-length (a lane touches its LDS counters once per run of equal lag class instead of once per pair).  Counts,
 * candidates and medians do not depend on the slot order.  The sampled digit passes keep `pairs`' own order (their
 * statistics assume unsorted tiles).  `sorted` is not owned and must outlive the calls; NULL (or `pairs` itself) unlinks.
 * (Test switch "vario_runs" = 0 ignores the link: include/xdemhip_test.h.) */
int xdemhip_pairs_link_sorted(xdemhip_pairs* pairs, xdemhip_pairs* sorted);
/* Round 5: float64 differences of float32 values (option "vario_diff" = 1 -- SciPy's pdist widens -- on float32 inputs) at the speed
 * of the float32 kernels.  `shadow` is a float32 set of the same blocks and edges as the float64 set `pairs` (whose values are all
 * exactly float32 numbers, widened): xdemhip_pairs_medians(pairs) then classifies every pair by its float32 difference on the
 * shadow (rounding is monotone, so the integer counts hold for the exact differences too), stages the candidates as exact float64
 * differences and selects among them in float64 -- the float64 medians bit for bit, the counting pass in 38 instead of 60 ms on
 * BASELINE's C5.  NULL removes the link; the shadow must outlive it (not owned). */
int xdemhip_pairs_link_shadow(xdemhip_pairs* pairs, xdemhip_pairs* shadow);
/* Whether exact medians on this pair set take the bracketed route (one counting pass against sample brackets: large sets, few
 * enough lag classes, the selection mode that allows it, no reduction hook) -- the only route that reads a float32 shadow, so a
 * caller builds one (two more uploads and their device memory) exactly when this says 1. */
int xdemhip_pairs_takes_brackets(xdemhip_pairs* pairs, int* yes);
void xdemhip_pairs_destroy(xdemhip_pairs* pairs);

/* ---- next row 8f-3: N-dimensional binned statistics ---------------------------------------------------------------
 * Replaces the array work of  nd_binning(values, list_var, list_var_names, list_var_bins, statistics, list_ranges)
 * xdem/spatialstats.py:91-216, i.e. scipy.stats.binned_statistic / binned_statistic_2d / binned_statistic_dd with the
 * statistics "count", np.nanmedian and nmad (1.4826 * nanmedian(|x - nanmedian(x)|), geoutils.stats.nmad) -- the binning
 * behind the heteroscedasticity inference (spatialstats.py:576-631).
 *
 *  create / add_var   values and explanatory variables, all of length n (float32 or float64 each).
 *  finalize           joint finiteness filter (spatialstats.py:140-143) -> n_valid, and every variable's min / max over
 *                     the kept rows (what SciPy's _bin_edges takes its range from).
 *  run                one binning over n_dims of the variables.  edges = the dimensions' bin edges concatenated
 *                     (n_edges[d] each, float64 numbers already rounded to SciPy's edge dtype); decimals[d] = SciPy's
 *                     `int(-log10(min edge spacing)) + 6` and sample_dtype = dtype of its sample matrix (both only matter
 *                     for samples at or beyond the rightmost edge, _binned_statistic.py:_bin_numbers).  Outputs are in C
 *                     order over the dimensions: counts (exact), medians (exact order statistics, mean of the two
 *                     middle values in the value dtype for even counts, NaN for empty bins), nmads (want_nmad != 0).
 */
typedef struct xdemhip_binstats xdemhip_binstats;
int xdemhip_binstats_create(xdemhip_ctx* ctx, const void* values, int dtype, int64_t n, int memspace, xdemhip_binstats** out);
int xdemhip_binstats_add_var(xdemhip_binstats* plan, const void* var, int dtype, int memspace); /* returns the variable id */
int xdemhip_binstats_finalize(xdemhip_binstats* plan, int64_t* n_valid, double* var_min, double* var_max);
int xdemhip_binstats_run(xdemhip_binstats* plan, int n_dims, const int* var_ids, const double* edges, const int* n_edges,
                         const int* decimals, int sample_dtype, int want_nmad, double nfact, int64_t* counts,
                         double* medians, double* nmads);
/* The flat bin number (C order over the dimensions of the LAST xdemhip_binstats_run) of each of the n samples, 0xFFFF for a sample
 * the joint finiteness filter dropped or that lies in no bin: what a binding needs to evaluate a statistic the device does not know
 * -- any Python callable nd_binning is given (xdem/spatialstats.py:143-157 hands it to scipy.stats.binned_statistic, which applies
 * it to the values of every bin in sample order) -- on the host.  bins_out: host array of n uint16. */
int xdemhip_binstats_bin_numbers(xdemhip_binstats* plan, uint16_t* bins_out);
void xdemhip_binstats_destroy(xdemhip_binstats* plan);

/* Global NMAD of an array, NaN-skipping: median = np.nanmedian(v), nmad = nfact * np.nanmedian(|v - median|) in the value
 * dtype; values with |v| > abs_limit are dropped first (two_step_standardization's outlier filter, spatialstats.py:556-561;
 * pass +inf to keep everything). */
int xdemhip_nmad(xdemhip_ctx* ctx, const void* values, int dtype, int64_t n, double nfact, double abs_limit, int memspace,
                 double* median, double* nmad, int64_t* count);

/* out[p] = scale * f(vars[0][p], ..., vars[n_dims-1][p]) for the multilinear interpolant f on a regular grid: the evaluation
 * scipy.interpolate.RegularGridInterpolator(axes, grid_values, method="linear", bounds_error=False, fill_value=None) performs
 * for the error function returned by interp_nd_binning (xdem/spatialstats.py:417-421) and applied to whole rasters by
 * infer_heteroscedasticity_from_stable (spatialstats.py:866-868).  axes = grid axes concatenated (n_axis[d] points each,
 * ascending), grid_values in C order, NaN in any coordinate -> NaN, coordinates outside the grid extrapolate linearly. */
int xdemhip_interp_grid_linear(xdemhip_ctx* ctx, int n_dims, const double* axes, const int* n_axis, const double* grid_values,
                               const void* const* vars, const int* var_dtypes, int64_t n, double scale, double* out, int memspace);

/* ---- caller of path 3: double sum of spatially correlated errors ------------------------------------------------------
 * out = sum_i sum_j  ae[i] * be[j] * rho(|a_i - b_j|),  rho(h) = 1 - sum_m gamma_m(h) / sum_m psill_m: the O(N^2) part of
 *   neff_exact(coords, errors, params_variogram_model)            xdem/spatialstats.py:2175-2236  (bx = NULL: B = A, every ordered
 *                                                                  pair including i = j)
 *   neff_hugonnet_approx(coords, errors, params, subsample, ...)  xdem/spatialstats.py:2239-2308  (B = the random subset)
 * model_type: 0 spherical, 1 exponential, 2 gaussian, 3 cubic, 4 stable (smooth[m] used); scikit-gstat's effective-range forms.
 * Coordinates and errors float64. */
int xdemhip_cov_double_sum(xdemhip_ctx* ctx, const double* ax, const double* ay, const double* ae, int64_t na, const double* bx,
                           const double* by, const double* be, int64_t nb, int n_models, const int* model_type, const double* range,
                           const double* psill, const double* smooth, double* out_sum, int memspace);

#ifdef __cplusplus
}
#endif
#endif /* XDEMHIP_H */
