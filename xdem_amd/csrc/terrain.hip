// terrain.hip -- fused terrain-attribute kernel for gfx950 (MI355X).
//
// One pass over the DEM produces every requested attribute (slope, aspect, hillshade, the curvatures,
// TPI, TRI): 4 B read + 4 B written per attribute per pixel -- an HBM-bound job, no MFMA (no dense
// contraction anywhere).  Layout of one workgroup (256 threads = 4 wave64):
//
//   * tile = 256 columns x TH rows of output; the DEM patch with its halo (TH + 2*HALO rows, 264 columns)
//     is staged in LDS with 16-byte coalesced row-major global loads (one float4 per lane, 1 KiB per wave
//     instruction); pixels outside the raster become NaN while staging, so the stencil needs no edge code;
//   * each thread owns ONE column and marches down the tile rows (terrain_math.h: march_column) holding a
//     rotating register window of per-row partial sums: LDS reads are lane-consecutive (conflict-free),
//     every store is a 256-byte contiguous row segment per wave and plane;
//   * tiles are handed to workgroups in an XCD-aware order: hardware round-robins consecutive workgroup
//     ids over the 8 XCDs, so logical tile = (id % 8) * (n/8) + id / 8 gives every XCD one contiguous band
//     of the raster and halo rows/columns shared by neighbouring tiles hit the same 4 MiB L2.
//
// Replaces xdem/terrain/surfit.py:1197-1305 (_get_surface_attributes), xdem/terrain/window.py:926-1002
// (_get_windowed_indexes, TPI/TRI) and the post-steps of xdem/terrain/terrain.py:586-596.
#include <stdlib.h>

#include "common.h"
#include "terrain_math.h"

namespace xd {

constexpr int TILE_W = 256;
constexpr int XPAD = 4;
constexpr int PITCH = TILE_W + 2 * XPAD;

template <typename TIN, typename TOUT> struct TileArgs {
    const TIN* dem;
    int64_t H, W, stride, halo_top, halo_bottom;
    int tiles_x, tiles_y, ntiles, grid8;  // grid8 = padded grid / 8
    int vec_ok;
    TerrainParams P;
    Planes<TOUT> out;
};

template <typename TIN> struct TileRows { static constexpr int v = (sizeof(TIN) == 4) ? 32 : 16; };

template <int FIT, bool CURV, bool WIN, class SP, typename TIN, typename TOUT, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void terrain_tile_kernel(const TileArgs<TIN, TOUT> a) {
    constexpr int HALO = Halo<FIT>::v;
    constexpr int TH = TileRows<TIN>::v;
    constexpr int VEC = 16 / sizeof(TIN);
    constexpr int NV = PITCH / VEC;
    __shared__ __attribute__((aligned(16))) TIN tile[(TH + 2 * HALO) * PITCH];

    // XCD-aware tile order (see file header)
    const int b = blockIdx.x;
    const int logical = (b & 7) * a.grid8 + (b >> 3);
    if (logical >= a.ntiles) return;
    const int ty = logical / a.tiles_x, tx = logical - ty * a.tiles_x;
    const int64_t x0 = (int64_t)tx * TILE_W, y0 = (int64_t)ty * TH;
    const int n_out = (int)((a.H - y0) < TH ? (a.H - y0) : TH);
    const int rows = n_out + 2 * HALO;
    const int tid = threadIdx.x;
    const TIN nan_in = (TIN)NAN;

    typedef TIN vec_t __attribute__((ext_vector_type(VEC)));
    for (int idx = tid; idx < rows * NV; idx += 256) {
        const int r = idx / NV, v = idx - r * NV;
        const int64_t gy = y0 - HALO + r;
        const int64_t gx = x0 - XPAD + (int64_t)v * VEC;
        const bool rowok = (gy >= -a.halo_top) && (gy < a.H + a.halo_bottom);
        const TIN* src = a.dem + (gy + a.halo_top) * a.stride + gx;
        vec_t val;
        if (rowok && a.vec_ok && gx >= 0 && gx + VEC <= a.W) {
            val = *reinterpret_cast<const vec_t*>(src);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int64_t x = gx + e;
                val[e] = (rowok && x >= 0 && x < a.W) ? src[e] : nan_in;
            }
        }
        *reinterpret_cast<vec_t*>(&tile[r * PITCH + v * VEC]) = val;
    }
    __syncthreads();

    if (x0 + tid < a.W) {
        // wave-uniform plane pointers at the tile origin (SGPR pairs) + 32-bit per-thread byte offsets
        // (readfirstlane pins the wave-uniform tile offset into SGPRs so the pointers below stay scalar)
        const uint64_t org_u = (uint64_t)(y0 * a.W + x0);
        // (the builtin returns a signed int: go through uint32_t, or a low half >= 2^31 sign-extends over the high half)
        const int64_t org_off = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(org_u >> 32)) << 32) |
                                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)org_u));
        Planes<TOUT> org;
#pragma unroll
        for (int k = 0; k < N_ATTR; ++k) org.p[k] = a.out.p[k] + org_off;
        march_column<FIT, CURV, WIN, SP, TIN, TOUT>(tile + XPAD + tid, PITCH, n_out, a.P, org,
                                                    (uint32_t)(tid * sizeof(TOUT)), (uint32_t)(a.W * sizeof(TOUT)));
    }
}

// TPI / TRI for an arbitrary odd window (reference default is 3, handled by the fused kernel above).
// One thread per pixel, window read through L1/L2; float64 accumulation in row-major order like the
// flattened footprint the reference's generic_filter callback sums (xdem/terrain/window.py:67-252).
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void window_generic_kernel(const TIN* dem, int64_t H, int64_t W, int64_t stride,
                                                              int64_t halo_top, int64_t halo_bottom, int w,
                                                              int tri_wilson, TOUT* tpi, TOUT* tri, TOUT* rough) {
    const int64_t x = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t y = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int h = w / 2;
    const double c = (double)dem[(y + halo_top) * stride + x];
    double sum = 0.0, acc = 0.0, mx = -INFINITY, mn = INFINITY;
    bool has_nan = false;
    for (int dy = -h; dy <= h; ++dy) {
        const int64_t yy = y + dy;
        const bool rowok = (yy >= -halo_top) && (yy < H + halo_bottom);
        for (int dx = -h; dx <= h; ++dx) {
            const int64_t xx = x + dx;
            const double v = (rowok && xx >= 0 && xx < W) ? (double)dem[(yy + halo_top) * stride + xx] : (double)NAN;
            sum += v;
            has_nan |= (v != v);
            mx = v > mx ? v : mx;
            mn = v < mn ? v : mn;
            const double d = fabs(v - c);
            acc = tri_wilson ? (acc + d) : fma(d, d, acc);
        }
    }
    const double nn = (double)(w * w - 1);
    const int64_t o = y * W + x;
    if (tpi) tpi[o] = (TOUT)(c - (sum - c) / nn);
    if (tri) tri[o] = (TOUT)(tri_wilson ? acc / nn : sqrt(acc));
    if (rough) rough[o] = has_nan ? (TOUT)NAN : (TOUT)(mx - mn);
}

static void fill_params(const TerrainLaunch& L, TerrainParams& P) {
    const double res = L.resolution;
    double c1 = 8.0, cxx = 1.0, cxy = 4.0;
    if (L.surface_fit == XDEMHIP_FIT_ZEVENBERGTHORNE) { c1 = 2.0; cxx = 1.0; cxy = 4.0; }
    if (L.surface_fit == XDEMHIP_FIT_FLORINSKY) { c1 = 420.0; cxx = 35.0; cxy = 100.0; }
    // same divider expressions as xdem/terrain/surfit.py:281-302, inverted once in double
    P.s1 = 1.0 / (c1 * res);
    P.sxx = 1.0 / (cxx * (res * res));
    P.sxy = 1.0 / (cxy * (res * res));
    const double deg = 0.017453292519943295;
    const double az = (360.0 - L.hs_az) * deg;  // np.deg2rad(360 - azimuth), surfit.py:614
    const double alt = L.hs_alt * deg;
    P.hs_sin_alt = 254.0 * sin(alt);
    P.hs_kx = 254.0 * (-cos(alt) * L.hs_z * cos(az));
    P.hs_ky = 254.0 * (cos(alt) * L.hs_z * sin(az));
    P.hs_zf2 = L.hs_z * L.hs_z;
    P.mask = L.attr_mask;
    P.curv_directional = L.curv_method == XDEMHIP_CURV_DIRECTIONAL;
    P.tri_wilson = L.tri_method == XDEMHIP_TRI_WILSON;
    P.degrees = L.degrees;
}

template <int FIT, bool CURV, bool WIN, class SP, typename TIN, typename TOUT, int MINW = 1>
static int launch_tiles(xdemhip_ctx* ctx, const TerrainLaunch& L, uint32_t mask) {
    TileArgs<TIN, TOUT> a;
    a.dem = static_cast<const TIN*>(L.dem);
    a.H = L.H; a.W = L.W; a.stride = L.row_stride; a.halo_top = L.halo_top; a.halo_bottom = L.halo_bottom;
    constexpr int TH = TileRows<TIN>::v;
    const int64_t tx = (L.W + TILE_W - 1) / TILE_W, ty = (L.H + TH - 1) / TH;
    // (a HIP dispatch carries the total work-item count in 32 bits: 2^24 tiles x 256 threads is the most one launch holds --
    // 1.4e11 pixels, beyond any device memory)
    if (tx * ty > (int64_t)0xfffff0) return xd_fail(ctx, XDEMHIP_EUNSUPPORTED, "raster too large for one launch");
    a.tiles_x = (int)tx; a.tiles_y = (int)ty; a.ntiles = (int)(tx * ty);
    a.grid8 = (a.ntiles + 7) / 8;
    constexpr int VEC = 16 / sizeof(TIN);
    a.vec_ok = ((reinterpret_cast<uintptr_t>(L.dem) & 15) == 0) && (L.row_stride % VEC == 0);
    fill_params(L, a.P);
    a.P.mask = mask;
    for (int k = 0; k < N_ATTR; ++k) a.out.p[k] = static_cast<TOUT*>(L.planes[k]);
    hipLaunchKernelGGL((terrain_tile_kernel<FIT, CURV, WIN, SP, TIN, TOUT, MINW>), dim3(a.grid8 * 8), dim3(256), 0, ctx->stream, a);
    XD_HIP_CHECK(ctx, hipGetLastError());
    return XDEMHIP_OK;
}

template <typename TIN, typename TOUT>
static int launch_typed(xdemhip_ctx* ctx, const TerrainLaunch& L) {
    const uint32_t surf = L.attr_mask & ~A_ANY_WIN;
    const uint32_t win = L.attr_mask & A_ANY_WIN;
    const bool fuse_win = win && L.window_size == 3;
    const uint32_t mask = surf | (fuse_win ? win : 0u);
    const bool curv = (surf & A_ANY_CURV) != 0;
    int rc = XDEMHIP_OK;
    if (mask) {
        const int fit = surf ? L.surface_fit : XDEMHIP_FIT_ZEVENBERGTHORNE;  // window-only: cheapest 3x3 march
        // Compile-time specialised kernels for the headline configurations (reference defaults: geometric
        // curvatures, degrees, Riley TRI, z_factor 1): all attribute branches fold away -> one schedulable basic block.
        const bool defaults = L.curv_method == XDEMHIP_CURV_GEOMETRIC && L.degrees && L.tri_method == XDEMHIP_TRI_RILEY &&
                              L.hs_z == 1.0;
        if (defaults && mask == MASK_FULL11 && fit == XDEMHIP_FIT_FLORINSKY) {
            static const int occ = getenv("XDEMHIP_TERRAIN_OCC") ? atoi(getenv("XDEMHIP_TERRAIN_OCC")) : 3;  // tuning knob
            if (occ == 4) return launch_tiles<2, true, true, Spec<MASK_FULL11, 0, 1, 0, 1>, TIN, TOUT, 4>(ctx, L, mask);
            return launch_tiles<2, true, true, Spec<MASK_FULL11, 0, 1, 0, 1>, TIN, TOUT>(ctx, L, mask);
        }
        if (defaults && mask == MASK_FULL11 && fit == XDEMHIP_FIT_ZEVENBERGTHORNE)
            return launch_tiles<1, true, true, Spec<MASK_FULL11, 0, 1, 0, 1>, TIN, TOUT>(ctx, L, mask);
        if (defaults && mask == MASK_SAH_WIN && fit == XDEMHIP_FIT_HORN)
            return launch_tiles<0, false, true, Spec<MASK_SAH_WIN, 0, 1, 0, 1>, TIN, TOUT>(ctx, L, mask);
#define XD_GO(F, C, Wn) rc = launch_tiles<F, C, Wn, SpecRuntime, TIN, TOUT>(ctx, L, mask)
        if (fit == XDEMHIP_FIT_HORN) { if (fuse_win) XD_GO(0, false, true); else XD_GO(0, false, false); }
        else if (fit == XDEMHIP_FIT_ZEVENBERGTHORNE) {
            if (curv) { if (fuse_win) XD_GO(1, true, true); else XD_GO(1, true, false); }
            else { if (fuse_win) XD_GO(1, false, true); else XD_GO(1, false, false); }
        } else {
            if (curv) { if (fuse_win) XD_GO(2, true, true); else XD_GO(2, true, false); }
            else { if (fuse_win) XD_GO(2, false, true); else XD_GO(2, false, false); }
        }
#undef XD_GO
        if (rc != XDEMHIP_OK) return rc;
    }
    if (win && !fuse_win) {
        dim3 grid((unsigned)((L.W + 63) / 64), (unsigned)((L.H + 3) / 4));
        hipLaunchKernelGGL((window_generic_kernel<TIN, TOUT>), grid, dim3(256), 0, ctx->stream,
                           static_cast<const TIN*>(L.dem), L.H, L.W, L.row_stride, L.halo_top, L.halo_bottom,
                           L.window_size, (int)(L.tri_method == XDEMHIP_TRI_WILSON),
                           (win & A_TPI) ? static_cast<TOUT*>(L.planes[P_TPI]) : nullptr,
                           (win & A_TRI) ? static_cast<TOUT*>(L.planes[P_TRI]) : nullptr,
                           (win & A_ROUGH) ? static_cast<TOUT*>(L.planes[P_ROUGH]) : nullptr);
        XD_HIP_CHECK(ctx, hipGetLastError());
    }
    return XDEMHIP_OK;
}

static int launch_core(xdemhip_ctx* ctx, const TerrainLaunch& L) {
    if (L.dem_dtype == XDEMHIP_F32 && L.out_dtype == XDEMHIP_F32) return launch_typed<float, float>(ctx, L);
    if (L.dem_dtype == XDEMHIP_F64 && L.out_dtype == XDEMHIP_F64) return launch_typed<double, double>(ctx, L);
    if (L.dem_dtype == XDEMHIP_F32 && L.out_dtype == XDEMHIP_F64) return launch_typed<float, double>(ctx, L);
    if (L.dem_dtype == XDEMHIP_F64 && L.out_dtype == XDEMHIP_F32) return launch_typed<double, float>(ctx, L);
    return xd_fail(ctx, XDEMHIP_EINVAL, "unsupported dtype combination");
}

int launch_terrain(xdemhip_ctx* ctx, const TerrainLaunch& L) {
    const uint32_t core = L.attr_mask & ((1u << N_ATTR) - 1u);  // attributes of the fused tile / generic window kernels
    if (core) {
        TerrainLaunch C = L;
        C.attr_mask = core;
        const int rc = launch_core(ctx, C);
        if (rc != XDEMHIP_OK) return rc;
    }
    if (L.attr_mask & ~((1u << N_ATTR) - 1u)) return launch_window_extra(ctx, L);  // rugosity, fractal roughness
    return XDEMHIP_OK;
}

}  // namespace xd
