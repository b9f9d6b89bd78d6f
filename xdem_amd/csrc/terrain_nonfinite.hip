// terrain_nonfinite.hip -- the Numba engine's treatment of +-Inf pixels (context option "terrain_nonfinite" = 1).
//
// The reference's two engines disagree around infinite pixels.  The SciPy engine blanks every output whose full window holds
// a non-finite value (binary dilation of ~isfinite, xdem/terrain/surfit.py:1185-1192).  The Numba engine has no such mask
// (surfit.py:1270-1303): it pads the DEM with NaN and lets IEEE arithmetic decide -- its per-pixel loop (surfit.py:948-971)
// adds value x weight for EVERY tap, zero weights included, in float64, and the attribute formulas (surfit.py:451-945) run on
// whatever comes out.  NaN in the window -> NaN everywhere, as in the SciPy engine; but a window that holds +-Inf and no NaN
// gives 0 x Inf = NaN only under a zero weight, Inf - Inf = NaN only where two infinite taps meet with opposite signs, and
// otherwise an infinite derivative: slope 90 deg, an aspect that is a multiple of 45 deg, hillshade 1.5 / 181.1 / 0,
// `curvature` -+Inf (pinned by tests/golden/terrain_T11_numba_engine.npz, outputs of the reference's own numba-engine code).
//
// The fused kernels implement the window rule.  Under option "terrain_nonfinite" = 1 (what engine="numba" sets on the Python
// side) two small kernels run after them: `nf_flag_kernel` looks for an infinite pixel in the rows at hand (one read of the
// DEM), and `nf_fix_kernel` -- which leaves at once when there is none, the flag never visits the host -- re-evaluates the
// pixels whose window holds an infinite value and no NaN exactly as the Numba engine's loop does: the reference's own double
// weights (fill_ref_weights), one non-contracted multiply and add per tap in row-major window order, the formulas of
// surfit.py in float64 with the library's elementary functions, one rounding to the output dtype, rad2deg / clip in the output
// dtype (xdem/terrain/terrain.py:586-596).  Everything about it is cold: a raster without +-Inf pays one streaming read.
#include "common.h"
#include "terrain_nonfinite.h"

namespace xd {

template <typename TIN>
__global__ void nf_flag_kernel(const TIN* __restrict__ dem, int64_t rows, int64_t W, int64_t stride, int* flag) {
    bool any = false;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        const TIN* row = dem + r * stride;
        for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < W; c += (int64_t)gridDim.x * blockDim.x)
            any |= (bool)isinf((double)row[c]);
    }
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

template <typename TIN, typename TOUT>
__global__ void nf_fix_kernel(const TIN* __restrict__ dem, int64_t H, int64_t W, int64_t stride, int64_t halo_top,
                              int64_t halo_bottom, NfParams P, NfPlanes<TOUT> out, const int* __restrict__ flag) {
    if (*flag == 0) return;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    for (int64_t r = blockIdx.y; r < H; r += gridDim.y) nf_pixel<TIN, TOUT>(dem, r, c, H, W, stride, halo_top, halo_bottom, P, out);
}

template <typename TIN, typename TOUT>
static int nf_launch(xdemhip_ctx* ctx, const TerrainLaunch& L, const NfParams& P) {
    NfPlanes<TOUT> out;
    for (int k = 0; k < 10; ++k) out.p[k] = static_cast<TOUT*>(L.planes[k]);
    const TIN* dem = static_cast<const TIN*>(L.dem);
    const int64_t rows_total = L.halo_top + L.H + L.halo_bottom;
    XD_HIP_CHECK(ctx, hipMemsetAsync(ctx->nf_flag, 0, sizeof(int), ctx->stream));
    const unsigned gx = (unsigned)((L.W + 255) / 256);
    hipLaunchKernelGGL((nf_flag_kernel<TIN>), dim3(gx > 64 ? 64 : gx, (unsigned)(rows_total < 1024 ? rows_total : 1024)), dim3(256), 0,
                       ctx->stream, dem, rows_total, L.W, L.row_stride, ctx->nf_flag);
    XD_HIP_CHECK(ctx, hipGetLastError());
    hipLaunchKernelGGL((nf_fix_kernel<TIN, TOUT>), dim3(gx, (unsigned)(L.H < 1024 ? L.H : 1024)), dim3(256), 0, ctx->stream, dem, L.H,
                       L.W, L.row_stride, L.halo_top, L.halo_bottom, P, out, ctx->nf_flag);
    XD_HIP_CHECK(ctx, hipGetLastError());
    return XDEMHIP_OK;
}

int launch_terrain_nonfinite(xdemhip_ctx* ctx, const TerrainLaunch& L) {
    const uint32_t surf = L.attr_mask & 0x3ffu;
    if (!surf) return XDEMHIP_OK;
    if (!ctx->nf_flag) XD_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->nf_flag), sizeof(int)));
    NfParams P;
    nf_fill_params(P, L.surface_fit, L.curv_method == XDEMHIP_CURV_DIRECTIONAL, L.resolution, L.hs_alt, L.hs_az, L.hs_z, L.degrees, surf, L.hs_unclipped ? 0 : 1);
    if (L.dem_dtype == XDEMHIP_F32 && L.out_dtype == XDEMHIP_F32) return nf_launch<float, float>(ctx, L, P);
    if (L.dem_dtype == XDEMHIP_F64 && L.out_dtype == XDEMHIP_F64) return nf_launch<double, double>(ctx, L, P);
    if (L.dem_dtype == XDEMHIP_F32 && L.out_dtype == XDEMHIP_F64) return nf_launch<float, double>(ctx, L, P);
    return nf_launch<double, float>(ctx, L, P);
}

}  // namespace xd
