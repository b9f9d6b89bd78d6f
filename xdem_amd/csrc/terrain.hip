// terrain.hip -- dtype dispatch of the fused terrain-attribute kernel (kernel templates: terrain_tile.h).
#include "common.h"
#include "terrain_math.h"

namespace xd {

int launch_typed_ff(xdemhip_ctx* ctx, const TerrainLaunch& L);
int launch_typed_dd(xdemhip_ctx* ctx, const TerrainLaunch& L);
int launch_typed_fd(xdemhip_ctx* ctx, const TerrainLaunch& L);
int launch_typed_df(xdemhip_ctx* ctx, const TerrainLaunch& L);

static int launch_core(xdemhip_ctx* ctx, const TerrainLaunch& L) {
    if (L.dem_dtype == XDEMHIP_F32 && L.out_dtype == XDEMHIP_F32) return launch_typed_ff(ctx, L);
    if (L.dem_dtype == XDEMHIP_F64 && L.out_dtype == XDEMHIP_F64) return launch_typed_dd(ctx, L);
    if (L.dem_dtype == XDEMHIP_F32 && L.out_dtype == XDEMHIP_F64) return launch_typed_fd(ctx, L);
    if (L.dem_dtype == XDEMHIP_F64 && L.out_dtype == XDEMHIP_F32) return launch_typed_df(ctx, L);
    return xd_fail(ctx, XDEMHIP_EINVAL, "unsupported dtype combination");
}

int launch_terrain(xdemhip_ctx* ctx, const TerrainLaunch& L) {
    const uint32_t core = L.attr_mask & ((1u << N_ATTR) - 1u);  // attributes of the fused tile / generic window kernels
    if (core) {
        TerrainLaunch C = L;
        C.attr_mask = core;
        int rc = launch_core(ctx, C);
        if (rc != XDEMHIP_OK) return rc;
        if (ctx->terrain_nonfinite && (core & 0x3ffu)) {  // the Numba engine's +-Inf rule on top of the fused kernels' planes
            rc = launch_terrain_nonfinite(ctx, C);
            if (rc != XDEMHIP_OK) return rc;
        }
    }
    if (L.attr_mask & ~((1u << N_ATTR) - 1u)) return launch_window_extra(ctx, L);  // rugosity, fractal roughness
    return XDEMHIP_OK;
}

}  // namespace xd
