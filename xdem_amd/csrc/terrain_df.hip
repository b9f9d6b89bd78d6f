// terrain_df.hip -- the fused terrain kernel for double DEMs and float attribute planes (see terrain_tile.h).
#include "terrain_tile.h"

namespace xd {
int launch_typed_df(xdemhip_ctx* ctx, const TerrainLaunch& L) { return launch_typed<double, float>(ctx, L); }
}  // namespace xd
