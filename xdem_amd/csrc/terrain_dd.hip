// terrain_dd.hip -- the fused terrain kernel for double DEMs and double attribute planes (see terrain_tile.h).
#include "terrain_tile.h"

namespace xd {
int launch_typed_dd(xdemhip_ctx* ctx, const TerrainLaunch& L) { return launch_typed<double, double>(ctx, L); }
}  // namespace xd
