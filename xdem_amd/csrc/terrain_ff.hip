// terrain_ff.hip -- the fused terrain kernel for float DEMs and float attribute planes (see terrain_tile.h).
#include "terrain_tile.h"

namespace xd {
int launch_typed_ff(xdemhip_ctx* ctx, const TerrainLaunch& L) { return launch_typed<float, float>(ctx, L); }
}  // namespace xd
