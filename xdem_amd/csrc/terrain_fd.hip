// terrain_fd.hip -- the fused terrain kernel for float DEMs and double attribute planes (see terrain_tile.h).
#include "terrain_tile.h"

namespace xd {
int launch_typed_fd(xdemhip_ctx* ctx, const TerrainLaunch& L) { return launch_typed<float, double>(ctx, L); }
}  // namespace xd
