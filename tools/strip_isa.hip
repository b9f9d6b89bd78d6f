// tools/strip_isa.hip -- one instantiation of the streaming terrain kernel for ISA work (measurement tool, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -DXD_MINW=4 -Ixdem_amd/csrc -Iinclude tools/strip_isa.hip -o /tmp/strip.s
//   python tools/isa_stats.py /tmp/strip.s ; python tools/isa_ledger.py /tmp/strip.s
#include "terrain_tile.h"
#ifndef XD_MINW
#define XD_MINW 1
#endif
#ifndef XD_FIT
#define XD_FIT 2
#endif
#ifndef XD_DIR
#define XD_DIR 0
#endif
template __global__ void xd::terrain_strip_kernel<XD_FIT, true, true, xd::Spec<xd::MASK_FULL11, XD_DIR, 1, 0, 1, 2>, 128, XD_MINW>(const xd::StripArgs);
