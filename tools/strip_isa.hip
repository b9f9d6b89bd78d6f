// tools/strip_isa.hip -- one instantiation of the streaming terrain kernel for ISA work (measurement tool, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -DXD_MINW=4 -Ixdem_amd/csrc -Iinclude tools/strip_isa.hip -o /tmp/strip.s
//   python tools/isa_stats.py /tmp/strip.s ; python tools/isa_ledger.py      (-DXD_MASK=<attribute bits>: another compile-time set)
#include "terrain_tile.h"
#ifndef XD_MINW
#define XD_MINW 4
#endif
#ifndef XD_FIT
#define XD_FIT 2
#endif
#ifndef XD_DIR
#define XD_DIR 0
#endif
#ifndef XD_MASK
#define XD_MASK 4087u
#endif
template __global__ void xd::terrain_strip_kernel<XD_FIT, ((XD_MASK) & xd::A_ANY_CURV) != 0, ((XD_MASK) & xd::A_ANY_WIN) != 0,
                                                  xd::Spec<(XD_MASK), XD_DIR, 1, 0, 1, 2>, 128, XD_MINW, XD_RING_BLOCK(__builtin_popcount(XD_MASK))>(const xd::StripArgs);
