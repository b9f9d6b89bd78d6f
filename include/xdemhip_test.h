/* xdemhip_test.h -- switches between internal routes of libxdemhip.so that must AGREE (round 6: split from the options of
 * include/xdemhip.h).  Not part of the drop-in boundary: nothing a caller of the library needs; the route-agreement tests
 * (tests/test_*_gpu.py) and the measurement tools (tools/) set them.  Every value of every switch returns the same planes /
 * integers / medians -- that is what the tests assert.
 *   "terrain_stream"     1 (default) streaming strips over the raster interior + tiles over its frame where the raster qualifies,
 *                        0 tiles only; 2 / 3 = one of the two launches only (debug); 128 / 256 / 512 = band height.
 *   "terrain_order"      0 (default) one band of strip groups per XCD, 1 natural order, 2 permuted, 3 column-major (measurement forms).
 *   "terrain_ring_wait"  0 (default) counted s_waitcnt for the LDS-DMA ring of the streaming kernels, 1 vmcnt(0): the check of the count.
 *   "terrain_window_lds" 1 (default) LDS-tiled kernel for windowed indexes of window sizes other than 3, 0 the per-pixel kernel.
 *   "nk_narrow"          -1 (default) sample brackets of the one-pass step narrowed by the measured rank offsets, 0 / 1 / 2 fixed.
 *   "vario_grid"         1 (default) raster-sampled points run the integer-lattice pair kernels, 0 always the float64-coordinate ones.
 *   "vario_runs"         1 (default) run-length counting pass of the exact-Dowd route on the Morton-ordered copy, 0 per-pair counters.
 *   "vario_sort"         1 (default) the host side uploads a Morton-ordered copy of every pair block, 0 one copy in the caller's order.
 * Measurement builds (-DXD_EXPERIMENT, csrc/Makefile: libxdemhip_exp*.so) add "terrain_store", "terrain_rows", "terrain_sync",
 * "vario_deff"; the product library refuses those names. */
#pragma once
#include "xdemhip.h"
#ifdef __cplusplus
extern "C" {
#endif
int xdemhip_set_test_switch(xdemhip_ctx* ctx, const char* name, int value);
#ifdef __cplusplus
}
#endif
