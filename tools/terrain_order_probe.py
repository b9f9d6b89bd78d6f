"""Strip-to-workgroup orders of the streaming terrain kernel against the physical backing of the planes, A/B inside one
process (measurement tool, round 4): does a kernel-side de-correlation of the row streams in flight (order 2: permuted strip
groups, order 3: column-major) bring contiguous planes to the rate of the scattered backing?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler

faulthandler.dump_traceback_later(280, exit=True)
import numpy as np
import torch

from xdem_amd import _lib, terrain
from xdem_amd.synth import fbm_torch

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
n = 40000
ctx = _lib.default_context(0)
dem = fbm_torch(n, n, "cuda", seed=42)


def timed(out, order, stream=1):
    ctx.set_option("terrain_order", order)
    ctx.set_option("terrain_stream", stream)
    t = []
    for i in range(8):
        terrain.terrain_attributes_device(dem, FULL, out=out, resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)
        if i >= 2:
            t.append(ctx.last_kernel_ms())
    ctx.set_option("terrain_order", 0)
    ctx.set_option("terrain_stream", 1)
    return float(np.median(t))


for backing in ("contiguous", "scattered", "torch", "contiguous"):
    t0 = time.perf_counter()
    out = terrain.alloc_planes(11, n, n, torch.float32, ctx, backing=backing)
    ta = time.perf_counter() - t0
    line = f"{backing:10s} (alloc {ta:4.1f} s)"
    for order in (0, 1, 2, 3):
        line += f"  order {order}: {timed(out, order):6.2f}"
    line += f"  order 0 / 256-row bands: {timed(out, 0, 256):6.2f}"
    print(line, flush=True)
    del out
    torch.cuda.empty_cache()
