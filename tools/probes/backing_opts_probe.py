"""Which launch options are sensitive to the physical backing of the planes?  (measurement tool)
Contiguous planes (slow-prone) and torch.empty planes, each under: default, linear tile order, longer strips, tile kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from xdem_amd import _lib, terrain
from xdem_amd.synth import fbm_torch

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
n = 40000
ctx = _lib.default_context(0)
dem = fbm_torch(n, n, "cuda", seed=42)
OPTS = [{}, {"terrain_stream": 0}]
DEFAULTS = {"terrain_order": 0, "terrain_stream": 1}


def timed(out, opts):
    for k, v in {**DEFAULTS, **opts}.items():
        ctx.set_option(k, v)
    t = []
    for i in range(7):
        terrain.terrain_attributes_device(dem, FULL, out=out, resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)
        if i >= 2:
            t.append(ctx.last_kernel_ms())
    for k, v in DEFAULTS.items():
        ctx.set_option(k, v)
    return float(np.median(t))


import time

for backing in ("default", "scattered", "contiguous", "scattered", "default"):
    t_alloc = time.perf_counter()
    out = terrain.alloc_planes(11, n, n, torch.float32, ctx, backing=backing)
    t_alloc = time.perf_counter() - t_alloc
    print(f"{backing:10s} (allocated in {t_alloc:5.1f} s) " + "  ".join(f"{str(o) if o else 'default'}: {timed(out, o):6.2f}" for o in OPTS), flush=True)
    del out
    torch.cuda.empty_cache()
