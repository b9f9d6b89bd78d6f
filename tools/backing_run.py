"""One backing of the resident planes under the headline launch (measurement tool; the command rocprofv3 wraps in
tools/backing_pmc.py):  python tools/backing_run.py <scattered|contiguous|torch|chunked> [launches] [terrain_order]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler

faulthandler.dump_traceback_later(170, exit=True)
import numpy as np
import torch

from xdem_amd import _lib, terrain

if os.environ.get("XD_LIB"):   # A/B of library builds across processes (measurement variants: xdem_amd/csrc/Makefile)
    _lib.LIB_PATH = os.environ["XD_LIB"]
from xdem_amd.synth import fbm_torch

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
backing = sys.argv[1]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 5
order = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = int(os.environ.get("BACKING_N", "40000"))
ctx = _lib.default_context(0)
ctx.set_option("terrain_order", order)
dem = fbm_torch(n, n, "cuda", seed=42)
out = terrain.alloc_planes(11, n, n, torch.float32, ctx, backing=backing)
t = []
for i in range(launches):
    terrain.terrain_attributes_device(dem, FULL, out=out, resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)
    t.append(ctx.last_kernel_ms())
print(f"lib {os.path.basename(_lib.LIB_PATH)} backing {backing} order {order}: launches {['%.2f' % x for x in t]} median {np.median(t[1:]):.3f} ms", flush=True)
