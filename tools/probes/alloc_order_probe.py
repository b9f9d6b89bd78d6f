"""Does the ORDER of the two big allocations decide which of the two speeds a process gets?  (measurement tool)
  python tools/probes/alloc_order_probe.py out_first|dem_first|dem_first_empty"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from xdem_amd import _lib
from xdem_amd.synth import fbm_torch
from xdem_amd.terrain import terrain_attributes_device

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
mode = sys.argv[1]
n = 40000
ctx = _lib.default_context(0)
if mode == "out_first":
    out = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
    dem = fbm_torch(n, n, "cuda", seed=42)
else:
    dem = fbm_torch(n, n, "cuda", seed=42)
    if mode == "dem_first_empty":
        torch.cuda.empty_cache()
    out = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
t = []
for i in range(8):
    terrain_attributes_device(dem, FULL, out=out, resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)
    if i >= 2:
        t.append(ctx.last_kernel_ms())
print(f"{mode:16s} out {out.data_ptr():#x} dem {dem.data_ptr():#x}  median {np.median(t):7.3f} ms", flush=True)
