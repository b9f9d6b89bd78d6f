"""Does GPU idle time before the launches decide their speed (clock / memory-clock ramp)?  (measurement tool)
Planes and DEM allocated once; then, three times: idle for `gap` seconds, 40 launches timed one by one."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from xdem_amd import _lib, terrain
from xdem_amd.synth import fbm_torch

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
n = 40000
ctx = _lib.default_context(0)
dem = fbm_torch(n, n, "cuda", seed=42)
out = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for gap in (0.0, 3.0, 0.3, 3.0, 10.0):
    time.sleep(gap)
    ms = []
    for i in range(40):
        terrain.terrain_attributes_device(dem, FULL, out=out, resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)
        ms.append(ctx.last_kernel_ms())
    print(f"idle {gap:4.1f} s -> " + " ".join(f"{m:5.2f}" for m in ms), flush=True)
