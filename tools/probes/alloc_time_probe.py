"""Does the slow mode of the headline launch wear off with time after the planes were allocated?  (measurement tool)
One fresh process, torch.empty planes (or the library's contiguous allocation with argument `contig`), 400 launches: elapsed time
since the allocation and the kernel time of every 10th."""
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
torch.cuda.synchronize()
t0 = time.perf_counter()
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "contig":
    out = terrain.alloc_planes(11, n, n, torch.float32, ctx, recycled=False)
elif mode == "recycled":
    out = terrain.alloc_planes(11, n, n, torch.float32, ctx, recycled=True)
elif mode == "recycled_plain":
    out = ctx.device_tensor((11, n, n), "float32", contiguous=False, recycled=True)
else:
    out = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
line = []
for i in range(int(os.environ.get("PROBE_LAUNCHES", "60"))):
    terrain.terrain_attributes_device(dem, FULL, out=out, resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)
    ms = ctx.last_kernel_ms()
    if i % 10 == 0:
        line.append(f"{time.perf_counter() - t0:5.2f}s:{ms:6.2f}")
print(f"{mode:15s}", " ".join(line), flush=True)
