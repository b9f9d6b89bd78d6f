"""Is the slow mode of the headline launch a matter of how the output planes are backed physically?  (measurement tool)
The same launch over (a) torch.empty planes, (b) ONE physically contiguous allocation (xdemhip_device_alloc), (c) eleven contiguous
planes, (d) a contiguous DEM as well -- alternating, several rounds, in one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from xdem_amd import _lib
from xdem_amd.synth import fbm_torch
from xdem_amd.terrain import launch_terrain, terrain_attributes_device

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _lib.default_context(0)
dem = fbm_torch(n, n, "cuda", seed=42)
torch.cuda.empty_cache()


def timed(dem_t, planes, reps=5):
    ptrs = {a: p.data_ptr() for a, p in zip(FULL, planes)}
    t = []
    for i in range(reps + 2):
        launch_terrain(ctx, dem_t.data_ptr(), np.float32, n, n, n, 0, 0, 10.0, "Florinsky", "geometric", FULL, "Riley", 3, 45.0,
                       315.0, 1.0, True, np.float32, ptrs, 1)
        ctx.synchronize()
        if i >= 2:
            t.append(ctx.last_kernel_ms())
    return float(np.median(t))


ref = None
for r in range(rounds):
    out = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
    ms = timed(dem, list(out))
    if ref is None:
        ref = out[:, :2048, :2048].clone()
    print(f"round {r}  torch.empty planes              {ms:7.3f} ms", flush=True)
    del out
    torch.cuda.empty_cache()
    try:
        one = ctx.device_tensor((11, n, n), "float32", contiguous=True)
        ms = timed(dem, list(one))
        same = bool(torch.equal(one[:, :2048, :2048].view(torch.int32), ref.view(torch.int32)))
        print(f"round {r}  one allocation, contiguous={one.xdem_contiguous!s:5s}  {ms:7.3f} ms   identical planes: {same}", flush=True)
        del one
    except Exception as e:
        print("one contiguous allocation failed:", e, flush=True)
    planes = [ctx.device_tensor((n, n), "float32", contiguous=True) for _ in FULL]
    ms = timed(dem, planes)
    print(f"round {r}  eleven planes, contiguous={all(p.xdem_contiguous for p in planes)!s:5s}   {ms:7.3f} ms", flush=True)
    demc = ctx.device_tensor((n, n), "float32", contiguous=True)
    demc.copy_(dem)
    ms = timed(demc, planes)
    print(f"round {r}  + contiguous DEM ({demc.xdem_contiguous})            {ms:7.3f} ms", flush=True)
    del planes, demc
