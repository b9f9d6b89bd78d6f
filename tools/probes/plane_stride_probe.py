"""Physically contiguous planes and the plane-to-plane stride (measurement tool).  One contiguous allocation (the layout that ran
slow as a process's first allocation in 9 of 10 trials); the eleven planes carved out of it with different paddings between them:
does some stride take the launch back to the fast mode?  Then the same strides on torch.empty memory."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from xdem_amd import _lib
from xdem_amd.synth import fbm_torch
from xdem_amd.terrain import launch_terrain

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
n = 40000
plane = n * n
ctx = _lib.default_context(0)
pads = [0, 64, 1024, 1024 + 64, 17 * 1024, 256 * 1024 + 1024, 47 * 1024 + 192, 524288 - 64 * 37]   # floats between planes
maxpad = max(pads)
raw = ctx.device_tensor((11 * (plane + maxpad),), "float32", contiguous=True)
dem = fbm_torch(n, n, "cuda", seed=42)


def timed(base, pad, reps=5):
    planes = [base[k * (plane + pad): k * (plane + pad) + plane] for k in range(11)]
    ptrs = {a: p.data_ptr() for a, p in zip(FULL, planes)}
    t = []
    for i in range(reps + 3):
        launch_terrain(ctx, dem.data_ptr(), np.float32, n, n, n, 0, 0, 10.0, "Florinsky", "geometric", FULL, "Riley", 3, 45.0,
                       315.0, 1.0, True, np.float32, ptrs, 1)
        ctx.synchronize()
        if i >= 3:
            t.append(ctx.last_kernel_ms())
    return float(np.median(t))


pads = [0, 1024]
print(f"contiguous allocation: {raw.xdem_contiguous}", flush=True)
print("contiguous  " + "  ".join(f"pad {p * 4 // 1024:5d}K: {timed(raw, p):6.2f}" for p in pads), flush=True)
del raw
try:
    ch = ctx.device_tensor((11 * (plane + maxpad),), "float32", chunked=True)
    print("chunked 64M " + "  ".join(f"pad {p * 4 // 1024:5d}K: {timed(ch, p):6.2f}" for p in pads), flush=True)
    del ch
except Exception as e:
    print("chunked allocation failed:", e, flush=True)
raw = ctx.device_tensor((11 * (plane + maxpad),), "float32", contiguous=True)
print("contiguous  " + "  ".join(f"pad {p * 4 // 1024:5d}K: {timed(raw, p):6.2f}" for p in pads), flush=True)
del raw
plain = torch.empty(11 * (plane + maxpad), dtype=torch.float32, device="cuda")
print("torch.empty " + "  ".join(f"pad {p * 4 // 1024:5d}K: {timed(plain, p):6.2f}" for p in pads), flush=True)
del plain
torch.cuda.empty_cache()
ch = ctx.device_tensor((11 * (plane + maxpad),), "float32", chunked=True)
print("chunked 64M " + "  ".join(f"pad {p * 4 // 1024:5d}K: {timed(ch, p):6.2f}" for p in pads), flush=True)
