"""Second placement probe (measurement tool): is the headline launch's bimodal duration a property of the ALLOCATION that holds
the output planes?  (a) plain allocations in a row, freed in between; (b) two allocations alive at once, timed A B A B;
(c) eleven separate plane allocations; (d) the "keep the faster of two" rule of xdem_amd.terrain.alloc_planes."""
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
ctx = _lib.default_context(0)
dem = fbm_torch(n, n, "cuda", seed=42)
kw = dict(resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)


def timed(out, reps=5):
    for _ in range(2):
        terrain_attributes_device(dem, FULL, out=out, **kw)
    t = []
    for _ in range(reps):
        terrain_attributes_device(dem, FULL, out=out, **kw)
        t.append(ctx.last_kernel_ms())
    return float(np.median(t))


def timed_planes(planes, reps=5):
    ptrs = {a: p.data_ptr() for a, p in zip(FULL, planes)}
    t = []
    for i in range(reps + 2):
        launch_terrain(ctx, dem.data_ptr(), np.float32, n, n, n, 0, 0, 10.0, "Florinsky", "geometric", FULL, "Riley", 3, 45.0,
                       315.0, 1.0, True, np.float32, ptrs, 1)
        torch.cuda.synchronize()
        if i >= 2:
            t.append(ctx.last_kernel_ms())
    return float(np.median(t))


print("(a) plain allocations, freed in between", flush=True)
for trial in range(8):
    out = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
    print(f"  a{trial}  base {out.data_ptr():#x}  {timed(out):7.3f} ms", flush=True)
    del out
    torch.cuda.empty_cache()

print("(b) two allocations alive, timed alternately", flush=True)
A = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
B = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
for k in range(3):
    print(f"  A {timed(A):7.3f} ms   B {timed(B):7.3f} ms", flush=True)
C = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
print(f"  C (third alive) {timed(C):7.3f} ms", flush=True)
del A, B, C
torch.cuda.empty_cache()

print("(c) eleven separate plane allocations", flush=True)
for trial in range(4):
    planes = [torch.empty((n, n), dtype=torch.float32, device="cuda") for _ in FULL]
    print(f"  c{trial}  {timed_planes(planes):7.3f} ms", flush=True)
    del planes
    torch.cuda.empty_cache()

print("(d) read side: fresh DEM allocations against one output allocation", flush=True)
out = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
base = timed(out)
for trial in range(4):
    dem2 = dem.clone()
    keep, dem = dem, dem2
    print(f"  d{trial}  out fixed ({base:7.3f} ms with the first DEM)  new DEM copy: {timed(out):7.3f} ms", flush=True)
    dem = keep
    del dem2
    torch.cuda.empty_cache()
