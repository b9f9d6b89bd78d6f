"""Does the placement of the 70 GB output tensor decide the terrain kernel's speed?  (measurement tool)
For several fresh allocations of the (11, n, n) planes -- plain, base aligned to 1 GiB, planes padded to 2 MiB multiples -- time
the headline launch and print the base address alignment next to it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from xdem_amd import _lib
from xdem_amd.synth import fbm_torch
from xdem_amd.terrain import terrain_attributes_device

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
ctx = _lib.default_context(0)
dem = fbm_torch(n, n, "cuda", seed=42)
kw = dict(resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)


def timed(out):
    for _ in range(3):
        terrain_attributes_device(dem, FULL, out=out, **kw)
    t = []
    for _ in range(6):
        terrain_attributes_device(dem, FULL, out=out, **kw)
        t.append(ctx.last_kernel_ms())
    t.sort()
    return t[0], t[len(t) // 2]


plane = n * n
for trial in range(8):
    mode = ["plain", "plain", "1GiB", "plain", "2MiB planes", "1GiB", "plain", "hog+plain"][trial]
    hog = None
    if mode == "hog+plain":
        hog = torch.empty(3 * (1 << 30) + 12345 * 4, dtype=torch.uint8, device="cuda")   # disturb the allocator's alignment
    if mode == "1GiB":
        raw = torch.empty(11 * plane + (1 << 28), dtype=torch.float32, device="cuda")
        off = (-raw.data_ptr()) % (1 << 30) // 4
        out = raw[off:off + 11 * plane].view(11, n, n)
    elif mode == "2MiB planes":
        pad = (-plane * 4) % (1 << 21) // 4
        raw = torch.empty(11 * (plane + pad) + (1 << 19), dtype=torch.float32, device="cuda")
        off = (-raw.data_ptr()) % (1 << 21) // 4
        out = raw[off:off + 11 * (plane + pad)].view(11, plane + pad)[:, :plane].view(11, n, n)
    else:
        raw = torch.empty((11, n, n), dtype=torch.float32, device="cuda")
        out = raw
    assert out.stride(1) == n and out.stride(2) == 1
    mn, med = timed(out)
    print(f"trial {trial} {mode:12s} base % 1GiB = {out.data_ptr() % (1 << 30):>11d}  % 2MiB = {out.data_ptr() % (1 << 21):>8d}  plane stride {out.stride(0) * 4} "
          f"min {mn:7.3f} ms  median {med:7.3f} ms", flush=True)
    del out, raw, hog
    torch.cuda.empty_cache()
