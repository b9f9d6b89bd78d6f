"""Kernel-time probe: HIP-event duration of the fused terrain kernel for several attribute subsets / fits.
Usage (GPU box): python tools/terrain_probe.py [size]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xdem_amd import _lib
from xdem_amd.synth import fbm_torch
from xdem_amd.terrain import terrain_attributes_device

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index",
        "terrain_ruggedness_index"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
dem = fbm_torch(n, n, dev, seed=42)
ctx = _lib.default_context(0)
out = torch.empty((len(FULL), n, n), device=dev, dtype=torch.float32)
cases = [
    ("FL full11", FULL, "Florinsky"), ("FL surf9", FULL[:9], "Florinsky"), ("FL SAH", FULL[:3], "Florinsky"),
    ("FL slope", FULL[:1], "Florinsky"), ("FL aspect", FULL[1:2], "Florinsky"), ("FL hillshade", FULL[2:3], "Florinsky"),
    ("FL curv6", FULL[3:9], "Florinsky"), ("FL win2", FULL[9:], "Florinsky"), ("FL tpi", FULL[9:10], "Florinsky"),
    ("ZT full11", FULL, "ZevenbergThorne"), ("ZT SAH", FULL[:3], "ZevenbergThorne"),
    ("Horn SAH+win", FULL[:3] + FULL[9:], "Horn"), ("Horn slope+aspect", FULL[:2], "Horn"),
    ("rugosity", ["rugosity"], "Florinsky"), ("roughness", ["roughness"], "Florinsky"),
    ("fractal w13", ["fractal_roughness"], "Florinsky"),
]
res = []
for name, attrs, fit in cases:
    o = out[: len(attrs)]
    ms = []
    for i in range(6):
        terrain_attributes_device(dem, attrs, resolution=10.0, surface_fit=fit, out=o, ctx=ctx)
        ms.append(ctx.last_kernel_ms())
    ms = sorted(ms[1:])
    t = ms[len(ms) // 2]
    bpp = 4 + 4 * len(attrs)
    r = {"case": name, "n": n, "ms": round(t, 3), "Mpix_s": round(n * n / t / 1e3, 1), "GBps": round(bpp * n * n / t / 1e6, 1)}
    res.append(r)
    print(json.dumps(r), flush=True)
