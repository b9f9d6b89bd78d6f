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
# texture shading (hipFFT + 3 streaming kernels), host-timed around the device-resident C call
import ctypes
import time

import numpy as np

tex = torch.empty((n, n), device=dev, dtype=torch.float32)
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.check(ctx._L.xdemhip_texture_shading(ctx.handle, dem.data_ptr(), 0, n, n, 0.8, 0, tex.data_ptr(), 1))
    torch.cuda.synchronize()
    tt = (time.perf_counter() - t0) * 1e3
print(json.dumps({"case": "texture_shading alpha=0.8 (incl. plan creation)", "n": n, "ms": round(tt, 3), "Mpix_s": round(n * n / tt / 1e3, 1)}), flush=True)
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
