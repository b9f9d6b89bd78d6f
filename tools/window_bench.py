"""Rates of the windowed indexes for window sizes other than 3 (TPI + TRI + roughness in one launch; measurement tool)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler

faulthandler.dump_traceback_later(280, exit=True)
import numpy as np
import torch

from xdem_amd import _lib, terrain
from xdem_amd.synth import fbm_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = _lib.default_context(0)
dem = fbm_torch(n, n, "cuda", seed=42)
attrs = ["topographic_position_index", "terrain_ruggedness_index", "roughness"]
out = torch.empty((3, n, n), device="cuda")
for w in (3, 5, 7, 9, 13, 21, 31):
    t = []
    for i in range(5):
        terrain.terrain_attributes_device(dem, attrs, out=out, window_size=w, ctx=ctx)
        t.append(ctx.last_kernel_ms())
    ms = float(np.median(t[1:]))
    print(f"window {w:2d}: TPI + TRI + roughness {ms:8.3f} ms = {n * n / ms / 1e6:8.2f} Gpixel/s ({16 * n * n / ms / 1e6:7.1f} GB/s of 4 + 12 B/pixel)", flush=True)
for w in (5, 13):
    t = []
    for i in range(4):
        terrain.terrain_attributes_device(dem, attrs[:1], out=out[:1], window_size=w, ctx=ctx)
        t.append(ctx.last_kernel_ms())
    ms = float(np.median(t[1:]))
    print(f"window {w:2d}: TPI alone {ms:8.3f} ms = {n * n / ms / 1e6:8.2f} Gpixel/s", flush=True)
