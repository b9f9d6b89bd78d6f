"""Small attribute sets of the 40000^2 raster under the band heights of the streaming kernel (option "terrain_stream" = 128 / 256 /
512) and the strip orders (option "terrain_order"): median kernel time of 7 launches each.  (measurement tool)
  python tools/small_sets_probe.py [size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from xdem_amd import _lib, terrain
from xdem_amd.synth import fbm_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
ctx = _lib.default_context(0)
dem = fbm_torch(n, n, "cuda", seed=42)
sets = {"slope": (["slope"], {}), "slope+aspect Horn": (["slope", "aspect"], {"surface_fit": "Horn"}),
        "slope+aspect Florinsky": (["slope", "aspect"], {}), "hillshade": (["hillshade"], {})}
for name, (attrs, kw) in sets.items():
    out = terrain.alloc_planes(len(attrs), n, n)
    row = []
    for opt, val in (("terrain_stream", 1), ("terrain_stream", 256), ("terrain_stream", 512), ("terrain_order", 1), ("terrain_order", 3)):
        ctx.set_option(opt, val)
        ts = []
        for _ in range(7):
            terrain.terrain_attributes_device(dem, attrs, resolution=10.0, out=out, **kw)
            torch.cuda.synchronize()
            ts.append(ctx.last_kernel_ms())
        ctx.set_option("terrain_stream", 1)
        ctx.set_option("terrain_order", 0)
        row.append(f"{opt}={val}: {np.median(ts):.3f}")
    print(f"{name:26s} " + "  ".join(row), flush=True)
    del out
