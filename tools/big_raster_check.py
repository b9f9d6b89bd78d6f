"""65536^2 float32 DEM (4.3e9 pixels > 2^32) through the fused kernel on one GPU: 206 GB of planes; checks a few crops against
separately computed crops (translation equivariance, bit-identical).  python tools/big_raster_check.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xdem_amd.synth import fbm_torch
from xdem_amd.terrain import terrain_attributes_device

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
dem = fbm_torch(n, n, dev, seed=42)
torch.cuda.synchronize()
print("dem built", float(dem[-1, -1]), torch.cuda.mem_get_info(), flush=True)
out = torch.empty((len(FULL), n, n), device=dev, dtype=torch.float32)
out.fill_(0)
torch.cuda.synchronize()
print("out allocated+touched", flush=True)
t0 = time.perf_counter()
terrain_attributes_device(dem, FULL, resolution=10.0, out=out)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n}x{n}: {dt*1e3:.1f} ms = {n*n/dt/1e6:.0f} Mpixel/s, {48*n*n/dt/1e9:.0f} GB/s algorithmic", flush=True)
ok = True
for (r, c) in ((0, 0), (n - 600, n - 900), (n // 2 + 13, 7), (40000, 61000)):
    r1, c1 = min(r + 600, n), min(c + 900, n)
    crop = dem[r:r1, c:c1].contiguous()
    oc = terrain_attributes_device(crop, FULL, resolution=10.0)
    torch.cuda.synchronize()
    a = oc[:, 2:-2, 2:-2].view(torch.int32)
    b = out[:, r + 2:r1 - 2, c + 2:c1 - 2].view(torch.int32)
    same = bool(torch.equal(a, b))
    ok &= same
    print("crop", (r, c), "interior bit-identical:", same, flush=True)
print("OK" if ok else "MISMATCH")
