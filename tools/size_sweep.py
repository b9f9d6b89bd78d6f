"""Fused 11-attribute terrain launch at several raster shapes (one GPU): separates size, row-stride (power-of-two widths) and
shape effects.  python tools/size_sweep.py "40000x40000,65536x65536,65536x65600,32768x131072,131072x32768" """
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xdem_amd.terrain import terrain_attributes_device

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[1] if len(sys.argv) > 1 else "40000x40000,65536x65536").split(",")]
dev = torch.device("cuda", 0)
for (h, w) in shapes:
    # cheap smooth-ish surface without big temporaries: separable ramps + a hashed ripple
    r = torch.arange(h, device=dev, dtype=torch.float32)[:, None]
    c = torch.arange(w, device=dev, dtype=torch.float32)[None, :]
    dem = torch.empty((h, w), device=dev, dtype=torch.float32)
    step = max(1, (1 << 28) // w)
    for i in range(0, h, step):
        rr = r[i:i + step]
        dem[i:i + step] = 1000.0 + 30.0 * torch.sin(rr * 0.013) * torch.cos(c * 0.011) + 0.002 * rr + 5.0 * torch.sin(c * 0.21 + rr * 0.17)
    out = torch.empty((len(FULL), h, w), device=dev, dtype=torch.float32)
    out.fill_(0)
    torch.cuda.synchronize()
    ts = []
    for k in range(5):
        t0 = time.perf_counter()
        terrain_attributes_device(dem, FULL, resolution=10.0, out=out)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    best, med = min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2]
    print(f"{h}x{w}: first {ts[0]*1e3:.1f} ms, median {med*1e3:.2f} ms, best {best*1e3:.2f} ms = {h*w/med/1e9:.1f} Gpx/s, frac {48.0*h*w/med/8e12:.3f}", flush=True)
    del dem, out, r, c
    torch.cuda.empty_cache()
