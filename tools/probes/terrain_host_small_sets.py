"""Few-plane terrain calls on HOST arrays: wall time of get_terrain_attribute for one / two / three planes (the direct route of round 6:
one copy up, one launch, one copy per plane down; XDEMHIP_HOST_DIRECT=0 = the chunked staging pipeline).   python tools/probes/terrain_host_small_sets.py [size]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from xdem_amd import terrain
from xdem_amd.synth import fbm_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
dem = fbm_torch(n, n, torch.device("cuda", 0), seed=42).cpu().numpy()
torch.cuda.synchronize()
terrain.slope(dem[:1024, :1024].copy(), resolution=10.0)
for label, attrs, fit in (("slope (Florinsky)", ["slope"], "Florinsky"), ("slope + aspect (Horn)", ["slope", "aspect"], "Horn"),
                          ("slope + aspect + hillshade (Florinsky)", ["slope", "aspect", "hillshade"], "Florinsky"),
                          ("four planes (staged pipeline either way)", ["slope", "aspect", "hillshade", "curvature"], "Florinsky")):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = terrain.get_terrain_attribute(dem, attrs, resolution=10.0, surface_fit=fit)
        ts.append(time.perf_counter() - t0)
        del out
    gb = 4.0 * (1 + len(attrs)) * n * n / 1e9
    print(f"[{label}] {n}x{n}, XDEMHIP_HOST_DIRECT={os.environ.get('XDEMHIP_HOST_DIRECT', '1')}: {min(ts) * 1e3:.1f} ms (best of 3; {gb / min(ts):.1f} GB/s over PCIe, both directions added)", flush=True)
