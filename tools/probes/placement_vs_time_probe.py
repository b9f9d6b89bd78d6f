"""Is a slow plane set slow because of WHERE it lies or because of WHEN it is used?  One allocation of each kind, the 40000^2 eleven-plane
launch timed on it again and again for ~12 s right after the process started (and right after a 70 GB set was freed).

  python tools/probes/placement_vs_time_probe.py [size=40000]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from xdem_amd import _lib, terrain
from xdem_amd.synth import fbm_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature", "flowline_curvature", "max_curvature",
        "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
ctx = _lib.default_context()
t_proc = time.perf_counter()
dem = fbm_torch(n, n, "cuda", seed=42)
kw = dict(resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)


def series(planes, label, seconds=12.0):
    out = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            terrain.terrain_attributes_device(dem, FULL, out=planes, **kw)
        e1.record()
        torch.cuda.synchronize()
        out.append((round(time.perf_counter() - t_proc, 1), round(e0.elapsed_time(e1) / 3, 2)))
        time.sleep(0.4)
    print(label, out, flush=True)


for kind in ("scattered", "torch"):
    planes = terrain.alloc_planes(len(FULL), n, n, torch.float32, ctx, backing=kind)
    series(planes, f"[{kind}, first allocation of its kind, seconds since process start / ms per launch]")
    del planes
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    planes = terrain.alloc_planes(len(FULL), n, n, torch.float32, ctx, backing=kind)
    series(planes, f"[{kind}, allocated right after the first one was freed]", 8.0)
    del planes
    gc.collect()
    torch.cuda.empty_cache()
