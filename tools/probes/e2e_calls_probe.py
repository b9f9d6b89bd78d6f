"""End-to-end wall time of the public calls on HOST arrays (what a user of the reference's signatures sees), with cProfile's top host
entries per call -- a search for host-side costs that dwarf the kernels.   python tools/probes/e2e_calls_probe.py [size=12000]"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from xdem_amd import spatialstats, terrain
from xdem_amd.synth import fbm_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
dev = torch.device("cuda", 0)
dem = (fbm_torch(n, n, dev, seed=42) * 1.0).cpu().numpy()
dh = fbm_torch(n, n, dev, seed=3, hurst=0.3).cpu().numpy()
torch.cuda.synchronize()
small = dem[:1024, :1024].copy()


def timed(label, fn, warm=None):
    if warm is not None:
        warm()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    out = fn()
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"[{label}] {n}x{n}: {dt:.3f} s wall", flush=True)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(6)
    lines = [l for l in s.getvalue().splitlines() if l.strip() and ("/" in l or "{" in l)]
    print("\n".join("    " + l[:170] for l in lines[:6]), flush=True)
    return out


slope = timed("slope (Florinsky)", lambda: terrain.slope(dem, resolution=10.0), lambda: terrain.slope(small, resolution=10.0))
timed("slope + aspect (Horn)", lambda: terrain.get_terrain_attribute(dem, ["slope", "aspect"], resolution=10.0, surface_fit="Horn"),
      lambda: terrain.get_terrain_attribute(small, ["slope", "aspect"], resolution=10.0, surface_fit="Horn"))
curv = timed("max curvature", lambda: terrain.get_terrain_attribute(dem, "max_curvature", resolution=10.0),
             lambda: terrain.get_terrain_attribute(small, "max_curvature", resolution=10.0))
timed("roughness + TPI + TRI (w=3)", lambda: terrain.get_terrain_attribute(dem, ["roughness", "topographic_position_index", "terrain_ruggedness_index"], resolution=10.0),
      lambda: terrain.get_terrain_attribute(small, ["roughness", "topographic_position_index", "terrain_ruggedness_index"], resolution=10.0))
timed("rugosity", lambda: terrain.rugosity(dem, resolution=10.0), lambda: terrain.rugosity(small, resolution=10.0))
timed("fractal roughness (w=13)", lambda: terrain.fractal_roughness(dem), lambda: terrain.fractal_roughness(small))
timed("texture shading (first call at this size: rocFFT builds its plan)", lambda: terrain.texture_shading(dem), lambda: terrain.texture_shading(small))
timed("texture shading (second call)", lambda: terrain.texture_shading(dem))
names = ["slope", "maxc"]
timed("nd_binning (2 variables, 10 bins)", lambda: spatialstats.nd_binning(dh, [slope, curv], names, list_var_bins=10),
      lambda: spatialstats.nd_binning(dh[:1024, :1024], [slope[:1024, :1024], curv[:1024, :1024]], names, list_var_bins=10))
timed("infer_heteroscedasticity_from_stable", lambda: spatialstats.infer_heteroscedasticity_from_stable(dh, [slope, curv], list_var_names=names),
      lambda: spatialstats.infer_heteroscedasticity_from_stable(dh[:1024, :1024], [slope[:1024, :1024], curv[:1024, :1024]], list_var_names=names))
timed("nmad", lambda: spatialstats.nmad(dh))
