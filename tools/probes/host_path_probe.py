"""End-to-end rate of the host-buffer entry (numpy in, numpy out: H2D + kernel + D2H): python tools/probes/host_path_probe.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from xdem_amd import terrain
from xdem_amd.synth import fbm_numpy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
dem = np.tile(fbm_numpy((2048, 2048), seed=1), (n // 2048, n // 2048))
from xdem_amd import _lib

threads = [int(a) for a in sys.argv[2:]] or [8]
for nt, attrs in [(t, a) for t in threads for a in (FULL, ["slope"])]:
    _lib.default_context().set_option("host_copy_threads", nt)
    print(f"[{nt} copy threads]", end=" ")
    terrain.get_terrain_attribute(dem[:1024, :1024], attrs, resolution=10.0)
    t0 = time.perf_counter()
    out = terrain.get_terrain_attribute(dem, attrs, resolution=10.0)
    dt = time.perf_counter() - t0
    nbytes = dem.nbytes * (1 + len(attrs))
    print(f"{n}x{n} host path, {len(attrs)} attribute(s): {dt*1e3:.0f} ms = {n*n/dt/1e6:.0f} Mpixel/s, {nbytes/dt/1e9:.1f} GB/s over PCIe + alloc", flush=True)
    del out
