"""Device-memory leak check (GPU box): repeated calls of every entry point must return the free-memory reading to where it
started.  python tools/leak_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from xdem_amd import coreg, terrain
from xdem_amd import spatialstats as ss
from xdem_amd.synth import fbm_numpy

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "max_curvature", "topographic_position_index",
        "terrain_ruggedness_index", "rugosity", "fractal_roughness", "texture_shading", "roughness"]


def free():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def once(dem, tba, rng):
    terrain.get_terrain_attribute(dem, FULL, resolution=10.0)
    nk = coreg.NuthKaab(subsample=1, max_iterations=3).fit(dem, tba, None, resolution=10.0)
    nk.apply(tba, 10.0)
    coreg.NuthKaab(subsample=1, max_iterations=2, bin_statistic=np.nanmean).fit(dem, tba, None, resolution=10.0)
    ss.sample_empirical_variogram(tba - dem, gsd=10.0, subsample=200, random_state=1, estimator="dowd")
    ss.sample_empirical_variogram(tba - dem, gsd=10.0, subsample=200, random_state=1, subsample_method="pdist_ring")
    v = rng.normal(size=300000).astype(np.float32)
    ss.nd_binning(v, [rng.uniform(size=v.size).astype(np.float32)], ["a"], list_var_bins=10)
    ss.nmad_device(v)


dem = fbm_numpy((700, 900), seed=1)
tba = (np.roll(dem, (1, -1), (0, 1)) + 1.0).astype(np.float32)
rng = np.random.default_rng(0)
once(dem, tba, rng)
once(dem, tba, rng)
f0 = free()
for i in range(25):
    once(dem, tba, rng)
f1 = free()
print(f"free before {f0 / 2**20:.1f} MiB, after 25 rounds {f1 / 2**20:.1f} MiB, delta {(f0 - f1) / 2**20:.2f} MiB")
print("OK" if f0 - f1 < 8 * 2**20 else "LEAK")
