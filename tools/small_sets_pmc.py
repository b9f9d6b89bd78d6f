"""Five launches each of the small attribute sets (slope Florinsky / slope+aspect Horn / hillshade / full 11) at 40000^2 -- the
command the SQ counters of the small sets are collected on (rocprofv3 --pmc passes, tools/sessions/gpu_r05d.sh).  (measurement tool)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xdem_amd import _lib, terrain
from xdem_amd.synth import fbm_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
ctx = _lib.default_context(0)
dem = fbm_torch(n, n, "cuda", seed=42)
FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature", "flowline_curvature",
        "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
for attrs, kw in ((["slope"], {}), (["slope", "aspect"], {"surface_fit": "Horn"}), (["hillshade"], {}), (FULL, {})):
    out = terrain.alloc_planes(len(attrs), n, n)
    for _ in range(5):
        terrain.terrain_attributes_device(dem, attrs, resolution=10.0, out=out, **kw)
    torch.cuda.synchronize()
    print(attrs[0], len(attrs), ctx.last_kernel_ms(), flush=True)
    del out
