"""Do planes in freshly mapped scattered ranges (HIP virtual memory management) always hold what the kernel wrote?  Repeats:
allocate the planes of a 46400^2 raster (scattered, 17 GB), run a window kernel into them, compare EVERY pixel with the same
launch into torch-allocated planes, free; between repeats other scattered ranges and torch blocks come and go (the pattern of
the GPU test file in which `test_raster_beyond_2g_pixels` saw 4 MiB of stale pixels on two boxes).  (measurement tool)

  python tools/vmm_stale_probe.py [repeats] [n]"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from xdem_amd import _lib, terrain
from xdem_amd.synth import fbm_torch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 46400
attrs, kw = ["roughness", "topographic_position_index"], {"window_size": 5}
ctx = _lib.default_context(0)
dem = fbm_torch(n, n, "cuda", seed=7)
ref = torch.empty((2, n, n), device="cuda")
terrain.terrain_attributes_device(dem, attrs, resolution=10.0, out=ref, **kw)
torch.cuda.synchronize()
rng = np.random.default_rng(3)
bad = 0
for it in range(reps):
    # churn: scattered ranges of other sizes (pooled / freed), torch blocks cached and returned
    junk = [terrain.alloc_planes(int(k), 8192, 8192, backing="scattered") for k in rng.integers(1, 12, 3)]
    tj = [torch.empty(int(m) << 20, device="cuda") for m in rng.integers(64, 2048, 4)]
    for j in junk:
        j.fill_(float(it))
    del junk, tj
    if it % 3 == 2:
        torch.cuda.empty_cache()
    if it % 4 == 3:
        ctx.release_pool()
    gc.collect()
    out = terrain.terrain_attributes_device(dem, attrs, resolution=10.0, **kw)
    torch.cuda.synchronize()
    neq = out.view(torch.int32) != ref.view(torch.int32)
    # NaN patterns compare equal bitwise here (same kernel, same inputs)
    cnt = int(neq.sum())
    msg = ""
    if cnt:
        bad += 1
        rows_any = neq.any(dim=2)                       # (2, n)
        per_plane = [int(neq[k].sum()) for k in range(2)]
        rr = rows_any.nonzero()
        rows = rr[:, 1]
        r_lo, r_hi = int(rows.min()), int(rows.max())
        cols = neq[:, r_lo:r_hi + 1].any(dim=1).any(dim=0).nonzero()
        off0 = (int(rr[0, 0]) * n * n + r_lo * n + int(cols.min())) * 4
        msg = (f" per plane {per_plane} rows {r_lo}..{r_hi} ({torch.unique(rows).numel()} rows) cols {int(cols.min())}..{int(cols.max())}"
               f" first byte offset {off0:#x} (mod 2 MiB {off0 % (2 << 20):#x}, mod 8 MiB {off0 % (8 << 20):#x})")
        # a second look at the same memory, and a second launch into the same range
        torch.cuda.synchronize()
        cnt2 = int((out.view(torch.int32) != ref.view(torch.int32)).sum())
        terrain.terrain_attributes_device(dem, attrs, resolution=10.0, out=out, **kw)
        torch.cuda.synchronize()
        cnt3 = int((out.view(torch.int32) != ref.view(torch.int32)).sum())
        msg += f"; re-read {cnt2}, after a second launch into the same range {cnt3}"
    print(f"repeat {it}: ptr {out.data_ptr():#x} differing pixels {cnt}{msg}", flush=True)
    del out, neq
print("repeats with stale pixels:", bad, "of", reps)
