"""Device-resident rates of the two public functions of round 6's last session.

  python tools/probes/conv_probe.py [size=16384]

convolution: the five Florinsky stencil tables (5 x 5) in one call on a size^2 float32 DEM -- 4 B read + 5 x 8 B written per
pixel --, then three 3 x 3 filters (28 B per pixel); kernel times from the context's events (xdemhip_last_kernel_ms).
get_perbin_nd_binning: two float32 variables of size^2 pixels through the C-ABI on device arrays (8 B read + 8 B written)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from xdem_amd import _lib, spatialstats
from xdem_amd.synth import fbm_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = _lib.default_context()
dem = fbm_torch(n, n, "cuda", seed=1)[None].contiguous()
# the five Florinsky stencil tables over their dividers at res = 10 (the structure of fill_ref_weights, csrc/terrain_math.h)
u5, c5 = np.array([-2, -1, 0, 1, 2]), np.array([2, -1, -2, -1, 2])
al, be = np.array([44, 62, 68, 62, 44]), np.array([-31, 5, 17, 5, -31])
a5, b5 = np.array([0, -1, 0, 1, 0]), np.array([-1, 0, 0, 0, 1])
zx = np.outer(al, a5) + np.outer(be, b5)
fl = np.stack([zx / 4200.0, -zx.T / 4200.0, np.tile(c5, (5, 1)) / 3500.0, np.tile(c5[:, None], (1, 5)) / 3500.0, -np.outer(u5, u5) / 10000.0]).astype(np.float64)
rng = np.random.default_rng(0)
for label, filt in (("5 Florinsky tables 5x5", fl), ("3 filters 3x3", rng.normal(size=(3, 3, 3))), ("1 filter 9x9", rng.normal(size=(1, 9, 9)))):
    for method in ("scipy", "numba"):
        out = spatialstats.convolution(dem, filt, method=method)
        torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            out = spatialstats.convolution(dem, filt, method=method)
            torch.cuda.synchronize()
            ms.append(ctx.last_kernel_ms())
        b = (4 + 8 * filt.shape[0]) * n * n
        print(f"convolution {label:24s} {method:5s} {n}^2 f32: {min(ms):8.3f} ms  {b / min(ms) / 1e6:7.1f} GB/s  ({b / n / n} B/px)", flush=True)
        del out
# per-bin lookup on device arrays through the C-ABI
a = torch.rand((n, n), device="cuda") * 40
b_ = torch.rand((n, n), device="cuda") * 5
out = torch.empty((n, n), dtype=torch.float64, device="cuda")
na, nb = 10, 10
left = np.concatenate([np.arange(na) * 4.0, np.arange(nb) * 0.5])
right = np.concatenate([(np.arange(na) + 1) * 4.0, (np.arange(nb) + 1) * 0.5])
table = rng.normal(size=na * nb)
kind = np.ones(na * nb, dtype=np.uint8)
ptrs = (ctypes.c_void_p * 2)(a.data_ptr(), b_.data_ptr())
dts = (ctypes.c_int * 2)(_lib.F32, _lib.F32)
nint = (ctypes.c_int * 2)(na, nb)
miss = ctypes.c_int64()
dp = ctypes.POINTER(ctypes.c_double)
for disjoint in (1, 0):
    ms = []
    for _ in range(3):
        rc = ctx._L.xdemhip_perbin_lookup(ctx.handle, ptrs, dts, 2, n * n, nint, left.ctypes.data_as(dp), right.ctypes.data_as(dp), table.ctypes.data_as(dp),
                                          kind.ctypes.data_as(ctypes.c_char_p), disjoint, out.data_ptr(), ctypes.byref(miss), _lib.DEVICE)
        ctx.check(rc)
        ms.append(ctx.last_kernel_ms())
    print(f"perbin lookup 2 vars x 10 intervals, disjoint={disjoint}, {n}^2: {min(ms):8.3f} ms  {16 * n * n / min(ms) / 1e6:7.1f} GB/s (16 B/px), finite {float(torch.isfinite(out).double().mean()):.3f}", flush=True)
# the public call end to end on host arrays
ah, bh = a.cpu().numpy(), b_.cpu().numpy()
import pandas as pd

df = pd.DataFrame({"a": [pd.Interval(np.float64(left[i]), np.float64(right[i]), closed="left") for i in range(na) for _ in range(nb)],
                   "b": [pd.Interval(np.float64(left[na + j]), np.float64(right[na + j]), closed="left") for _ in range(na) for j in range(nb)],
                   "count": np.full(na * nb, 50.0), "nmad": table})
t0 = time.perf_counter()
res = spatialstats.get_perbin_nd_binning(df, [ah, bh], ["a", "b"], statistic="nmad", min_count=0)
print(f"get_perbin_nd_binning on host arrays {n}^2 x 2 variables: {time.perf_counter() - t0:.2f} s wall", flush=True)
assert np.array_equal(res, out.cpu().numpy(), equal_nan=True)
