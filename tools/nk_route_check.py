"""Nuth-Kaab step at full size: the fused / lean route against the plain route (generic kernels, plain digit passes) on the
same pair, and run-to-run determinism -- every output must be identical.  python tools/nk_route_check.py [size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from xdem_amd import _lib, coreg
from xdem_amd.synth import fbm_torch

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda", 0)
ref = fbm_torch(m, m, dev, seed=42)
tba = torch.roll(ref, shifts=(1, -2), dims=(0, 1)) + 2.0 + 0.5 * torch.randn((m, m), device=dev)
hole = fbm_torch(m, m, dev, seed=44)
tba[hole < torch.quantile(hole[::16, ::16].flatten(), 0.2)] = float("nan")
del hole
torch.cuda.synchronize()
ctx = _lib.default_context(0)
res = {}
for mode in (0, 1, 0):
    ctx.set_option("selection", mode)
    plan = coreg.NKPlan(ref, tba, None, ctx)
    outs = []
    for (sx, sy) in ((0.0, 0.0), (3.0, -4.0), (13.7, 21.3), (3.0, -4.0)):
        d = plan.step(sx, sy, (10.0, 10.0), 72)
        outs.append((d["n_valid"], d["vshift"], d["counts"].copy(), d["medians"].copy(), d["y_mean"], d["y_std"]))
    plan.close()
    res.setdefault(mode, []).append(outs)
ok = True
def same(a, b):
    return a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3], equal_nan=True)
for k in range(4):
    f1, f2, p = res[0][0][k], res[0][1][k], res[1][0][k]
    print(k, "n_valid", f1[0], p[0], "vshift", f1[1], p[1], "fused==fused", same(f1, f2), "fused==plain", same(f1, p),
          "mean rel diff", abs(f1[4] - p[4]) / abs(p[4]), flush=True)
    ok &= same(f1, f2) and same(f1, p)
print("ROUTES AGREE" if ok else "ROUTES DIFFER")
