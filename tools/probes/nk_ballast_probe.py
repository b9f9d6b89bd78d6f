"""Does the Nuth-Kaab step care WHERE its rasters lie (profiles/r06ax_placement_vs_time.txt: the first tens of GB a fresh process gets can be slow memory)?
The C3 pair and plan created first thing in the process, against the same created while a ballast allocation holds the first `ballast_gb` GB.

  python tools/probes/nk_ballast_probe.py [ballast_gb=0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from xdem_amd import _lib, coreg

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
ballast = torch.empty(int(gb * 1e9), dtype=torch.uint8, device=dev) if gb > 0 else None
ref, tba = bench._c3_pair(dev, 20000)
plan = coreg.NKPlan(ref.contiguous(), tba.contiguous(), None, ctx)
del ballast
res = (10.0, 10.0)
plan.step(0.0, 0.0, res, 72)
out = []
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(3):
        plan.step(3.0 + i + rep, -4.0, res, 72)
    sampled = (time.perf_counter() - t0) / 3
    for i in range(3):
        plan.step(1.7 + 2e-3 * i, 0.6, res, 72)
    t0 = time.perf_counter()
    for i in range(5):
        plan.step(1.7 + 2e-3 * ((i * 7) % 5 - 2), 0.6 + 1.5e-3 * ((i * 3) % 5 - 2), res, 72)
    settled = (time.perf_counter() - t0) / 5
    out.append((round(sampled * 1e3, 3), round(settled * 1e3, 3)))
print(f"ballast {gb:5.0f} GB: (sampled, settled) ms per step x3:", out, plan.route_counts(), flush=True)
plan.close()
