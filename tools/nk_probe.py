"""Nuth-Kaab step timing probe (GPU box): python tools/nk_probe.py [size]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from xdem_amd import _lib, coreg
from xdem_amd.synth import fbm_numpy

m = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ref = fbm_numpy((m, m), seed=42)
tba = (np.roll(ref, (1, -2), (0, 1)) + 2.0).astype(np.float32)
hole = fbm_numpy((m, m), seed=44, hurst=1.0, mean=0.0, std=1.0)
tba[hole < np.percentile(hole, 20)] = np.nan
ctx = _lib.default_context(0)
plan = coreg.NKPlan(ref, tba, None, ctx)
plan.step(0.0, 0.0, (10.0, 10.0), 72)
for sh in ((3.0, -4.0), (17.0, 6.0)):
    t0 = time.perf_counter()
    d = plan.step(sh[0], sh[1], (10.0, 10.0), 72)
    dt = time.perf_counter() - t0
    print(f"step {m}x{m}: {dt*1e3:.2f} ms -> {m*m/dt/1e6:.0f} Mpixel-iterations/s  (n_valid {d['n_valid']})", flush=True)
plan.close()
