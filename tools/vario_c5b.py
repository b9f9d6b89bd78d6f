"""The variogram workload of bench.py's `secondary` block alone (C5 reading B: 8.3e10 pairs, 50 lags) -- the command the
variogram counter profiles are collected on.  python tools/vario_c5b.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from xdem_amd import _lib
from xdem_amd import spatialstats as ss

rng = np.random.default_rng(45)
runs, samples, rings, L = int(os.environ.get("C5_RUNS", "100")), 9091, 10, 20000
lattice = os.environ.get("C5_LATTICE", "1") == "1"   # raster pixels (integer-lattice kernels) or arbitrary coordinates (float64 kernels)
blocks = []
for _ in range(runs):
    if lattice:
        ax, ay = rng.integers(0, L, samples).astype(np.float64), rng.integers(0, L, samples).astype(np.float64)
        bx, by = rng.integers(0, L, samples * rings).astype(np.float64), rng.integers(0, L, samples * rings).astype(np.float64)
    else:
        ax, ay = rng.uniform(0, L, samples), rng.uniform(0, L, samples)
        bx, by = rng.uniform(0, L, samples * rings), rng.uniform(0, L, samples * rings)
    av = (np.sin(ax / 900.0) + 0.2 * rng.normal(size=samples)).astype(np.float32)
    bv = (np.sin(bx / 900.0) + 0.2 * rng.normal(size=samples * rings)).astype(np.float32)
    blocks.append((ax, ay, av, bx, by, bv))
edges = np.geomspace(np.sqrt(2), np.hypot(L, L), 50)
ctx = _lib.default_context(0)
ps = ss.PairSet(blocks, edges, ctx)
for _ in range(3):
    ps.sums(0)
ms = ctx.last_kernel_ms()
t0 = time.perf_counter()
ss.class_medians(ps)
dt = time.perf_counter() - t0
print(f"pairs {ps.n_pairs:.3e}: Matheron pass {ms:.2f} ms = {ps.n_pairs / ms / 1e6:.0f} Gpairs/s; exact Dowd {dt * 1e3:.1f} ms = "
      f"{ps.n_pairs / dt / 1e9:.0f} Gpairs/s", flush=True)
ps.close()
