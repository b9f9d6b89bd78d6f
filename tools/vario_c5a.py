"""BASELINE C5, SURVEY 8d reading A: subsample = 1e7 in the reference's sense -> runs = 100 centre disks of 223607 points,
each paired with 10 equidistant rings of 223607 points (5.0e13 pairs), 50 lag classes.  Times one Matheron pass and the
exact Dowd medians.  python tools/vario_c5a.py [runs] [samples]   (GPU box; ~5 GB of host points at full size)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from xdem_amd import _lib
from xdem_amd import spatialstats as ss

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 223607
rings, size = 10, 20000.0
r0 = size * np.sqrt(2) / np.sqrt(2) ** rings  # centre-disk radius = maxdist / 32
rng = np.random.default_rng(45)


def ring_points(cx, cy, rin, rout, n):
    r = np.sqrt(rng.uniform(rin * rin, rout * rout, n))
    t = rng.uniform(0, 2 * np.pi, n)
    x, y = cx + r * np.cos(t), cy + r * np.sin(t)
    v = (np.sin(x / 900.0) * np.cos(y / 700.0) + 0.3 * rng.standard_normal(n)).astype(np.float32)
    return x, y, v


t0 = time.perf_counter()
blocks = []
for _ in range(runs):
    cx, cy = rng.uniform(0, size, 2)
    ax, ay, av = ring_points(cx, cy, 0.0, r0, samples)
    parts = [ring_points(cx, cy, r0 * np.sqrt(2) ** i, r0 * np.sqrt(2) ** (i + 1), samples) for i in range(rings)]
    blocks.append((ax, ay, av, np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
                   np.concatenate([p[2] for p in parts])))
print(f"points generated in {time.perf_counter() - t0:.1f} s", flush=True)
edges = np.geomspace(np.sqrt(2), np.hypot(size, size), 50)
ctx = _lib.default_context(0)
t0 = time.perf_counter()
ps = ss.PairSet(blocks, edges, ctx)
print(f"pair set: {ps.n_pairs:.3e} pairs, upload {time.perf_counter() - t0:.1f} s", flush=True)
del blocks
t0 = time.perf_counter()
s, c = ps.sums(0)
dt = time.perf_counter() - t0
print(f"Matheron pass: {dt:.2f} s wall, kernel {ctx.last_kernel_ms() / 1e3:.2f} s -> {ps.n_pairs / dt / 1e9:.1f} Gpairs/s, "
      f"{int(c.sum()):.3e} pairs inside the lags", flush=True)
t0 = time.perf_counter()
med, cnt = ss.class_medians(ps)
dt = time.perf_counter() - t0
print(f"exact Dowd medians: {dt:.2f} s wall -> {ps.n_pairs / dt / 1e9:.1f} Gpairs/s; counts equal: {bool(np.array_equal(cnt, c))}; "
      f"median |dv| per class from {np.nanmin(med):.4f} to {np.nanmax(med):.4f}", flush=True)
ps.close()
