"""Pair-kernel throughput probe (GPU box): Gpairs/s of the sums and histogram passes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from xdem_amd import _lib
from xdem_amd import spatialstats as ss

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
x, y = rng.uniform(0, 20000, n), rng.uniform(0, 20000, n)
v = (np.sin(x / 900) + 0.2 * rng.normal(size=n)).astype(np.float32)
edges = np.geomspace(np.sqrt(2), np.hypot(20000, 20000), 50)
ctx = _lib.default_context(0)
for mode in ("pdist", "cdist"):
    blocks = [(x, y, v)] if mode == "pdist" else [(x[: n // 8], y[: n // 8], v[: n // 8], x, y, v)]
    ps = ss.PairSet(blocks, edges, ctx)
    for name, fn in (("sums_sq", lambda: ps.sums(0)), ("sums_sqrt", lambda: ps.sums(1)),
                     ("hist_first", lambda: ps.hist(24, True, None))):
        fn()
        t0 = time.perf_counter()
        fn()
        wall = time.perf_counter() - t0
        ms = ctx.last_kernel_ms()
        print(f"{mode} {name}: pairs={ps.n_pairs:.3e} kernel={ms:.2f} ms -> {ps.n_pairs / ms / 1e6:.1f} Gpairs/s (wall {wall*1e3:.1f} ms)", flush=True)
    # per-pass kernel times of the radix selection
    prefix = np.zeros(ps.nb, dtype=np.uint64)
    for pno, shift in enumerate((24, 16, 8, 0)):
        h = ps.hist(shift, pno == 0, None if pno == 0 else prefix)
        ms = ctx.last_kernel_ms()
        if pno == 0:
            count = h.sum(1); rank = np.where(count > 0, (count - 1) // 2, 0).astype(np.uint64)
        cum = np.cumsum(h, 1); d = np.minimum((cum > rank[:, None]).argmax(1), 255)
        before = cum[np.arange(ps.nb), d] - h[np.arange(ps.nb), d]
        prefix |= d.astype(np.uint64) << np.uint64(shift); rank = rank - before
        print(f"   pass {pno} shift {shift}: {ms:.2f} ms -> {ps.n_pairs / ms / 1e6:.1f} Gpairs/s", flush=True)
    t0 = time.perf_counter()
    med, cnt = ss.class_medians(ps)
    print(f"{mode} dowd full select: {time.perf_counter() - t0:.3f} s -> {ps.n_pairs / (time.perf_counter() - t0) / 1e9:.2f} Gpairs/s effective", flush=True)
    ps.close()
