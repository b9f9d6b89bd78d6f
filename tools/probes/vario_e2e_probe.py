"""End-to-end time of the variogram API call -- host-side preparation included -- next to the GPU pair passes bench.py times.

  python tools/probes/vario_e2e_probe.py [size=20000] [subsample=1000000] [n_variograms=10]

xdem_amd.spatialstats.sample_empirical_variogram(values (NumPy, host), gsd, subsample, n_variograms, estimator="dowd",
random_state=42): sampling of the equidistant metric space on the host (NumPy), upload of the sampled points, pair passes, exact
medians, DataFrame.  Prints the wall time, cProfile's top entries by cumulative time and the pairs formed."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from xdem_amd import spatialstats
from xdem_amd.synth import fbm_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
sub = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
nv = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dh = fbm_torch(n, n, "cuda", seed=3, hurst=0.3).cpu().numpy()
torch.cuda.synchronize()
for est in ("matheron", "dowd"):
    kw = dict(gsd=10.0, subsample=sub, n_variograms=nv, estimator=est, random_state=42, n_lags=50)
    spatialstats.sample_empirical_variogram(dh[:2000, :2000], **{**kw, "subsample": 1000, "n_variograms": 1})   # (library, allocator, first launches)
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    df = spatialstats.sample_empirical_variogram(dh, **kw)
    pr.disable()
    dt = time.perf_counter() - t0
    pairs = float(df["count"].sum())
    print(f"[{est}] {n}x{n}, subsample {sub:g}, {nv} runs: {dt:.2f} s wall, {pairs:.3e} pairs in the kept classes -> {pairs / dt / 1e9:.1f} Gpairs/s end to end", flush=True)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:3500], flush=True)
