"""End-to-end time of the Nuth-Kaab API call on HOST arrays (plan creation, uploads, iterations, host curve fits) next to the per-step
grid work bench.py times.   python tools/probes/nk_e2e_probe.py [size=20000]"""
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

import bench
from xdem_amd import coreg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ref_d, tba_d = bench._c3_pair(torch.device("cuda", 0), n)
ref, tba = ref_d.cpu().numpy(), tba_d.cpu().numpy()
del ref_d, tba_d
torch.cuda.empty_cache()
coreg.NuthKaab(subsample=1, max_iterations=2).fit(ref[:2000, :2000].copy(), tba[:2000, :2000].copy(), None, resolution=10.0)   # (library, first launches)
for label, kw in (("subsample=1 (all valid pixels)", dict(subsample=1)), ("default subsample=5e5", dict())):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    nk = coreg.NuthKaab(max_iterations=10, offset_threshold=0.0, **kw).fit(ref, tba, None, resolution=10.0)
    pr.disable()
    dt = time.perf_counter() - t0
    a = nk.meta["outputs"]["affine"]
    print(f"[{label}] {n}x{n} host arrays, 10 iterations: {dt:.2f} s wall; shift_x {a['shift_x']:.3f} shift_y {a['shift_y']:.3f} shift_z {a['shift_z']:.3f}", flush=True)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(12)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:2600], flush=True)
t0 = time.perf_counter()
out = coreg.apply_translation(tba, 17.0, 6.0, 2.0, 10.0)
print(f"[apply_translation] {n}x{n} host array: {time.perf_counter() - t0:.2f} s wall", flush=True)
