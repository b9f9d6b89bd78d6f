"""Every step of a FRESH plan's 10-iteration fit timed on its own (the first step of a plan fills the aspect-bin cache and takes the
full bracket rule; the bench line's whole-fit figure averages over it):   XDEMHIP_DEBUG=1 python tools/probes/nk_fit_steps_probe.py [size]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import scipy.optimize
import torch

import bench
from xdem_amd import _lib, coreg

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda", 0)
ref, tba = bench._c3_pair(dev, m)
ctx = _lib.default_context(0)
for trial in range(2):
    plan = coreg.NKPlan(ref.contiguous(), tba.contiguous(), None, ctx)
    inner = plan.step
    times = []

    def timed_step(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = inner(*a, **k)
        times.append((time.perf_counter() - t0) * 1e3)
        return r

    plan.step = timed_step
    t0 = time.perf_counter()
    off = coreg._iterate(plan, (10.0, 10.0), 0.0, 10, 72, scipy.optimize.curve_fit, True)
    total = (time.perf_counter() - t0) * 1e3
    print(f"trial {trial}: steps (ms) " + " ".join(f"{t:.2f}" for t in times) + f" | sum {sum(times):.2f}, fit {total:.2f} ms, routes {plan.route_counts()}", flush=True)
    plan.close()
