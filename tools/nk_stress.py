"""Many one-pass Nuth-Kaab steps in a row, round 5's forms (option nk_binseg = 1: value-bucket selections, histogram passes whose last
workgroup advances the selection states behind a ticket) against round 4's generic selections (nk_binseg = 0) on the same pair and the
same sequence of shifts: every output identical, and every step answered by the one-pass route -- a hand-over that ever read a stale
histogram would give a bracket that misses, which the integer counts catch and the route counter shows (the step then falls to the
two-pass route: exact, but slower).     python tools/nk_stress.py [size] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler

faulthandler.dump_traceback_later(500, exit=True)
import numpy as np
import torch

import bench
from xdem_amd import _lib, coreg

m = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
ref, tba = bench._c3_pair(dev, m)
ref, tba = ref.contiguous(), tba.contiguous()
out = {}
for form in (1, 0):
    ctx = _lib.Context(0)
    ctx.set_option("nk_binseg", form)
    plan = coreg.NKPlan(ref, tba, None, ctx)
    plan.step(0.0, 0.0, (10.0, 10.0), 72)
    res = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        res.append(plan.step(3.0 + 0.0137 * i, -4.0 + 0.0091 * i, (10.0, 10.0), 72 if i % 3 else 36))
    dt = (time.perf_counter() - t0) / k
    routes = plan.route_counts()
    print(f"nk_binseg {form}: {k} steps at {m}x{m}, {dt * 1e3:.3f} ms per step, routes {routes}", flush=True)
    out[form] = (res, routes)
    plan.close()
    ctx.close()
bad = 0
for i, (a, b) in enumerate(zip(out[1][0], out[0][0])):
    same = (a["n_valid"] == b["n_valid"] and a["vshift"] == b["vshift"] and np.array_equal(a["counts"], b["counts"])
            and np.array_equal(a["medians"], b["medians"], equal_nan=True) and np.array_equal(a["edges"], b["edges"]))
    if not same:
        bad += 1
        print("step", i, "differs", flush=True)
fell = out[1][1]["twopass"] + out[1][1]["plain"]
print(f"steps that differ: {bad}; steps of the round-5 form that left the one-pass route: {fell}", flush=True)
sys.exit(1 if bad or fell else 0)
