import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, scipy.optimize
import bench
from xdem_amd import _lib, coreg
dev = torch.device("cuda", 0)
ref, tba = bench._c3_pair(dev, 20000)
ctx = _lib.default_context(0)
plan = coreg.NKPlan(ref.contiguous(), tba.contiguous(), None, ctx)
plan.step(0.0, 0.0, (10.0, 10.0), 72)
for i in range(3):
    plan.step(3.0 + i, -4.0, (10.0, 10.0), 72)
print("routes after steps", plan.route_counts(), flush=True)
off = coreg._iterate(plan, (10.0, 10.0), 0.0, 10, 72, scipy.optimize.curve_fit, True)
print("routes after fit", plan.route_counts(), off, flush=True)
plan.close()
