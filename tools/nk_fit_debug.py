"""bench.py's Nuth-Kaab leg step by step with the library's debug lines (which steps are predicted, how far off the centres lay,
which missed): XDEMHIP_DEBUG=1 python tools/nk_fit_debug.py [size]   (measurement tool)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, scipy.optimize
import bench
from xdem_amd import _lib, coreg
m = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda", 0)
ref, tba = bench._c3_pair(dev, m)
ctx = _lib.default_context(0)
plan = coreg.NKPlan(ref.contiguous(), tba.contiguous(), None, ctx)
res = (10.0, 10.0)
plan.step(0.0, 0.0, res, 72)
for i in range(3):
    plan.step(3.0 + i, -4.0, res, 72)
print("routes after steps", plan.route_counts(), flush=True)
off = coreg._iterate(plan, res, 0.0, 10, 72, scipy.optimize.curve_fit, True)
print("routes after fit", plan.route_counts(), off, flush=True)
for i in range(5):
    t0 = time.perf_counter()
    plan.step(off[0] + 2e-3 * ((i * 7) % 5 - 2), off[1] + 1.5e-3 * ((i * 3) % 5 - 2), res, 72)
    print(f"settled step {i}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
print("routes after settled steps", plan.route_counts(), flush=True)
plan.close()
