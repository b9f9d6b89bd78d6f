"""Nuth-Kaab step timing probe (GPU box), device-resident C3-like pair: python tools/nk_probe.py [size] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xdem_amd import _lib, coreg

if os.environ.get("NK_LIB"):  # A/B runs inside one session: another build of the library
    _lib.LIB_PATH = os.environ["NK_LIB"]
from xdem_amd.synth import fbm_torch

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
ref = fbm_torch(m, m, dev, seed=42)
tba = torch.roll(ref, shifts=(1, -2), dims=(0, 1)) + 2.0
hole = fbm_torch(m, m, dev, seed=44)
tba[hole < torch.quantile(hole[::16, ::16].flatten(), 0.2)] = float("nan")
del hole
torch.cuda.synchronize()
ctx = _lib.default_context(0)
if os.environ.get("NK_SELECTION"):
    ctx.set_option("selection", int(os.environ["NK_SELECTION"]))
plan = coreg.NKPlan(ref, tba, None, ctx)
plan.step(0.0, 0.0, (10.0, 10.0), 72)
for i in range(k):
    t0 = time.perf_counter()
    d = plan.step(3.0 + i, -4.0, (10.0, 10.0), 72)
    dt = time.perf_counter() - t0
    print(f"step {m}x{m}: {dt*1e3:.2f} ms -> {m*m/dt/1e6:.0f} Mpixel-iterations/s  (n_valid {d['n_valid']})", flush=True)
plan.close()
