"""Two Nuth-Kaab steps on bench.py's C3 pair (the command rocprofv3 --kernel-trace wraps; tools/trace_sequence.py prints the
dispatch sequence of the last step): python tools/nk_trace.py [size] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler

faulthandler.dump_traceback_later(200, exit=True)
import torch

import bench
from xdem_amd import _lib, coreg

if os.environ.get("NK_LIB"):  # A/B of library builds across processes (measurement variants: xdem_amd/csrc/Makefile)
    _lib.LIB_PATH = os.environ["NK_LIB"]

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
ref, tba = bench._c3_pair(dev, m)
ctx = _lib.default_context(0)
if os.environ.get("NK_NARROW"):   # sample brackets of the one-pass step: -1 adaptive (default), 0 / 1 / 2 fixed
    ctx.set_option("nk_narrow", int(os.environ["NK_NARROW"]))
if os.environ.get("NK_FUSED"):    # 1 the one-pass step (default), 0 the plain route (stored dh, selections over the stored arrays)
    ctx.set_option("nk_fused", int(os.environ["NK_FUSED"]))
group = None
if os.environ.get("NK_HOOKED"):   # the partitioned plan's route on one GPU: a 1-rank RCCL group, reductions through the device-side hook
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29633")
    dist.init_process_group("nccl", rank=0, world_size=1)
    group = "world"
    if os.environ.get("NK_FUSED_DIST"):
        ctx.set_option("nk_fused_dist", int(os.environ["NK_FUSED_DIST"]))
plan = coreg.NKPlan(ref.contiguous(), tba.contiguous(), None, ctx, group)
if os.environ.get("NK_STAT") == "mean":   # bin_statistic = np.nanmean (plain route: one pass of per-bin sums)
    plan.set_statistic("mean")
if os.environ.get("NK_PREDICT"):   # round 6: 1 settled steps take predicted brackets (default), 0 every step samples
    ctx.set_option("nk_predict", int(os.environ["NK_PREDICT"]))
settled = os.environ.get("NK_SETTLED") == "1"   # the timed steps are those of a fit that has converged: shift changes of ~1e-4 px
plan.step(0.0, 0.0, (10.0, 10.0), 72)
plan.step(1.0, 0.0, (10.0, 10.0), 72)
if settled:
    for sx, sy in ((-15.0, -5.0), (-16.8, -5.9), (-16.98, -5.99), (-17.0, -6.0), (-17.003, -6.002)):
        plan.step(sx, sy, (10.0, 10.0), 72)
for i in range(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d = plan.step(-17.003 + 0.0007 * ((i * 7) % 5 - 2), -6.002 + 0.0005 * ((i * 3) % 5 - 2), (10.0, 10.0), 72) if settled else plan.step(3.0 + i, -4.0, (10.0, 10.0), 72)
    dt = time.perf_counter() - t0
    print(f"[{os.path.basename(_lib.LIB_PATH)}] step {m}x{m}: {dt * 1e3:.3f} ms (n_valid {d['n_valid']}, vshift {d['vshift']:.6f})", flush=True)
print("routes", plan.route_counts(), "reductions (host, device)", ctx.reduction_calls(), flush=True)
plan.close()
if group is not None:
    import torch.distributed as dist

    dist.destroy_process_group()
