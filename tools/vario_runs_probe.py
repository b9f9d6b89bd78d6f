"""Exact Dowd medians on bench.py's C5 input with the counting pass (i) run-length on the Morton-ordered copy (round 4), (ii) per
pair on the Morton-ordered copy, (iii) per pair on the caller's order (round 3): time per call, phase times (XDEMHIP_DEBUG lines
on stderr), identical medians.  (measurement tool)

  XDEMHIP_DEBUG=1 python tools/vario_runs_probe.py [samples [runs]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from xdem_amd import _lib
from xdem_amd import spatialstats as ss
from xdem_amd.synth import c5_variogram_blocks

if os.environ.get("XD_LIB"):   # A/B of library builds across processes
    _lib.LIB_PATH = os.environ["XD_LIB"]
samples = int(sys.argv[1]) if len(sys.argv) > 1 else 9091
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = _lib.default_context(0)
blocks, edges = c5_variogram_blocks(torch.device("cuda:0"), runs=runs, samples=samples)
total = sum(int(b[0].size) * int(b[3].size) for b in blocks)
ps = ss.PairSet(blocks, edges, ctx)
del blocks
ps.sums(0)
s_m, c_m = ps.sums(0)
print(f"pairs {total:.4e}; Matheron pass {ctx.last_kernel_ms():.2f} ms ({total / ctx.last_kernel_ms() / 1e6:.0f} Gpairs/s)", flush=True)
ref = None
cfgs = (("run-length, sorted copy", 1, True), ("per pair, sorted copy", 0, True), ("per pair, caller's order", 1, False),
        ("run-length, sorted copy", 1, True))
if os.environ.get("PROBE_CFG"):   # counter profiles: one form only
    cfgs = tuple(cfgs[int(k)] for k in os.environ["PROBE_CFG"].split(","))
for name, runs_opt, linked in cfgs:
    ctx.set_option("vario_runs", runs_opt)
    ctx.check(ctx._L.xdemhip_pairs_link_sorted(ps.handle_sel, ps.handle if linked else None))
    ss.class_medians(ps)
    sys.stderr.write(f"---- {name}: timed calls\n")
    sys.stderr.flush()
    dts = []
    for _ in range(3):
        t0 = time.perf_counter()
        med, cnt = ss.class_medians(ps)
        dts.append(time.perf_counter() - t0)
    dt = min(dts)
    same = "reference" if ref is None else ("identical" if (np.array_equal(med, ref[0], equal_nan=True) and np.array_equal(cnt, ref[1])) else "DIFFERENT")
    if ref is None:
        ref = (med, cnt)
    assert np.array_equal(cnt, c_m), "class counts differ from the Matheron pass"
    print(f"{name:28s}: {dt * 1e3:8.2f} ms  {total / dt / 1e9:8.1f} Gpairs/s  kernel {ctx.last_kernel_ms():.2f} ms  medians {same}", flush=True)
ps.close()
