"""Exact Dowd medians on bench.py's C5 input (fBm values, equidistant disk / ring sampler) for several assumed
design effects of the pair sample (option "vario_deff"): time per call, phase times and how centred the brackets were
(XDEMHIP_DEBUG lines on stderr), and that every setting returns the same medians.  (measurement tool)

  XDEMHIP_DEBUG=1 python tools/vario_c5_probe.py [deff ...]      (0 = the library's built-in rule)"""
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
deffs = [int(a) for a in sys.argv[1:]] or [4096, 0]   # (non-zero values need an -DXD_EXPERIMENT build: "vario_deff" is a measurement switch)
runs = int(os.environ.get("C5_RUNS", "100"))
ctx = _lib.default_context(0)
blocks, edges = c5_variogram_blocks(torch.device("cuda:0"), runs=runs, samples=9091)
total = sum(int(b[0].size) * int(b[3].size) for b in blocks)
ps = ss.PairSet(blocks, edges, ctx)
del blocks
ps.sums(0)
s_m, c_m = ps.sums(0)
print(f"pairs {total:.4e}; Matheron pass {ctx.last_kernel_ms():.2f} ms", flush=True)
ref = None
for d in deffs:
    if d != 0:
        ctx.set_option("vario_deff", d)
    ss.class_medians(ps)                      # first call under this setting (candidate buffers may grow)
    sys.stderr.write(f"---- vario_deff = {d}: timed call\n")
    sys.stderr.flush()
    t0 = time.perf_counter()
    med, cnt = ss.class_medians(ps)
    dt = time.perf_counter() - t0
    same = "reference" if ref is None else ("identical" if (np.array_equal(med, ref[0], equal_nan=True) and np.array_equal(cnt, ref[1])) else "DIFFERENT")
    if ref is None:
        ref = (med, cnt)
    assert np.array_equal(cnt, c_m), "class counts differ from the Matheron pass"
    print(f"vario_deff {d:5d}: exact Dowd {dt * 1e3:7.1f} ms = {total / dt / 1e9:7.1f} Gpairs/s   medians {same}", flush=True)
if any(deffs):
    ctx.set_option("vario_deff", 0)
ps.close()
