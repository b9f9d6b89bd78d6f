"""Phase timings of the exact Dowd route on bench.py's C5 reading-B input under the switchable scikit-gstat conventions
(XDEMHIP_DEBUG prints the phases of xdemhip_pairs_medians): default, vario_edge = 1, vario_diff = 1, both.
  XDEMHIP_DEBUG=1 python tools/vario_conventions_probe.py [runs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xdem_amd import _lib
from xdem_amd import spatialstats as ss
from xdem_amd.synth import c5_variogram_blocks

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
ctx = _lib.default_context(0)
blocks, edges = c5_variogram_blocks(dev, runs=runs, samples=9091)
total = sum(int(b[0].size) * int(b[3].size) for b in blocks)
ref = None
for edge, diff in ((0, 0), (1, 0), (0, 1), (1, 1)):
    ctx.set_option("vario_edge", edge)
    ctx.set_option("vario_diff", diff)
    ps = ss.PairSet(blocks, edges, ctx)
    ps.sums(0)
    for rep in range(2):
        sys.stderr.write(f"---- vario_edge {edge} vario_diff {diff} call {rep}\n")
        sys.stderr.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        med, cnt = ss.class_medians(ps)
        dt = time.perf_counter() - t0
    print(f"vario_edge {edge} vario_diff {diff}: exact Dowd {dt * 1e3:.1f} ms = {total / dt / 1e9:.0f} Gpairs/s; counts sum {int(cnt.sum())}", flush=True)
    if ref is None:
        ref = (med, cnt)
    else:
        import numpy as np

        print("   medians vs default: max rel diff", float(np.nanmax(np.abs(med - ref[0]) / np.abs(ref[0]))), "counts equal", bool((cnt == ref[1]).all()), flush=True)
    ps.close()
ctx.set_option("vario_edge", 0)
ctx.set_option("vario_diff", 0)
