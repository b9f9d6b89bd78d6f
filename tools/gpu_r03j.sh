#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03j}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
XDEMHIP_DEBUG=1 timeout 300 python - > $OUT/dowd.log 2>&1 <<PY
import sys, time
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import numpy as np
from xdem_amd import _lib, spatialstats as ss
from xdem_amd.synth import c5_variogram_blocks
ctx=_lib.default_context(0)
blocks, edges = c5_variogram_blocks("cuda", runs=100, samples=9091)
ps = ss.PairSet(blocks, edges, ctx)
s,c = ps.sums(0); s,c = ps.sums(0); print("matheron ms", ctx.last_kernel_ms(), ps.n_pairs/ctx.last_kernel_ms()/1e6, "Gpairs/s")
for i in range(2):
    t=time.time(); med,cnt = ss.class_medians(ps); dt=time.time()-t; print("dowd wall", dt, ps.n_pairs/dt/1e9, "Gpairs/s", "counting pass ms", ctx.last_kernel_ms())
assert np.array_equal(cnt, c)
ctx.set_option("selection", 1); t=time.time(); med1,cnt1 = ss.class_medians(ps); print("plain passes", time.time()-t); ctx.set_option("selection", 0)
assert np.array_equal(med, med1, equal_nan=True) and np.array_equal(cnt, cnt1)
print("medians equal between routes")
PY
grep -v amdgpu $OUT/dowd.log | tail -12
true
