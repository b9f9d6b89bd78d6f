#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03q}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
XDEMHIP_DEBUG=1 timeout 300 python - > $OUT/vario.log 2>&1 <<PY
import sys, time
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import numpy as np
from xdem_amd import _lib, spatialstats as ss
from xdem_amd.synth import c5_variogram_blocks
ctx=_lib.default_context(0)
blocks, edges = c5_variogram_blocks("cuda", runs=100, samples=9091)
res={}
for srt in (1,0):
    ctx.set_option("vario_sort", srt)
    t=time.time(); ps = ss.PairSet(blocks, edges, ctx); print("sort",srt,"create",round(time.time()-t,2))
    s,c = ps.sums(0); s,c = ps.sums(0); print("sort",srt,"matheron ms", ctx.last_kernel_ms(), ps.n_pairs/ctx.last_kernel_ms()/1e6, "Gpairs/s")
    for rep in range(3):
        t=time.time(); med,cnt = ss.class_medians(ps); dt=time.time()-t; print("sort",srt,"dowd wall", round(dt,4), round(ps.n_pairs/dt/1e9,1), "Gpairs/s", flush=True)
    res[srt]=(s,c,med,cnt); ps.close()
ctx.set_option("vario_sort", 1)
assert np.array_equal(res[1][1],res[0][1]) and np.array_equal(res[1][3],res[0][3]) and np.array_equal(res[1][2],res[0][2],equal_nan=True)
print("max rel diff of sums", np.max(np.abs(res[1][0]-res[0][0])/res[0][0]))
PY
grep -v amdgpu $OUT/vario.log | grep -v "pair medians" | tail -12
grep "pair medians" $OUT/vario.log | head -12; timeout 900 python -m pytest tests/test_variogram_gpu.py -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log
