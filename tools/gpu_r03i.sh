#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03i}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bt -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > $OUT/bench_trace.log 2>&1
cd $GRAFT_REPO_ROOT
tail -c 1500 $OUT/bench_trace.log
python - <<PY
import csv,glob
f=glob.glob("$OUT/bt/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]: print(r['Name'][:110], r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
find $OUT/bt -name "*.csv" -size +1M -delete
timeout 300 python tools/terrain_opts_bench.py --size 40000 --reps 10 --rounds 2 --opts "" --combos "terrain_math=2+terrain_stream=1;terrain_math=0+terrain_stream=0" > $OUT/opts.log 2>&1; grep -v amdgpu $OUT/opts.log
XDEMHIP_DEBUG=1 timeout 300 python - > $OUT/dowd.log 2>&1 <<PY
import sys, time
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import numpy as np
from xdem_amd import _lib, spatialstats as ss
from xdem_amd.synth import c5_variogram_blocks
ctx=_lib.default_context(0)
blocks, edges = c5_variogram_blocks("cuda", runs=100, samples=9091)
ps = ss.PairSet(blocks, edges, ctx)
s,c = ps.sums(0)
print("counts per class", c.tolist())
t=time.time(); med,cnt = ss.class_medians(ps); print("dowd wall", time.time()-t)
PY
grep -v amdgpu $OUT/dowd.log | tail -20
timeout 600 python -m pytest tests/test_nuthkaab_gpu.py tests/test_patches_gpu.py tests/test_dist_gpu.py -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
