#!/bin/bash
# NK parity tests + step timing + kernel timeline (run through gpurun)
TAG=${1:-r02i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nuthkaab_gpu.py ${2:-} -m gpu -x -q > $OUT/pytest_nk.log 2>&1; tail -4 $OUT/pytest_nk.log
timeout 300 python tools/nk_probe.py 20000 4 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/nk -o nk -- python $GRAFT_REPO_ROOT/tools/nk_probe.py 20000 3 > $OUT/nk.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py $OUT/nk/nk_kernel_trace.csv 64 > $OUT/timeline.txt; grep -v "fillBuffer\|select_advance\|copyBuffer" $OUT/timeline.txt | tail -40
