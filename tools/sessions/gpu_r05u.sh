#!/bin/bash
# round-5, session u: per-workgroup aggregation of the gathered keys -- NK tests, step time, dispatch sequence
TAG=${1:-r05u}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-300
timeout 300 python -u tools/nk_trace.py 20000 6 > $O/nk_default.log 2>&1; grep -E "step|routes" $O/nk_default.log | tail -7
XDEM_NK_BINSEG=0 timeout 300 python -u tools/nk_trace.py 20000 6 > $O/nk_old.log 2>&1; grep -E "step|routes" $O/nk_old.log | tail -7
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/nktrace -o nk -- python $R/tools/nk_trace.py 20000 3 > $R/$O/nktrace.log 2>&1 )
python tools/trace_sequence.py $O/nktrace 24 > $O/nk_sequence.txt 2>&1; tail -26 $O/nk_sequence.txt | cut -c1-120
find $O -name '*.csv' -size +2M -delete
