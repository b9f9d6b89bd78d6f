#!/bin/bash
# round-5, session q: the final form of the NK step's selections (dh median among candidates in 3 launches; the dh sample's passes advance
# their own states) -- step time with / without, dispatch sequence, then the whole GPU suite
TAG=${1:-r05q}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python -u tools/nk_trace.py 20000 6 > $O/nk_default.log 2>&1; grep -E "step|routes" $O/nk_default.log | tail -8
XDEM_NK_BINSEG=0 timeout 300 python -u tools/nk_trace.py 20000 6 > $O/nk_old.log 2>&1; grep -E "step|routes" $O/nk_old.log | tail -8
timeout 300 python -u tools/nk_trace.py 20000 6 > $O/nk_default2.log 2>&1; grep -E "step|routes" $O/nk_default2.log | tail -8
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/nktrace -o nk -- python $R/tools/nk_trace.py 20000 3 > $R/$O/nktrace.log 2>&1 )
python tools/trace_sequence.py $O/nktrace 26 > $O/nk_sequence.txt 2>&1; tail -28 $O/nk_sequence.txt | cut -c1-120
find $O -name '*.csv' -size +2M -delete
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
tail -6 $O/pytest_all.log | cut -c1-300
