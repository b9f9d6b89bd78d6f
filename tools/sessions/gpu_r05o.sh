#!/bin/bash
# round-5, session o: sample selections whose passes advance their own states (hist_pass_kernel<T, true>) -- NK tests, step time with / without, dispatch sequence
TAG=${1:-r05o}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -X faulthandler -m pytest tests/test_nuthkaab_gpu.py tests/test_variogram_gpu.py -q -m gpu --maxfail=8 -k "routes or lean or onepass or named_binning or C3 or dilating" -p no:cacheprovider > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-300
grep -E "^E  " $O/pytest.log | head -20 | cut -c1-250
timeout 300 python -u tools/nk_trace.py 20000 6 > $O/nk_default.log 2>&1; grep -E "step|routes" $O/nk_default.log | tail -8
XDEM_NK_BINSEG=0 timeout 300 python -u tools/nk_trace.py 20000 6 > $O/nk_old.log 2>&1; grep -E "step|routes" $O/nk_old.log | tail -8
timeout 300 python -u tools/nk_trace.py 20000 6 > $O/nk_default2.log 2>&1; grep -E "step|routes" $O/nk_default2.log | tail -8
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/nktrace -o nk -- python $R/tools/nk_trace.py 20000 3 > $R/$O/nktrace.log 2>&1 )
python tools/trace_sequence.py $O/nktrace 40 > $O/nk_sequence.txt 2>&1; tail -42 $O/nk_sequence.txt | cut -c1-120
find $O -name '*.csv' -size +2M -delete
