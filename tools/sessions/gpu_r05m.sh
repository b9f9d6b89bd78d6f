#!/bin/bash
# round-5, session m: the median of dh among its candidates in two launches (nk_dhsel_*), the bad-bit mask for row blocks (rules 2 / 3),
# bin_func='even': the Nuth-Kaab and partitioned tests, the step's time with and without, the dispatch sequence
TAG=${1:-r05m}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -X faulthandler -m pytest tests/test_nuthkaab_gpu.py tests/test_dist_gpu.py tests/test_variogram_gpu.py -q -m gpu --maxfail=8 -k "nuthkaab or nuth_kaab or named_binning or partitioned or sharded" -p no:cacheprovider > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300
grep -E "^E  " $O/pytest.log | head -20 | cut -c1-250
timeout 300 python -u tools/nk_trace.py 20000 6 > $O/nk_default.log 2>&1; grep -E "step|routes" $O/nk_default.log | tail -8
XDEM_NK_BINSEG=0 timeout 300 python -u tools/nk_trace.py 20000 6 > $O/nk_old.log 2>&1; grep -E "step|routes" $O/nk_old.log | tail -8
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/nktrace -o nk -- python $R/tools/nk_trace.py 20000 3 > $R/$O/nktrace.log 2>&1 )
python tools/trace_sequence.py $O/nktrace 40 > $O/nk_sequence.txt 2>&1; tail -44 $O/nk_sequence.txt | cut -c1-120
find $O -name '*.csv' -size +2M -delete
