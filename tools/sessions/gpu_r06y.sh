#!/bin/bash
# round 6, session y: the sums of the one-pass step added in a fixed order (two runs of a fit bit for bit); histogram passes with staggered flushes and loads issued ahead (three runs: r06y, r06y2 twice)
O=gpurun_out/r06y; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 200 python -u tools/nk_trace.py 20000 6 > $O/steps_sampled.log 2>&1; grep -E "step 20000|routes" $O/steps_sampled.log | cut -c1-200
NK_PREDICT=0 timeout 200 python -u tools/nk_trace.py 20000 6 > $O/steps_sampled_nopredict.log 2>&1; grep -E "step 20000|routes" $O/steps_sampled_nopredict.log | cut -c1-160
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python -u $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 3 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python tools/trace_sequence.py $O/trace 21 > $O/sequence.txt 2>&1; tail -24 $O/sequence.txt | cut -c1-150
find $O -name '*.csv' -size +3M -delete
timeout 1500 python -m pytest tests/test_nuthkaab_gpu.py tests/test_binning_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_nk.log 2>&1; echo "nk+binning suite rc=$?"; tail -4 $O/pytest_nk.log | cut -c1-300
