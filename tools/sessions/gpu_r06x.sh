#!/bin/bash
# round 6, session x: sample brackets in sixteenths of the rule (4.8 sigma), one sample kernel on steps with a predicted bracket of the median of dh
O=gpurun_out/r06x; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_nk.log 2>&1; echo "nk suite rc=$?"; tail -4 $O/pytest_nk.log | cut -c1-300
timeout 1200 python -m pytest tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider -x -k "nuth or nk" > $O/pytest_dist_nk.log 2>&1; echo "dist nk rc=$?"; tail -4 $O/pytest_dist_nk.log | cut -c1-300
XDEMHIP_DEBUG=1 timeout 200 python -u tools/nk_trace.py 20000 6 > $O/steps_sampled.log 2>&1; grep -E "step 20000|routes|brackets x|falls" $O/steps_sampled.log | cut -c1-200
NK_PREDICT=0 timeout 200 python -u tools/nk_trace.py 20000 6 > $O/steps_sampled_nopredict.log 2>&1; grep -E "step 20000|routes" $O/steps_sampled_nopredict.log | cut -c1-160
NK_SETTLED=1 timeout 200 python -u tools/nk_trace.py 20000 6 > $O/steps_settled.log 2>&1; grep -E "step 20000|routes" $O/steps_settled.log | cut -c1-160
XDEMHIP_DEBUG=1 timeout 300 python -u tools/nk_fit_debug.py > $O/fit_debug.log 2>&1; grep -E "one-pass step \(|routes|settled step" $O/fit_debug.log | cut -c1-220
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python -u $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 3 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python tools/trace_sequence.py $O/trace 21 > $O/sequence.txt 2>&1; tail -24 $O/sequence.txt | cut -c1-150
find $O -name '*.csv' -size +3M -delete
