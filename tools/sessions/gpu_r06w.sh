#!/bin/bash
# round 6, session w: dispatch sequence of a SAMPLED Nuth-Kaab step of the end state (steps that move by 0.1 px), before work on the sample kernels
O=gpurun_out/r06w; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 200 python -u tools/nk_trace.py 20000 4 > $O/steps_sampled.log 2>&1; grep -E "step 20000|routes" $O/steps_sampled.log | cut -c1-160
NK_PREDICT=0 timeout 200 python -u tools/nk_trace.py 20000 4 > $O/steps_sampled_nopredict.log 2>&1; grep -E "step 20000|routes" $O/steps_sampled_nopredict.log | cut -c1-160
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python -u $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 3 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python tools/trace_sequence.py $O/trace 30 > $O/sequence.txt 2>&1; tail -34 $O/sequence.txt | cut -c1-150
find $O -name '*.csv' -size +3M -delete
