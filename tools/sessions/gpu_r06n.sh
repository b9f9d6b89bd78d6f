#!/bin/bash
# round 6, session n: one-byte aspect-bin cache -- Nuth-Kaab suite, partitioned tests, settled / sampled step times
O=gpurun_out/r06n; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_nk.log 2>&1; echo "nk suite rc=$?"; tail -4 $O/pytest_nk.log | cut -c1-300
timeout 1200 python -m pytest tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider -x -k "nuth or nk or bench" > $O/pytest_dist_nk.log 2>&1; echo "dist nk rc=$?"; tail -4 $O/pytest_dist_nk.log | cut -c1-300
NK_SETTLED=1 timeout 200 python -u tools/nk_trace.py 20000 6 > $O/steps_settled.log 2>&1; grep -E "step 20000|routes" $O/steps_settled.log | cut -c1-160
timeout 200 python -u tools/nk_trace.py 20000 4 > $O/steps_sampled.log 2>&1; grep -E "step 20000|routes" $O/steps_sampled.log | cut -c1-160
cd /tmp && export TMPDIR=/tmp
NK_SETTLED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python -u $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 3 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python tools/trace_sequence.py $O/trace 14 > $O/sequence.txt 2>&1; tail -16 $O/sequence.txt | cut -c1-150
find $O -name '*.csv' -size +3M -delete
