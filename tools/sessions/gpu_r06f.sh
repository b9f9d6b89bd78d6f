#!/bin/bash
# round 6, session f: predicted brackets of the settled Nuth-Kaab step -- the new test, the NK suite, step times and the dispatch sequence
# of settled steps with and without prediction; piece sizes once more (whatever box this is)
O=gpurun_out/r06f; mkdir -p $O
export PYTHONUNBUFFERED=1
XDEMHIP_DEBUG=1 timeout 600 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -x -s -k "predicted_brackets" > $O/pytest_pred.log 2>&1; echo "pred rc=$?"; grep -E "routes with|passed|failed|Error|assert" $O/pytest_pred.log | cut -c1-300 | tail -12; grep -E "one-pass step \(" $O/pytest_pred.log | cut -c1-220 | head -40
NK_SETTLED=1 XDEMHIP_DEBUG=1 timeout 200 python -u tools/nk_trace.py 20000 6 > $O/steps_settled_pred.log 2>&1; grep -E "step 20000|routes|PREDICTED|falls" $O/steps_settled_pred.log | cut -c1-220
NK_SETTLED=1 NK_PREDICT=0 timeout 200 python -u tools/nk_trace.py 20000 6 > $O/steps_settled_sampled.log 2>&1; grep -E "step 20000|routes" $O/steps_settled_sampled.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
NK_SETTLED=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_pred -o pred -- python -u $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 3 > $GRAFT_REPO_ROOT/$O/trace_pred.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python tools/trace_sequence.py $O/trace_pred 40 > $O/sequence_pred.txt 2>&1; tail -45 $O/sequence_pred.txt | cut -c1-150
find $O -name '*.csv' -size +3M -delete
timeout 1500 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_nk.log 2>&1; echo "nk suite rc=$?"; tail -4 $O/pytest_nk.log | cut -c1-300
timeout 600 python tools/piece_probe.py --pieces 8,128 > $O/piece_probe.txt 2>&1; grep -v "^/opt" $O/piece_probe.txt | tail -10
