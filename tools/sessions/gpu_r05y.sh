#!/bin/bash
# round-5, session y: after the 1024-thread forms of the per-bin exchange kernels -- whole GPU suite (no -x), the hooked step's times and
# dispatch sequence, the shared-GPU probe
TAG=${1:-r05y}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 560 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
tail -6 $O/pytest_all.log | cut -c1-300
NK_HOOKED=1 timeout 150 python -u tools/nk_trace.py 20000 4 > $O/steps_hooked.log 2>&1; grep -E "step|routes" $O/steps_hooked.log | cut -c1-200
timeout 150 python -u tools/nk_trace.py 20000 4 > $O/steps_plain.log 2>&1; grep -E "step|routes" $O/steps_plain.log | cut -c1-200
timeout 200 python -u tools/nk_dist_probe.py 20000 2 5 > $O/nk_dist_probe.log 2>&1; echo "probe rc=$?"; grep "^\[20000" $O/nk_dist_probe.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
NK_HOOKED=1 timeout 150 rocprofv3 --kernel-trace --output-format csv -d $O/trace_hooked -o hooked -- python -u tools/nk_trace.py 20000 2 > $O/trace_hooked.log 2>&1; echo "trace rc=$?"
python tools/trace_sequence.py $O/trace_hooked 64 > $O/sequence_hooked.txt 2>&1; tail -66 $O/sequence_hooked.txt | cut -c1-120
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
