#!/bin/bash
# round-5, session w: dispatch sequence of the one-pass step on a hooked plan (1-rank RCCL group, device-side hook) next to the hook-less one,
# C3 pair; un-profiled step times of both first
TAG=${1:-r05w}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
NK_HOOKED=1 timeout 200 python -u tools/nk_trace.py 20000 4 > $O/steps_hooked.log 2>&1; grep -E "step|routes" $O/steps_hooked.log | cut -c1-200
timeout 200 python -u tools/nk_trace.py 20000 4 > $O/steps_plain.log 2>&1; grep -E "step|routes" $O/steps_plain.log | cut -c1-200
NK_HOOKED=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_hooked -o hooked -- python -u tools/nk_trace.py 20000 2 > $O/trace_hooked.log 2>&1; echo "trace rc=$?"
python tools/trace_sequence.py $O/trace_hooked 64 > $O/sequence_hooked.txt 2>&1; tail -70 $O/sequence_hooked.txt | cut -c1-150
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
