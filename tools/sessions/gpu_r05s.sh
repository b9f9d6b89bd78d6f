#!/bin/bash
# round-5, session s: stress of the one-pass step's hand-overs (tools/nk_stress.py), then bench.py (the line of the round's end state)
TAG=${1:-r05s}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -u tools/nk_stress.py 12000 400 > $O/nk_stress.log 2>&1; echo "rc=$?" >> $O/nk_stress.log; tail -5 $O/nk_stress.log
timeout 600 python -u tools/nk_stress.py 20000 150 > $O/nk_stress_c3.log 2>&1; echo "rc=$?" >> $O/nk_stress_c3.log; tail -5 $O/nk_stress_c3.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; tail -c 6000 $O/bench.log
