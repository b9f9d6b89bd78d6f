#!/bin/bash
# round-4 session 10: run-length counting pass after its diet, dual first digit from one histogram, merged bracket kernel
O=gpurun_out/r04m; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_variogram_gpu.py tests/test_binstats_gpu.py tests/test_nuthkaab_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
XDEMHIP_DEBUG=1 timeout 300 python tools/vario_runs_probe.py 9091 100 > $O/runs_probe_b.txt 2> $O/runs_probe_b.err; cat $O/runs_probe_b.txt
for k in 0 -1; do
  NK_NARROW=$k XDEMHIP_DEBUG=1 timeout 200 python tools/nk_trace.py 20000 6 > $O/trace_n$k.txt 2> $O/trace_n$k.err
  echo "== nk_narrow $k"; grep -E "step 2|routes" $O/trace_n$k.txt; grep "one-pass step" $O/trace_n$k.err | tail -2
done
