#!/bin/bash
# round-4 session 9: one-pass Nuth-Kaab step with narrowed sample brackets -- parity, then wall time per setting
O=gpurun_out/r04l; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_nuthkaab_gpu.py -x -q -m gpu -s -k "lean or routes or degenerate or ext_route" > $O/pytest_nk.log 2>&1; tail -5 $O/pytest_nk.log; grep "nk_narrow" $O/pytest_nk.log
for k in 0 1 2 -1; do
  NK_NARROW=$k XDEMHIP_DEBUG=1 timeout 200 python tools/nk_trace.py 20000 6 > $O/trace_n$k.txt 2> $O/trace_n$k.err
  echo "== nk_narrow $k"; grep -E "step 2|routes" $O/trace_n$k.txt; grep "one-pass step" $O/trace_n$k.err | tail -3
done
