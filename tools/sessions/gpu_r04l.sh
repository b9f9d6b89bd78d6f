#!/bin/bash
# round-4 session 12: run-length counting pass with partial A tiles and 32-pair flushes
O=gpurun_out/r04o; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_variogram_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -m pytest tests/test_nuthkaab_gpu.py -x -q -m gpu -k "cost_little or degenerate" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
timeout 300 python tools/vario_runs_probe.py 9091 100 > $O/runs_probe_b.txt 2> $O/runs_probe_b.err; cat $O/runs_probe_b.txt
