#!/bin/bash
# round-4 session 8: run-length counting pass of the exact Dowd selection -- parity tests, then A/B of the three forms
O=gpurun_out/r04k; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_variogram_gpu.py -x -q -m gpu > $O/pytest_vario.log 2>&1; tail -3 $O/pytest_vario.log
XDEMHIP_DEBUG=1 timeout 300 python tools/vario_runs_probe.py 9091 100 > $O/runs_probe_b.txt 2> $O/runs_probe_b.err; cat $O/runs_probe_b.txt
grep -E "^----|counting|candidates \(" $O/runs_probe_b.err | head -60
