#!/bin/bash
# round-4 session 13: spread design effect 4, cached selection blocks -- parity, offsets of the brackets, timings (readings B and A)
O=gpurun_out/r04p; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_variogram_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
XDEMHIP_DEBUG=1 timeout 300 python tools/vario_c5_probe.py 16 4 1 0 > $O/deff_probe.txt 2> $O/deff_probe.err; cat $O/deff_probe.txt
grep -E "^----|wanted rank|candidates \(|attempt" $O/deff_probe.err | head -40
PROBE_CFG=0,2 timeout 300 python tools/vario_runs_probe.py 9091 100 > $O/runs_probe_b.txt 2>&1; cat $O/runs_probe_b.txt
