#!/bin/bash
# is the regression test sensitive?  with the quarantine switched off (limit 0: freed ranges go straight back) it should fail
O=gpurun_out/r04x; mkdir -p $O
export PYTHONUNBUFFERED=1
XDEMHIP_VMM_QUARANTINE_TB=0 timeout 600 python -m pytest tests/test_terrain_gpu.py -x -q -m gpu -k "fresh_scattered" > $O/pytest_noq.log 2>&1; tail -3 $O/pytest_noq.log | cut -c1-300; grep -n "AssertionError" $O/pytest_noq.log | head -3 | cut -c1-200
