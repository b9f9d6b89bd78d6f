#!/bin/bash
O=gpurun_out/r04t; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 400 python tools/vmm_stale_probe.py 20 > $O/stale_probe_fixed.txt 2>&1; tail -4 $O/stale_probe_fixed.txt | cut -c1-250
timeout 600 python -m pytest tests/test_terrain_gpu.py -x -q -m gpu > $O/pytest_terrain.log 2>&1; tail -2 $O/pytest_terrain.log | cut -c1-200
