#!/bin/bash
O=gpurun_out/r04r; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_terrain_gpu.py -x -q -m gpu > $O/pytest_all.log 2>&1; tail -2 $O/pytest_all.log | cut -c1-200; grep -n "AssertionError: (" $O/pytest_all.log | cut -c1-1500
