#!/bin/bash
O=gpurun_out/r04w; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python tools/small_sets_probe.py > $O/small_sets.txt 2>&1; tail -6 $O/small_sets.txt | cut -c1-300
