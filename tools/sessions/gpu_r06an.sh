#!/bin/bash
# round 6, session an: SURVEY 8d's C5 reading A end to end on the end-state library (subsample 1e7, one variogram, host raster in)
O=gpurun_out/r06an; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python tools/probes/vario_e2e_probe.py 20000 10000000 1 > $O/e2e_1e7.txt 2>&1; grep -E "wall|equidistant_blocks|_native|__init__|empirical_variogram_pairs|class_medians|sums" $O/e2e_1e7.txt | cut -c1-200
