#!/bin/bash
# round 6, session u: the native host-side sampler (csrc/hostprep.hip): variogram GPU tests, then the API call end to end (host preparation
# included) at 20000^2, subsample 1e6 x 10 variograms (as before the change: 72 s) and the C5 size (subsample 1e7, one variogram)
O=gpurun_out/r06u; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_variogram_gpu.py tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_vario.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_vario.log | cut -c1-200
timeout 900 python tools/probes/vario_e2e_probe.py 20000 1000000 10 > $O/e2e_1e6.txt 2>&1; grep -E "wall|equidistant_blocks|_native|__init__|empirical_variogram_pairs|gather|isfinite" $O/e2e_1e6.txt | cut -c1-200
XDEM_HOST_THREADS=1 timeout 900 python tools/probes/vario_e2e_probe.py 20000 1000000 2 > $O/e2e_1e6_1thread.txt 2>&1; grep -E "wall" $O/e2e_1e6_1thread.txt | cut -c1-200
timeout 1500 python tools/probes/vario_e2e_probe.py 20000 10000000 1 > $O/e2e_1e7.txt 2>&1; grep -E "wall|equidistant_blocks|_native|__init__|empirical_variogram_pairs" $O/e2e_1e7.txt | cut -c1-200
