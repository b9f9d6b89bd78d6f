#!/bin/bash
# round 6, session ad: pair sets of sum-only estimators without the caller-order copy
O=gpurun_out/r06ad; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python tools/probes/vario_e2e_probe.py 20000 1000000 1 > $O/e2e_1e6_one.txt 2>&1; grep -E "wall" $O/e2e_1e6_one.txt | cut -c1-200
timeout 900 python tools/probes/vario_e2e_probe.py 20000 1000000 10 > $O/e2e_1e6.txt 2>&1; grep -E "wall" $O/e2e_1e6.txt | cut -c1-200
timeout 1500 python -m pytest tests/test_variogram_gpu.py tests/test_binning_gpu.py tests/test_patches_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_vario.log 2>&1; echo "vario suite rc=$?"; tail -3 $O/pytest_vario.log | cut -c1-300
