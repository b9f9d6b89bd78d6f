#!/bin/bash
# round 6, session at: nd_binning with any statistic (host half on bin numbers from the device), the NuthKaab callable route on the shared helper
O=gpurun_out/r06at; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_binning_gpu.py tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log | cut -c1-250
