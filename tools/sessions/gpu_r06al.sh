#!/bin/bash
# round 6, session al: infer_heteroscedasticity_from_stable without the boolean-index copies when no mask is given
O=gpurun_out/r06al; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_binning_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_binning.log 2>&1; echo "binning tests rc=$?"; tail -3 $O/pytest_binning.log | cut -c1-300
timeout 1200 python -u tools/probes/e2e_calls_probe.py 12000 > $O/e2e_calls.log 2>&1; echo "probe rc=$?"; grep -E "^\[" $O/e2e_calls.log | cut -c1-200
