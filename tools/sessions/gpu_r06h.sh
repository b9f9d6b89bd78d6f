#!/bin/bash
# round 6, session h: which step of bench.py's Nuth-Kaab leg missed its predicted bracket; float64 hooked test; variogram tests after the
# 'even' / shadow-decision changes; concurrency test after the context call lock
O=gpurun_out/r06h; mkdir -p $O
export PYTHONUNBUFFERED=1
XDEMHIP_DEBUG=1 timeout 300 python -u tools/nk_fit_debug.py > $O/nk_fit_debug.log 2>&1; grep -E "one-pass step|routes|settled step|falls" $O/nk_fit_debug.log | cut -c1-230
timeout 600 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -x -k "hooked_plan" > $O/pytest_hooked.log 2>&1; echo "hooked rc=$?"; tail -4 $O/pytest_hooked.log | cut -c1-300
timeout 1500 python -m pytest tests/test_variogram_gpu.py tests/test_concurrency_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_vario.log 2>&1; echo "vario rc=$?"; tail -5 $O/pytest_vario.log | cut -c1-300
