#!/bin/bash
# round 6, session aa: xdemhip_nk_subsample -- nuth_kaab's random subsample as ranks drawn on the host and turned into pixels on the device
O=gpurun_out/r06aa; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -x -k "subsample or class_api or full_fit" > $O/pytest_sub.log 2>&1; echo "subsample tests rc=$?"; tail -5 $O/pytest_sub.log | cut -c1-300
timeout 600 python -u tools/probes/nk_e2e_probe.py 20000 > $O/nk_e2e.log 2>&1; echo "probe rc=$?"; grep -E "^\[" $O/nk_e2e.log | cut -c1-250
timeout 1500 python -m pytest tests/test_nuthkaab_gpu.py tests/test_cabi_and_host.py -q -m gpu -p no:cacheprovider -x > $O/pytest_nk.log 2>&1; echo "nk suite rc=$?"; tail -4 $O/pytest_nk.log | cut -c1-300
