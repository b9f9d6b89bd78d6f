#!/bin/bash
# round 6, session af: apply_translation -- prefaulted output pages + the all-NaN check on host threads
O=gpurun_out/r06af; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -u tools/probes/nk_e2e_probe.py 20000 > $O/nk_e2e.log 2>&1; echo "probe rc=$?"; grep -E "^\[" $O/nk_e2e.log | cut -c1-250
timeout 900 python -m pytest tests/test_nuthkaab_gpu.py tests/test_concurrency_gpu.py -q -m gpu -p no:cacheprovider -x -k "apply or translation or concurr or shift" > $O/pytest_apply.log 2>&1; echo "apply tests rc=$?"; tail -3 $O/pytest_apply.log | cut -c1-300
