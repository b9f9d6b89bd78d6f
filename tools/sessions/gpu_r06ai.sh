#!/bin/bash
# round 6, session ai: few-plane terrain calls on host arrays -- direct copies + resident output pages against the staged pipeline
O=gpurun_out/r06ai; mkdir -p $O
export PYTHONUNBUFFERED=1
for n in 12000 30000; do
  XDEMHIP_HOST_DIRECT=0 timeout 600 python -u tools/probes/terrain_host_small_sets.py $n 2>&1 | grep "^\[" | tee -a $O/small_sets.txt | cut -c1-200
  timeout 600 python -u tools/probes/terrain_host_small_sets.py $n 2>&1 | grep "^\[" | tee -a $O/small_sets.txt | cut -c1-200
done
timeout 1500 python -m pytest tests/test_terrain_gpu.py tests/test_error_parity.py tests/test_concurrency_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_terrain.log 2>&1; echo "terrain tests rc=$?"; tail -3 $O/pytest_terrain.log | cut -c1-300
