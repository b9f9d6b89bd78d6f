#!/bin/bash
# NK step A/B (previous build vs current) + the GPU tests of every user of the shared selection code
O=gpurun_out/r03nk; mkdir -p $O
for lib in libxdemhip_base.so libxdemhip.so libxdemhip_base.so libxdemhip.so; do
  echo "== $lib" >> $O/nk_ab.log
  NK_LIB=$PWD/xdem_amd/csrc/$lib timeout 200 python -u tools/nk_probe.py 20000 6 2>/dev/null | tail -4 >> $O/nk_ab.log
done
cat $O/nk_ab.log
timeout 1500 python -X faulthandler -m pytest tests/test_nuthkaab_gpu.py tests/test_binning_gpu.py tests/test_variogram_gpu.py -x -q -m gpu > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
