#!/bin/bash
# round-3 session X: A/B of the trimmed strip kernel against the previous build, terrain parity, new variogram test
O=gpurun_out/r03x; mkdir -p $O
timeout 300 python -u tools/ab_libs.py --reps 6 --rounds 3 base=xdem_amd/csrc/libxdemhip_base.so new=xdem_amd/csrc/libxdemhip.so > $O/ab.log 2>&1
cat $O/ab.log | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_terrain_gpu.py tests/test_variogram_gpu.py -x -q -m gpu -k "not C5_full and not c4_size" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
