#!/bin/bash
# round 6, session b: forward-accumulated Florinsky marcher at four waves per SIMD against the round-5 library (A/B in one process), terrain
# GPU parity tests, lag of the sysfs metrics behind the load
O=gpurun_out/r06b; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python tools/ab_libs.py --reps 8 --rounds 3 r05=xdem_amd/csrc/libxdemhip_r05.so new=xdem_amd/csrc/libxdemhip.so > $O/ab_full11.txt 2>&1; echo "ab rc=$?"; grep -v "^\[" $O/ab_full11.txt | tail -20
timeout 300 python tools/ab_libs.py --reps 6 --rounds 2 --fit 1 r05=xdem_amd/csrc/libxdemhip_r05.so new=xdem_amd/csrc/libxdemhip.so > $O/ab_zt.txt 2>&1; grep -v "^\[" $O/ab_zt.txt | tail -4
for m in 1 3 7; do timeout 300 python tools/ab_libs.py --reps 6 --rounds 2 --mask $m r05=xdem_amd/csrc/libxdemhip_r05.so new=xdem_amd/csrc/libxdemhip.so > $O/ab_m$m.txt 2>&1; echo "mask $m"; grep -v "^\[" $O/ab_m$m.txt | tail -4; done
timeout 120 python tools/sampler_lag_probe.py 5 > $O/sampler_lag.txt 2>&1; tail -80 $O/sampler_lag.txt
timeout 1500 python -m pytest tests/test_terrain_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_terrain.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_terrain.log | cut -c1-300
