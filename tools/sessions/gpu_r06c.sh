#!/bin/bash
# round 6, session c: three builds in one process on both plane backings (r05 = round 5's library; fw = forward-accumulated Florinsky marcher at
# four waves per SIMD; new = fw + degrees-direct slope / aspect, scalar cold-path test), the s_sleep clock probe, terrain GPU tests
O=gpurun_out/r06c; mkdir -p $O
export PYTHONUNBUFFERED=1
LIBS="r05=xdem_amd/csrc/libxdemhip_r05.so fw=xdem_amd/csrc/libxdemhip_fw.so new=xdem_amd/csrc/libxdemhip.so"
timeout 900 python tools/ab_libs.py --planes both --reps 8 --rounds 3 $LIBS > $O/ab_full11.txt 2>&1; echo "ab rc=$?"; grep -v "^/opt" $O/ab_full11.txt | tail -20
timeout 600 python tools/ab_libs.py --planes both --reps 6 --rounds 2 --curv 1 $LIBS > $O/ab_dir.txt 2>&1; grep -v "^/opt" $O/ab_dir.txt | tail -8
for m in 1 3 7; do timeout 300 python tools/ab_libs.py --reps 6 --rounds 2 --mask $m $LIBS > $O/ab_m$m.txt 2>&1; echo "mask $m"; grep -v "^/opt" $O/ab_m$m.txt | tail -5; done
timeout 120 tools/clock_probe > $O/clock_probe.txt 2>&1; cat $O/clock_probe.txt
timeout 1500 python -m pytest tests/test_terrain_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_terrain.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_terrain.log | cut -c1-300
