#!/bin/bash
# round 6, session j: B points in polar Morton order against Cartesian Morton order (variogram legs of the bench, two processes); T9 full-fit test;
# piece sizes on whatever box this is; rocprofv3 passes of the bench command (kernel stats + PMC)
O=gpurun_out/r06j; mkdir -p $O
export PYTHONUNBUFFERED=1
XDEMHIP_DEBUG=1 timeout 400 python tools/vario_runs_probe.py 9091 100 > $O/vario_morton.log 2> $O/vario_morton.err; grep -E "pairs|run-length|per pair" $O/vario_morton.log | cut -c1-200
XDEM_VARIO_B_ORDER=polar XDEMHIP_DEBUG=1 timeout 400 python tools/vario_runs_probe.py 9091 100 > $O/vario_polar.log 2> $O/vario_polar.err; grep -E "pairs|run-length|per pair" $O/vario_polar.log | cut -c1-200
timeout 600 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -x -k "full_fit or class_api" > $O/pytest_t9.log 2>&1; echo "t9 rc=$?"; tail -3 $O/pytest_t9.log | cut -c1-200
timeout 600 python tools/piece_probe.py --pieces 8,32,128 > $O/piece_probe.txt 2>&1; grep -v "^/opt" $O/piece_probe.txt | tail -10
timeout 2400 bash tools/profile_bench.sh r06 40000 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -30 $O/profile.log | cut -c1-220
