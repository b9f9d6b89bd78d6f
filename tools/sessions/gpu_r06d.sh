#!/bin/bash
# round 6, session d: the three builds of session c once more, this time without the product loader in the process (ab_libs.py after its fix)
O=gpurun_out/r06d; mkdir -p $O
export PYTHONUNBUFFERED=1
LIBS="r05=xdem_amd/csrc/libxdemhip_r05.so fw=xdem_amd/csrc/libxdemhip_fw.so new=xdem_amd/csrc/libxdemhip.so"
timeout 900 python tools/ab_libs.py --planes both --reps 8 --rounds 3 $LIBS > $O/ab_full11.txt 2>&1; echo "ab rc=$?"; grep -v "^/opt" $O/ab_full11.txt | tail -20
timeout 600 python tools/ab_libs.py --planes both --reps 6 --rounds 2 --curv 1 $LIBS > $O/ab_dir.txt 2>&1; grep -v "^/opt" $O/ab_dir.txt | tail -8
timeout 600 python tools/ab_libs.py --planes both --reps 6 --rounds 2 --fit 1 $LIBS > $O/ab_zt.txt 2>&1; grep -v "^/opt" $O/ab_zt.txt | tail -8
