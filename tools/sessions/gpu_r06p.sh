#!/bin/bash
# round 6, session p: what bounds the SMALL attribute sets -- the streaming kernel, the same without ring refills, the same with stores compiled out
O=gpurun_out/r06p; mkdir -p $O
export PYTHONUNBUFFERED=1
LIBS="new=xdem_amd/csrc/libxdemhip.so norefill=xdem_amd/csrc/libxdemhip_expnr.so nostore=xdem_amd/csrc/libxdemhip_expnostore.so"
for m in 1 3 7; do echo "mask $m (Florinsky)"; timeout 300 python tools/ab_libs.py --reps 6 --rounds 2 --mask $m $LIBS > $O/ab_m$m.txt 2>&1; grep "planes\]" $O/ab_m$m.txt; done
echo "mask 3 (Horn)"; timeout 300 python tools/ab_libs.py --reps 6 --rounds 2 --mask 3 --fit 0 $LIBS > $O/ab_m3_horn.txt 2>&1; grep "planes\]" $O/ab_m3_horn.txt
