#!/bin/bash
# round 6, session q: LDS-DMA ring in four 8-row blocks (refills 19 rows ahead) against two 16-row halves (11 rows ahead): every set, both forms, one process;
# terrain GPU tests (incl. the counted-wait check under terrain_ring_wait = 1)
O=gpurun_out/r06q; mkdir -p $O
export PYTHONUNBUFFERED=1
LIBS="rb16=xdem_amd/csrc/libxdemhip_exprb16.so rb8=xdem_amd/csrc/libxdemhip_exprb8.so new=xdem_amd/csrc/libxdemhip.so"
for m in 1 3 7; do echo "mask $m (Florinsky)"; timeout 300 python tools/ab_libs.py --reps 6 --rounds 3 --mask $m $LIBS > $O/ab_m$m.txt 2>&1; grep -E "planes\]|vs" $O/ab_m$m.txt; done
echo "mask 3 (Horn)"; timeout 300 python tools/ab_libs.py --reps 6 --rounds 3 --mask 3 --fit 0 $LIBS > $O/ab_m3_horn.txt 2>&1; grep -E "planes\]|vs" $O/ab_m3_horn.txt
echo "mask 4 hillshade"; timeout 300 python tools/ab_libs.py --reps 6 --rounds 3 --mask 4 $LIBS > $O/ab_m4.txt 2>&1; grep -E "planes\]|vs" $O/ab_m4.txt
echo "full 11"; timeout 600 python tools/ab_libs.py --planes both --reps 6 --rounds 3 $LIBS > $O/ab_full.txt 2>&1; grep -E "planes\]|vs" $O/ab_full.txt
echo "full 11 ZT"; timeout 600 python tools/ab_libs.py --planes scattered --reps 6 --rounds 2 --fit 1 $LIBS > $O/ab_zt.txt 2>&1; grep -E "planes\]|vs" $O/ab_zt.txt
timeout 1500 python -m pytest tests/test_terrain_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_terrain.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_terrain.log | cut -c1-200
