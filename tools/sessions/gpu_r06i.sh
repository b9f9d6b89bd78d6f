#!/bin/bash
# round 6, session i: the tightened prediction rule on bench.py's Nuth-Kaab sequence + tests; what the strip kernel's ring refills and its
# stores cost (measurement builds: no refills after the first two blocks / stores compiled out)
O=gpurun_out/r06i; mkdir -p $O
export PYTHONUNBUFFERED=1
XDEMHIP_DEBUG=1 timeout 300 python -u tools/nk_fit_debug.py > $O/nk_fit_debug.log 2>&1; grep -E "one-pass step \(|routes|settled step|falls" $O/nk_fit_debug.log | cut -c1-230
timeout 900 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -x -k "hooked_plan or predicted or whole_fit" > $O/pytest_nk.log 2>&1; echo "nk rc=$?"; tail -4 $O/pytest_nk.log | cut -c1-300
timeout 900 python -m pytest tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider -x -k "one_pass" > $O/pytest_dist.log 2>&1; echo "dist rc=$?"; tail -4 $O/pytest_dist.log | cut -c1-300
LIBS="new=xdem_amd/csrc/libxdemhip.so norefill=xdem_amd/csrc/libxdemhip_expnr.so nostore=xdem_amd/csrc/libxdemhip_expnostore.so r05=xdem_amd/csrc/libxdemhip_r05.so"
timeout 900 python tools/ab_libs.py --planes both --reps 8 --rounds 2 $LIBS > $O/ab_variants.txt 2>&1; echo "ab rc=$?"; grep -v "^/opt" $O/ab_variants.txt | tail -14
timeout 600 python tools/ab_libs.py --planes scattered --reps 6 --rounds 2 --curv 1 $LIBS > $O/ab_variants_dir.txt 2>&1; grep -v "^/opt" $O/ab_variants_dir.txt | tail -8
