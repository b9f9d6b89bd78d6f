#!/bin/bash
# round-5, session aa: A/B of the small Florinsky sets' NaN rule (libxdemhip_base.so = the window sum of their own, libxdemhip.so = the two
# derivative sums + the centre row's partial), one process, launches interleaved; then the terrain GPU tests
TAG=${1:-r05aa}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
for m in 1 3 4 7 4087; do
  echo "== mask $m (Florinsky)"; timeout 200 python tools/ab_libs.py --mask $m --fit 2 --reps 7 --rounds 3 base=xdem_amd/csrc/libxdemhip_base.so new=xdem_amd/csrc/libxdemhip.so 2>&1 | tail -4
done > $O/ab_small_sets.txt 2>&1
cat $O/ab_small_sets.txt
timeout 420 python -m pytest tests/test_terrain_gpu.py tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider > $O/pytest_terrain.log 2>&1; echo "pytest rc=$?" >> $O/pytest_terrain.log; tail -4 $O/pytest_terrain.log | cut -c1-300
