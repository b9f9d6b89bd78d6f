#!/bin/bash
# One GPU-box session of round 2 (run through gpurun): terrain parity on the HIP path, variant timing, ulp report.
set -x
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_terrain_gpu.py -m gpu -x -q -s > $OUT/pytest_terrain.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_terrain.log
tail -5 $OUT/pytest_terrain.log
timeout 600 python tools/variant_bench.py --size 40000 --reps 4 --rounds 2 --json $OUT/variants.json > $OUT/variants.log 2>&1
cat $OUT/variants.log
timeout 300 python tools/ulp_report.py --gpu --size 3000 --json $OUT/ulp_florinsky.json > $OUT/ulp.log 2>&1
tail -14 $OUT/ulp.log
