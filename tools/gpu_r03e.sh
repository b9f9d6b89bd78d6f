#!/bin/bash
# round 3, session E: lean tail x streaming route -- parity (terrain GPU suite), then A/B timing
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03e}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_terrain_gpu.py -x -q > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 600 python tools/terrain_opts_bench.py --size 40000 --reps 4 --rounds 3 --opts "" --combos "terrain_math=0+terrain_stream=0;terrain_math=0+terrain_stream=1;terrain_math=2+terrain_stream=0;terrain_math=2+terrain_stream=1;terrain_math=2+terrain_stream=256" --json $OUT/opts.json > $OUT/opts.log 2>&1
grep -v amdgpu.ids $OUT/opts.log
timeout 300 python tools/ulp_report.py --gpu --size 6000 --tail 2 > $OUT/ulp_lean.txt 2>&1
grep -v amdgpu.ids $OUT/ulp_lean.txt
