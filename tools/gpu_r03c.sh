#!/bin/bash
# round 3, session C: streaming-strip terrain kernel -- parity test, then A/B timing against the tile kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03c}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_terrain_gpu.py -x -q -k "streaming or halo_rows or fbm_f32" > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 600 python tools/terrain_opts_bench.py --size 40000 --reps 4 --rounds 3 --opts "terrain_stream=0,1,256,512" --json $OUT/opts.json > $OUT/opts.log 2>&1
cat $OUT/opts.log | grep -v amdgpu.ids
