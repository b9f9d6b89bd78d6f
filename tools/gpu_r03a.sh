#!/bin/bash
# round 3, session A: memory-pattern ceilings, VALU rates, runtime options of the terrain kernel, the whole GPU suite
# (with the new C5 lattice x bracket parity test) and a kernel trace of that test.
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 120 tools/ubench > $OUT/ubench.log 2>&1
timeout 300 tools/membench3 40000 > $OUT/membench3.log 2>&1
timeout 600 python tools/terrain_opts_bench.py --size 40000 --reps 4 --rounds 3 --json $OUT/opts.json > $OUT/opts.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5trace -o c5 -- python -m pytest $GRAFT_REPO_ROOT/tests/test_variogram_gpu.py -q -x -k C5_sampler > $OUT/c5trace.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/c5trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c5_kernel_stats.csv
find $OUT/c5trace -name "*.csv" ! -name "*stats*" -size +1M -delete
timeout 300 python bench.py --steps 10 --warmup 3 > $OUT/bench.log 2>&1
tail -c 3000 $OUT/opts.log; tail -40 $OUT/membench3.log
