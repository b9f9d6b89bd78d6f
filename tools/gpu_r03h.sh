#!/bin/bash
# round 3: the whole GPU suite, smoke(), and the default bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03h}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.log 2> $OUT/bench.err
tail -c 6000 $OUT/bench.log; tail -5 $OUT/bench.err
