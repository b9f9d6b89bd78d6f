#!/bin/bash
set -x
TAG=${1:-r02c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_terrain_gpu.py -m gpu -x -q > $OUT/pytest_terrain.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_terrain.log
tail -3 $OUT/pytest_terrain.log
timeout 600 python tools/variant_bench.py --size 40000 --reps 4 --rounds 2 --json $OUT/variants.json > $OUT/variants.log 2>&1
cat $OUT/variants.log
cd /tmp
for V in MIX/store0/rows32 MIX/store1/rows16; do
  N=$(echo $V | tr '/' '_')
  CMD="python $GRAFT_REPO_ROOT/tools/variant_bench.py --size 40000 --reps 3 --rounds 1 --only $V"
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/sqa_$N -o v -- $CMD > $OUT/sqa_$N.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm_$N -o v -- $CMD > $OUT/grbm_$N.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT
