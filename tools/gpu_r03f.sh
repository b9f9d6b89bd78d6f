#!/bin/bash
# round 3, session F: clocks and SQ counters of the terrain kernel variants (tile / strip x mixed / lean tail)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03f}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/terrain_opts_bench.py --size 40000 --reps 3 --rounds 1 --opts= --combos terrain_math=0+terrain_stream=0;terrain_math=0+terrain_stream=1;terrain_math=2+terrain_stream=0;terrain_math=2+terrain_stream=1"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm -o v -- $CMD > $OUT/grbm.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq -o v -- $CMD > $OUT/sq.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_by_kernel.py $OUT terrain_ > $OUT/by_kernel.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/by_kernel.txt | head -120
