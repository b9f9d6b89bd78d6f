#!/bin/bash
# rocprofv3 evidence for the one data pass of the Nuth-Kaab step (nk_fused_kernel) on bench.py's C3 pair: kernel trace + the PMC passes of
# tools/profile_bench.sh (FETCH / WRITE / GRBM / SQ, one pass each), summarised per launch by tools/summarize_pmc.py.
#   bash tools/profile_nk.sh <tag> [size=20000]      (from the repo root, through gpurun)
set -u
TAG=${1:-r06}
SIZE=${2:-20000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_nk_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NK_SETTLED=1
CMD="python $GRAFT_REPO_ROOT/tools/nk_trace.py $SIZE 4"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o nk -- $CMD > $OUT/stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o nk -- $CMD > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o nk -- $CMD > $OUT/write.log 2>&1
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm -o nk -- $CMD > $OUT/grbm.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/sq -o nk -- $CMD > $OUT/sq.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq2 -o nk -- $CMD > $OUT/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_pmc.py $OUT $OUT/${TAG}_nk_fused nk_fused_kernel $((SIZE*SIZE)) > $OUT/summary.log 2>&1
find $OUT -name '*.csv' -size +3M -delete
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f | cut -c1-200; done
cat $OUT/summary.log | tail -5
