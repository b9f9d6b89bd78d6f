#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run from the repo root through gpurun):
#   1. --kernel-trace --stats   -> per-kernel durations (CSV)
#   2. --pmc FETCH_SIZE         -> HBM read bytes  (own pass: TCC slots)
#   3. --pmc WRITE_SIZE         -> HBM write bytes (own pass)
#   4. --pmc SQ_* VALU counters -> issue mix
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries worth keeping into profiles/.
# NB at most 8 SQ counters per --pmc pass on gfx950 (more: rocprofv3 aborts with "exceeds the capabilities of the hardware" and
# then hangs while finalising) -- and run every rocprofv3 command under `timeout`.
set -u
TAG=${1:-r01}
SIZE=${2:-40000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-end-to-end --size $SIZE"
# the stats pass also runs the secondary workloads (variogram C5-B, Nuth-Kaab C3) so their kernels show up in the summary
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --size $SIZE > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- $CMD > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm -o bench -- $CMD > $OUT/grbm.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/sq -o bench -- $CMD > $OUT/sq.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_pmc.py $OUT $OUT/${TAG}_bench_terrain terrain_strip_kernel,terrain_tile_kernel $((SIZE*SIZE)) > $OUT/summary.log 2>&1
find $OUT -name '*.csv' -size +3M -delete
find $OUT -type f | head -40
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
tail -1 $OUT/stats.log
