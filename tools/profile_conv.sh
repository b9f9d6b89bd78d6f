#!/bin/bash
# rocprofv3 evidence for the public convolution / per-bin lookup kernels (tools/probes/conv_probe.py at 8192^2): kernel trace + PMC passes
# (FETCH / WRITE / GRBM / SQ, one pass each), summarised per launch by tools/summarize_pmc.py.
#   bash tools/profile_conv.sh <tag> [size=8192]      (from the repo root, through gpurun)
set -u
TAG=${1:-r06}
SIZE=${2:-8192}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_conv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/probes/conv_probe.py $SIZE"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o conv -- $CMD > $OUT/stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o conv -- $CMD > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o conv -- $CMD > $OUT/write.log 2>&1
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm -o conv -- $CMD > $OUT/grbm.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/sq -o conv -- $CMD > $OUT/sq.log 2>&1
cd $GRAFT_REPO_ROOT
# (kernel-name substrings without commas: summarize_pmc.py splits its pattern argument at commas)
for kn in " 5>(float const*:window_5x5" " 3>(float const*:window_3x3" "perbin_kernel<true:perbin"; do
  k="${kn%%:*}"; n="${kn##*:}"
  python tools/summarize_pmc.py $OUT $OUT/${TAG}_conv_$n "$k" $((SIZE*SIZE)) > $OUT/summary_$n.log 2>&1; tail -4 $OUT/summary_$n.log | cut -c1-300
done
find $OUT -name '*.csv' -size +3M -delete
