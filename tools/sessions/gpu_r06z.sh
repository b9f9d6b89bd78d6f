#!/bin/bash
# round-6, session z: end state -- the bench line and the rocprofv3 kernel trace OF THE SAME PROCESS (tools/bench_same_process.py), terrain only
TAG=${1:-r06z}
O=gpurun_out/$TAG; mkdir -p $O/same
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/same -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-end-to-end > $R/$O/same/bench.log 2> $R/$O/same/bench.err )
python tools/bench_same_process.py $O/same $O/${TAG}_bench_same_process.json > $O/same_summary.txt 2>&1; cat $O/same_summary.txt | cut -c1-200
for f in $(find $O/same -name "*kernel_stats.csv"); do cp $f $O/${TAG}_same_process_kernel_stats.csv; done
find $O -name '*.csv' -size +2M -delete; find $O -name "*.db" -delete
# ... and the closing rocprofv3 set of the round on the end state (kernel stats + PMC passes of the bench command)
timeout 2400 bash tools/profile_bench.sh r06z 40000 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -12 $O/profile.log | cut -c1-200
