#!/bin/bash
# round 6, session av: the allocator's calibration of both plane placements (terrain.alloc_planes(probe=...)) in the bench line -- after a part of the GPU suite
# (the state of the box's free memory after other processes is what flips the better placement), alone, and under the same-process kernel trace
O=gpurun_out/r06av; mkdir -p $O/same
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_terrain_gpu.py tests/test_variogram_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_part.log 2>&1; tail -1 $O/pytest_part.log
show() { python - "$1" <<'PY'
import json, sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("frac", r["frac"], "kernel_ms", r["kernel_ms"], "| planes", r.get("planes"), "| caller(torch)", r.get("frac_caller_planes"), "scattered", r.get("frac_scattered_planes"), "| first/min/max", r["kernel_ms_series"][0], r["kernel_ms_min"], r["kernel_ms_max"])
print("workload:", d["config"]["workload"][-150:])
PY
}
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-end-to-end > $O/line$i.json 2>> $O/bench.err; echo "bench rc=$?"; show $O/line$i.json
done
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/same -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-end-to-end > $R/$O/same/bench.log 2> $R/$O/same/bench.err )
python tools/bench_same_process.py $O/same $O/r06av_bench_same_process.json > $O/same_summary.txt 2>&1; grep -E "frac|over_bench|backing" $O/same_summary.txt | head -20
find $O -name '*.csv' -size +2M -delete; find $O -name "*.db" -delete
