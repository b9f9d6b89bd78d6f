#!/bin/bash
# round 6, session s: the frame of edge tiles on a low-priority side stream next to the strips (test switch terrain_frame): A/B in one process on both
# plane backings, the terrain GPU tests, the default bench line
O=gpurun_out/r06s; mkdir -p $O
export PYTHONUNBUFFERED=1
for p in scattered torch; do echo "== planes $p"; timeout 600 python tools/terrain_opts_bench.py --planes $p --reps 6 --rounds 4 --opts "terrain_frame=0,1" 2>&1 | tail -3; done | tee $O/frame_ab.txt
timeout 1500 python -m pytest tests/test_terrain_gpu.py tests/test_dist_gpu.py tests/test_concurrency_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_terrain.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_terrain.log | cut -c1-200
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r06s/bench_line.json"))
r=d["roofline"]; print("frac", r["frac"], "kernel_ms", r["kernel_ms"], "caller", r.get("frac_caller_planes"), "clock", r.get("clock_GHz"))
PY
