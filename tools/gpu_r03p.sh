#!/bin/bash
# round 3 final evidence: bench (cold box first), rocprofv3 stats + PMC passes of the bench command, NK / variogram kernel
# stats, whole GPU suite, smoke
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03p}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_first.log 2> $OUT/bench_first.err
bash tools/profile_bench.sh r03 40000 > $OUT/profile.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_last.log 2> $OUT/bench_last.err
timeout 300 python tools/ulp_report.py --gpu --size 8000 --tail 2 > $OUT/ulp_lean.txt 2>&1
python - <<PY
import json
for f in ("bench_first","bench_last"):
    d=json.loads([l for l in open("$OUT/%s.log"%f) if l.startswith("{")][0])
    s=d.get("secondary",{})
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], "| vario", s.get("variogram",{}).get("matheron_pass_Gpairs_s"), s.get("variogram",{}).get("dowd_exact_median_Gpairs_s"), "| nk", s.get("nuthkaab",{}).get("ms_per_iteration"), s.get("error"), "| e2e", d.get("end_to_end",{}).get("Mpixels_s"))
PY
