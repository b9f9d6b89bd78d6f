#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03m}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nuthkaab_gpu.py tests/test_variogram_gpu.py -x -q -s > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log; grep "step plain" $OUT/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2> $OUT/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$OUT/bench.log") if l.startswith("{")][0])
print("terrain", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
s=d.get("secondary",{})
print("vario", {k:s.get("variogram",{}).get(k) for k in ("pairs","matheron_pass_Gpairs_s","dowd_exact_median_Gpairs_s")}, s.get("error"))
print("nk", {k:s.get("nuthkaab",{}).get(k) for k in ("ms_per_iteration","ms_per_iteration_whole_fit","fitted_shift_px")})
print("e2e", d.get("end_to_end"))
PY
tail -3 $OUT/bench.err
