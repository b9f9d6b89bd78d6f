#!/bin/bash
# round 6, session as: library with the unclipped-hillshade mode of the engine boundary -- whole GPU suite, headline line (driver's arguments)
O=gpurun_out/r06as; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_default.log 2>&1; echo "pytest default rc=$?" | tee -a $O/pytest_default.log; tail -3 $O/pytest_default.log | cut -c1-200
grep -E "FAILED|Error" $O/pytest_default.log | head -20
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-end-to-end > $O/line.json 2>> $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06as/line.json")); r=d["roofline"]; s=r["kernel_ms_series"]
print("kernel_ms", r["kernel_ms"], "frac", r["frac"], "caller", r.get("frac_caller_planes"), "clock", r.get("clock_GHz"), "| first", s[0], "min", min(s), "max", max(s))
PY
