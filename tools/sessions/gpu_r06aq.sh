#!/bin/bash
# round 6, session aq: the timed steps one by one (kernel_ms_series) under the driver's arguments and under the defaults
O=gpurun_out/r06aq; mkdir -p $O
export PYTHONUNBUFFERED=1
for a in "--steps 20 --warmup 5" "--steps 10 --warmup 3" "--steps 20 --warmup 5"; do
  timeout 600 python bench.py $a --no-secondary --no-cpu-baseline --no-end-to-end > $O/line.json 2>> $O/bench.err
  python - "$a" <<'PY'
import json, sys
d=json.load(open("gpurun_out/r06aq/line.json")); r=d["roofline"]
print(sys.argv[1], "| frac", r["frac"], "caller", r.get("frac_caller_planes"), "clock", r.get("clock_GHz"), "| series", r["kernel_ms_series"])
PY
  cat $O/line.json >> $O/lines.jsonl
done
