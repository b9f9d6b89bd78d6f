#!/bin/bash
# round 6, session ar: one headline line with the driver's arguments on whatever box the pool hands out (box table of the end-state harness)
O=gpurun_out/r06ar; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-end-to-end > $O/line.json 2>> $O/bench.err
python - <<'PY'
import json, socket, time
d=json.load(open("gpurun_out/r06ar/line.json")); r=d["roofline"]
s=r["kernel_ms_series"]
print(time.strftime("%H:%M:%S"), "| planes", (r.get("planes") or {}).get("backing"), (r.get("planes") or {}).get("calibration_ms"), "scattered", r.get("frac_scattered_planes"), "| kernel_ms", r["kernel_ms"], "frac", r["frac"], "caller", r.get("frac_caller_planes"), "clock", r.get("clock_GHz"), r.get("clock_GHz_caller_planes"), "| first", s[0], "min", min(s), "max", max(s))
PY
cat $O/line.json >> $O/lines.jsonl
