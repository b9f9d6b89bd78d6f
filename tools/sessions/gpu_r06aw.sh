#!/bin/bash
# round 6, session aw: the complete default bench line (driver-style: python bench.py --steps 20 --warmup 5) on the last commit -- rc, wall time, the headline and the placement it ran on
O=gpurun_out/r06aw; mkdir -p $O
export PYTHONUNBUFFERED=1
t0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06aw/bench_line.json")); r=d["roofline"]
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", r["frac"], "kernel_ms", r["kernel_ms"], "| planes", r["planes"], "| caller", r.get("frac_caller_planes"), "scattered", r.get("frac_scattered_planes"))
print("keys", sorted(d.keys())); print("secondary", sorted(d["secondary"].keys()))
nk=d["secondary"]["nuthkaab"]; print("nk", nk["ms_per_iteration"], nk["ms_per_iteration_settled"], nk["ms_per_iteration_whole_fit"]); v=d["secondary"]["variogram"]; print("vario", v["matheron_pass_Gpairs_s"], v["dowd_exact_median_Gpairs_s"])
PY
tail -3 $O/bench.err | cut -c1-200
