#!/bin/bash
# round 6, session z2: whole GPU suite + bench line after the Nuth-Kaab changes of the second session (sixteenth brackets, one sample kernel, fixed-order sums)
O=gpurun_out/r06z2; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_all.log | cut -c1-200
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06z2/bench_line.json"))
r=d["roofline"]; print("frac", r["frac"], "kernel_ms", r["kernel_ms"], "caller", r.get("frac_caller_planes"), "clock", r.get("clock_GHz"), r.get("clock_GHz_caller_planes"))
nk=d["secondary"]["nuthkaab"]; print("nk", nk["ms_per_iteration"], nk["ms_per_iteration_settled"], nk["ms_per_iteration_whole_fit"], nk["routes"], nk["roofline"].get("data_pass"))
v=d["secondary"]["variogram"]; print("vario", v["matheron_pass_Gpairs_s"], v["dowd_exact_median_Gpairs_s"]); v=d["secondary"]["variogram_c5a"]; print("c5a", v["matheron_pass_Gpairs_s"], v["dowd_exact_median_Gpairs_s"])
PY
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06z2/bench_line.json"))
print("e2e calls", json.dumps(d.get("end_to_end_calls"))[:600]); print("e2e terrain", d["end_to_end"]["seconds"], d["end_to_end"]["effective_GBps_over_PCIe"])
PY
tail -2 gpurun_out/r06z2/bench.err | cut -c1-300
