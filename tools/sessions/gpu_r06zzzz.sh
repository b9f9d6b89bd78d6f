#!/bin/bash
# round 6, last call: whole GPU suite + smoke + the default bench line on the binary of the round's last commit (per-bin kernel templated on n_var, whole-filter weight loads)
O=gpurun_out/r06zzzz; mkdir -p $O
export PYTHONUNBUFFERED=1
md5sum xdem_amd/csrc/libxdemhip.so | tee $O/md5.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_default.log 2>&1; echo "pytest default rc=$?" | tee -a $O/pytest_default.log; tail -3 $O/pytest_default.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-200
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06zzzz/bench_line.json"))
r=d["roofline"]; print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", r["frac"], "kernel_ms", r["kernel_ms"], "caller", r.get("frac_caller_planes"), "clock", r.get("clock_GHz"))
print("public", json.dumps(d["secondary"].get("public_functions"))[:700])
nk=d["secondary"]["nuthkaab"]; print("nk", nk["ms_per_iteration"], nk["ms_per_iteration_settled"], nk["ms_per_iteration_whole_fit"])
v=d["secondary"]["variogram"]; print("vario", v["matheron_pass_Gpairs_s"], v["dowd_exact_median_Gpairs_s"])
PY
