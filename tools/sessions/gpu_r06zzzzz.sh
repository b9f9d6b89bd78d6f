#!/bin/bash
# round 6, session zzzzz: FINAL STATE of the round, with the calibrated plane placement (terrain.alloc_planes(probe=...)) in the bench line -- whole GPU suite under the default conventions and under the non-default decision file, smoke, the default bench
# line, the bench line + rocprofv3 kernel trace of ONE process, the closing PMC set of the bench command, the Nuth-Kaab dispatch sequences (settled / sampled)
TAG=r06zzzzz
O=gpurun_out/$TAG; mkdir -p $O/same
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_default.log 2>&1; echo "pytest default rc=$?" | tee -a $O/pytest_default.log; tail -3 $O/pytest_default.log | cut -c1-200
cat > /tmp/alt_decision.json <<'JSON'
{"nk_nan_rule": 3, "vario_edge": 1, "vario_diff": 1}
JSON
XDEM_THIRDPARTY_DECISION=/tmp/alt_decision.json timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_alt.log 2>&1; echo "pytest alt rc=$?" | tee -a $O/pytest_alt.log; tail -3 $O/pytest_alt.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-200
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06zzzzz/bench_line.json"))
r=d["roofline"]; print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", r["frac"], "kernel_ms", r["kernel_ms"], "caller", r.get("frac_caller_planes"), "clock", r.get("clock_GHz"), r.get("clock_GHz_caller_planes"))
nk=d["secondary"]["nuthkaab"]; print("nk", nk["ms_per_iteration"], nk["ms_per_iteration_settled"], nk["ms_per_iteration_whole_fit"], nk["routes"])
v=d["secondary"]["variogram"]; print("vario", v["matheron_pass_Gpairs_s"], v["dowd_exact_median_Gpairs_s"]); v=d["secondary"]["variogram_c5a"]; print("c5a", v["matheron_pass_Gpairs_s"], v["dowd_exact_median_Gpairs_s"])
print("public", json.dumps(d["secondary"].get("public_functions"))[:600]); print("series", r.get("kernel_ms_series")); print("planes", r.get("planes"), "caller", r.get("frac_caller_planes"), "scattered", r.get("frac_scattered_planes")); print("e2e calls", json.dumps(d.get("end_to_end_calls"))[:1200]); print("e2e terrain", d["end_to_end"]["seconds"], d["end_to_end"]["effective_GBps_over_PCIe"])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
PY
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/same -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-end-to-end > $R/$O/same/bench.log 2> $R/$O/same/bench.err )
python tools/bench_same_process.py $O/same $O/${TAG}_bench_same_process.json > $O/same_summary.txt 2>&1; cat $O/same_summary.txt | cut -c1-200
for f in $(find $O/same -name "*kernel_stats.csv"); do cp $f $O/${TAG}_same_process_kernel_stats.csv; done
# Nuth-Kaab dispatch sequences of the end state
cd /tmp && export TMPDIR=/tmp
NK_SETTLED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_settled -o t -- python -u $R/tools/nk_trace.py 20000 3 > $R/$O/trace_settled.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_sampled -o t -- python -u $R/tools/nk_trace.py 20000 3 > $R/$O/trace_sampled.log 2>&1
cd $R
python tools/trace_sequence.py $O/trace_settled 14 > $O/sequence_settled.txt 2>&1; tail -16 $O/sequence_settled.txt | cut -c1-150
python tools/trace_sequence.py $O/trace_sampled 21 > $O/sequence_sampled.txt 2>&1; tail -23 $O/sequence_sampled.txt | cut -c1-150
grep -E "step 20000|routes" $O/trace_settled.log $O/trace_sampled.log | cut -c1-200
find $O -name '*.csv' -size +2M -delete; find $O -name "*.db" -delete
# the closing rocprofv3 set of the round on the end state (kernel stats + PMC passes of the bench command)
timeout 2400 bash tools/profile_bench.sh r06zzzzz 40000 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -12 $O/profile.log | cut -c1-200
