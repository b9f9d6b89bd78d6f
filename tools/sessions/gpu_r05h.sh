#!/bin/bash
# round-5 validation session: whole GPU suite, smoke, the default bench line, rocprofv3 passes of the bench command (stats, FETCH_SIZE, WRITE_SIZE,
# GRBM, SQ) -> profiles/r05_bench_terrain_*
TAG=${1:-r05h}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1700 python -X faulthandler -m pytest tests -q -m gpu --maxfail=8 > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300
grep -E "^E  " $O/pytest.log | head -12 | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python -u bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err | cut -c1-300
python - $O <<'P'
import json, sys
for l in open(sys.argv[1] + "/bench.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; s = d.get("secondary", {})
        print("headline", d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], "caller planes", r.get("kernel_ms_caller_planes"), r.get("frac_caller_planes"), "spot", r.get("output_spot_check"), "traffic", r.get("traffic_source", "")[:40])
        print("  cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("gpu_vs_oracle_on_this_sample"))
        for k, v in s.get("terrain_sets", {}).get("sets", {}).items():
            print("  set", k[:44], v["kernel_ms_median"], v["frac_of_hbm_peak"])
        for k in ("variogram", "variogram_c5a"):
            v = s.get(k, {}); print("  ", k, v.get("pairs"), v.get("matheron_pass_Gpairs_s"), v.get("dowd_exact_median_Gpairs_s"))
        n = s.get("nuthkaab", {}); print("   nk", n.get("ms_per_iteration"), n.get("ms_per_iteration_whole_fit"), n.get("routes"), n.get("roofline", {}).get("frac"), n.get("roofline", {}).get("frac_at_survey_bytes"), "e2e", d.get("end_to_end", {}).get("Mpixels_s"), s.get("error"))
P
bash tools/profile_bench.sh $TAG 40000 > $O/profile_bench.log 2>&1; tail -14 $O/profile_bench.log | cut -c1-220
