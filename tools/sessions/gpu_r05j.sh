#!/bin/bash
# round-5: A/B of the per-class record in the run-length counting pass; the tests the last changes touch (variogram, multi-rank bench flow);
# the default bench line; rocprofv3 passes of the bench command -> profiles/r05_bench_terrain_*
TAG=${1:-r05j}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2; do
  for lib in libxdemhip.so libxdemhip_va0.so; do
    XD_LIB=$GRAFT_REPO_ROOT/xdem_amd/csrc/$lib PROBE_CFG=0,0 timeout 300 python -u tools/vario_runs_probe.py > $O/probe_${lib}_$rep.log 2>&1
    echo "$lib rep $rep:"; grep -E "run-length" $O/probe_${lib}_$rep.log | cut -c1-150
  done
done
timeout 1200 python -X faulthandler -m pytest tests/test_variogram_gpu.py tests/test_dist_gpu.py -q -m gpu --maxfail=6 > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-200
grep -E "^E  " $O/pytest.log | head -8 | cut -c1-250
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
bash tools/profile_bench.sh $TAG 40000 > $O/profile_bench.log 2>&1; tail -6 $O/profile_bench.log | cut -c1-220
