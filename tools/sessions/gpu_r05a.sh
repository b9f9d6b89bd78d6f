#!/bin/bash
# round-5 first session: whole GPU suite (a8 fixtures, +-Inf rule, mp_config, decision-file contexts), smoke, the bench line and the
# rocprofv3 kernel trace OF THE SAME PROCESS (tools/bench_same_process.py), then the default bench line
TAG=${1:-r05a}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --maxfail=8 > $O/pytest.log 2>&1
tail -6 $O/pytest.log | cut -c1-300
grep -E "^T11|^T3 " $O/pytest.log | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
R=$GRAFT_REPO_ROOT
mkdir -p $O/same
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/same -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-end-to-end > $R/$O/same/bench.log 2> $R/$O/same/bench.err )
python tools/bench_same_process.py $O/same $O/${TAG}_bench_same_process.json > $O/same_summary.txt 2>&1; cat $O/same_summary.txt | cut -c1-200
for f in $(find $O/same -name "*kernel_stats.csv"); do cp $f $O/${TAG}_same_process_kernel_stats.csv; done
find $O -name '*.csv' -size +2M -delete
timeout 900 python -u bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err | cut -c1-300
python - $O <<'P'
import json, sys
for l in open(sys.argv[1] + "/bench.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; s = d.get("secondary", {})
        print("headline", d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], "caller planes", r.get("kernel_ms_caller_planes"), r.get("frac_caller_planes"))
        for k, v in s.get("terrain_sets", {}).get("sets", {}).items():
            print("  set", k[:40], v["kernel_ms_median"], v["Mpixels_s"], v["frac_of_hbm_peak"])
        for k in ("variogram", "variogram_c5a"):
            v = s.get(k, {}); print("  ", k, v.get("pairs"), v.get("matheron_pass_Gpairs_s"), v.get("dowd_exact_median_Gpairs_s"))
        n = s.get("nuthkaab", {}); print("   nk", n.get("ms_per_iteration"), n.get("ms_per_iteration_whole_fit"), n.get("routes"), n.get("roofline", {}).get("frac"), "e2e", d.get("end_to_end", {}).get("Mpixels_s"), s.get("error"))
P
