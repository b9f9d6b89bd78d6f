#!/bin/bash
# round-4 closing session: whole GPU suite, bench, rocprofv3 passes of the bench command, dispatch sequence of the one-pass step
TAG=${1:-r04}
O=gpurun_out/${TAG}z; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --maxfail=8 > $O/pytest.log 2>&1
tail -6 $O/pytest.log | cut -c1-300
timeout 700 python -u bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err | cut -c1-300
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
bash tools/profile_bench.sh $TAG 40000 > $O/profile_bench.log 2>&1; tail -12 $O/profile_bench.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/nktrace -o nk -- python $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 3 > $GRAFT_REPO_ROOT/$O/nktrace.log 2>&1
cd $GRAFT_REPO_ROOT
grep -E "step 2|routes" $O/nktrace.log
python tools/trace_sequence.py $O/nktrace 56 > $O/nk_sequence.txt 2>&1; tail -3 $O/nk_sequence.txt
find $O -name '*.csv' -size +2M -delete
