#!/bin/bash
# round-4 session 5: whole GPU suite, bench, dispatch sequence of the one-pass step
O=gpurun_out/r04e; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --maxfail=8 > $O/pytest.log 2>&1
tail -14 $O/pytest.log | cut -c1-300
timeout 600 python -u bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err | cut -c1-300
python - <<'P'
import json
for l in open("gpurun_out/r04e/bench.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; s = d.get("secondary", {})
        print("headline", d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], "caller planes", r.get("kernel_ms_caller_planes"), r.get("frac_caller_planes"))
        for k, v in s.get("terrain_sets", {}).get("sets", {}).items():
            print("  set", k[:40], v["kernel_ms_median"], v["Mpixels_s"], v["frac_of_hbm_peak"])
        for k in ("variogram", "variogram_c5a"):
            v = s.get(k, {}); print("  ", k, v.get("pairs"), v.get("matheron_pass_Gpairs_s"), v.get("dowd_exact_median_Gpairs_s"))
        n = s.get("nuthkaab", {}); print("   nk", n.get("ms_per_iteration"), n.get("ms_per_iteration_whole_fit"), n.get("routes"), n.get("roofline", {}).get("frac"), "e2e", d.get("end_to_end", {}).get("Mpixels_s"), s.get("error"))
P
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/nktrace -o nk -- python $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 2 > $GRAFT_REPO_ROOT/$O/nktrace.log 2>&1
cd $GRAFT_REPO_ROOT
grep "^step" $O/nktrace.log
python tools/trace_sequence.py $O/nktrace 64 > $O/nk_sequence.txt 2>&1; grep -v "select_advance\|bracket_keys\|select_reset\|rebase_shift" $O/nk_sequence.txt | tail -40
find $O -name '*.csv' -size +2M -delete
