#!/bin/bash
# round-4 session 1: GPU suite, bench (A/B of the plane backings, terrain sets, reading A), order probe, TCC / EA counters of
# contiguous against scattered planes, window-kernel baseline, dispatch sequence of a Nuth-Kaab step
O=gpurun_out/r04a; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -X faulthandler -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -4 $O/pytest.log | cut -c1-300
timeout 280 python -u tools/terrain_order_probe.py > $O/order_probe.log 2>&1; cat $O/order_probe.log | tail -6
timeout 600 python -u bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err | cut -c1-300
python - <<'P'
import json
for l in open("gpurun_out/r04a/bench.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; s = d.get("secondary", {})
        print("headline", d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], "caller planes", r.get("kernel_ms_caller_planes"), r.get("frac_caller_planes"))
        for k, v in s.get("terrain_sets", {}).get("sets", {}).items():
            print("  set", k, v["kernel_ms_median"], v["Mpixels_s"], v["frac_of_hbm_peak"])
        for k in ("variogram", "variogram_c5a"):
            v = s.get(k, {}); print("  ", k, v.get("pairs"), v.get("matheron_pass_Gpairs_s"), v.get("dowd_exact_median_Gpairs_s"))
        n = s.get("nuthkaab", {}); print("   nk", n.get("ms_per_iteration"), n.get("ms_per_iteration_whole_fit"), "e2e", d.get("end_to_end", {}).get("Mpixels_s"), s.get("error"))
P
timeout 280 python -u tools/window_bench.py > $O/window_bench.log 2>&1; cat $O/window_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/nktrace -o nk -- python $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 2 > $GRAFT_REPO_ROOT/$O/nktrace.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 $O/nktrace.log
python tools/trace_sequence.py $O/nktrace 75 > $O/nk_sequence.txt 2>&1; tail -80 $O/nk_sequence.txt
find $O/nktrace -name '*.csv' -size +2M -delete
timeout 1500 python -u tools/backing_pmc.py $O/backing > $O/backing_pmc.log 2>&1; tail -60 $O/backing_pmc.log
