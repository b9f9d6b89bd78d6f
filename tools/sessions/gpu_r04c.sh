#!/bin/bash
# round-4 session 2: GPU suite (one-pass Nuth-Kaab step, LDS window kernel, small-set strips, RCCL exchange test), bench,
# strip-sync variants on alternating contiguous placements, window-kernel rates, dispatch sequence of the one-pass step
O=gpurun_out/r04c; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -X faulthandler -m pytest tests -q -m gpu --maxfail=6 > $O/pytest.log 2>&1
tail -12 $O/pytest.log | cut -c1-400
timeout 600 python -u bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err | cut -c1-300
python - <<'P'
import json
for l in open("gpurun_out/r04c/bench.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; s = d.get("secondary", {})
        print("headline", d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], "caller planes", r.get("kernel_ms_caller_planes"), r.get("frac_caller_planes"))
        for k, v in s.get("terrain_sets", {}).get("sets", {}).items():
            print("  set", k[:40], v["kernel_ms_median"], v["Mpixels_s"], v["frac_of_hbm_peak"])
        for k in ("variogram", "variogram_c5a"):
            v = s.get(k, {}); print("  ", k, v.get("pairs"), v.get("matheron_pass_Gpairs_s"), v.get("dowd_exact_median_Gpairs_s"))
        n = s.get("nuthkaab", {}); print("   nk", n.get("ms_per_iteration"), n.get("ms_per_iteration_whole_fit"), n.get("routes"), n.get("roofline", {}).get("frac"), "e2e", d.get("end_to_end", {}).get("Mpixels_s"), s.get("error"))
P
timeout 280 python -u tools/window_bench.py > $O/window_bench.log 2>&1; cat $O/window_bench.log
for lib in libxdemhip.so libxdemhip_expsy1.so libxdemhip_expsy2.so libxdemhip_expsy4.so libxdemhip.so; do
  for rep in 1 2; do
    XD_LIB=$GRAFT_REPO_ROOT/xdem_amd/csrc/$lib timeout 120 python -u tools/backing_run.py contiguous 6 2>&1 | grep "^lib" | cut -c1-200
  done
done | tee $O/sync_probe.log
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/nktrace -o nk -- python $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 2 > $GRAFT_REPO_ROOT/$O/nktrace.log 2>&1
cd $GRAFT_REPO_ROOT
grep "^step" $O/nktrace.log
python tools/trace_sequence.py $O/nktrace 70 > $O/nk_sequence.txt 2>&1; tail -72 $O/nk_sequence.txt
find $O/nktrace -name '*.csv' -size +2M -delete
