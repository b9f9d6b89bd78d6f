#!/bin/bash
# round-3 closing session: whole GPU suite, smoke, bench (first / last), Dowd probe
O=gpurun_out/r03final; mkdir -p $O
timeout 300 python -u bench.py > $O/bench_first.log 2> $O/bench_first.err
timeout 2400 python -X faulthandler -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-200
timeout 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
XDEMHIP_DEBUG=1 timeout 300 python -u tools/vario_c5_probe.py 0 > $O/dowd_probe.log 2> $O/dowd_probe.err
timeout 300 python -u bench.py > $O/bench_last.log 2> $O/bench_last.err
python - <<'P'
import json
for f in ("bench_first","bench_last"):
    for l in open("gpurun_out/r03final/%s.log"%f):
        if l.startswith("{"):
            d=json.loads(l); r=d["roofline"]; s=d.get("secondary",{})
            print(f, d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_max"], r["frac"])
            v=s.get("variogram",{}); n=s.get("nuthkaab",{})
            print("   vario", v.get("matheron_pass_Gpairs_s"), v.get("dowd_exact_median_Gpairs_s"), v.get("dowd_first_call_Gpairs_s"), "nk", n.get("ms_per_iteration"), n.get("ms_per_iteration_whole_fit"), "e2e", d.get("end_to_end",{}).get("Mpixels_s"), s.get("error"))
P
# closing kernel summary of the final build (one rocprofv3 --kernel-trace --stats pass of bench.py)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03final/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > $GRAFT_REPO_ROOT/gpurun_out/r03final/stats.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r03final/stats -name '*.csv' -size +3M -delete
head -9 $(find gpurun_out/r03final/stats -name '*kernel_stats.csv' | head -1) | cut -c1-110,250-330
