#!/bin/bash
# round-5 second session: whole GPU suite (rules 2 / 3 through the streaming Nuth-Kaab kernels, directional specialisations, mp_config),
# the default bench line, then the SECONDARY legs once more under the non-default third-party conventions (nk_nan_rule = 3,
# vario_edge = 1, vario_diff = 1) through a decision file -- the conventions a later thirdparty_decision.json could land on
TAG=${1:-r05b}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1700 python -X faulthandler -m pytest tests -q -m gpu --maxfail=8 > $O/pytest.log 2>&1
tail -6 $O/pytest.log | cut -c1-300
timeout 900 python -u bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err | cut -c1-300
echo '{"nk_nan_rule": 3, "vario_edge": 1, "vario_diff": 1, "_source": "tools/sessions/gpu_r05b.sh: the non-default conventions"}' > /tmp/alt_decision.json
XDEM_THIRDPARTY_DECISION=/tmp/alt_decision.json timeout 900 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/bench_alt.log 2> $O/bench_alt.err; echo "bench (alternative conventions) rc $?"; tail -2 $O/bench_alt.err | cut -c1-300
python - $O <<'P'
import json, sys
for f in ("bench.log", "bench_alt.log"):
    for l in open(sys.argv[1] + "/" + f):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]; s = d.get("secondary", {})
            print(f, "headline", d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], "caller planes", r.get("kernel_ms_caller_planes"), r.get("frac_caller_planes"))
            for k, v in s.get("terrain_sets", {}).get("sets", {}).items():
                print("  set", k[:44], v["kernel_ms_median"], v["Mpixels_s"], v["frac_of_hbm_peak"])
            for k in ("variogram", "variogram_c5a"):
                v = s.get(k, {}); print("  ", k, v.get("pairs"), v.get("matheron_pass_Gpairs_s"), v.get("dowd_exact_median_Gpairs_s"), v.get("conventions"))
            n = s.get("nuthkaab", {}); print("   nk", n.get("ms_per_iteration"), n.get("ms_per_iteration_whole_fit"), n.get("routes"), "rule", n.get("nk_nan_rule"), n.get("roofline", {}).get("frac"), n.get("roofline", {}).get("frac_at_survey_bytes"), "e2e", d.get("end_to_end", {}).get("Mpixels_s"), s.get("error"))
P
