#!/bin/bash
# round-5, session t: after the tie handling of the dh selection (key buckets, single-key buckets) -- the two new tests, the stress, the whole
# GPU suite, bench.py (the line of the round's end state)
TAG=${1:-r05t}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -k "whole_fit or C3_pair" > $O/pytest_new.log 2>&1; tail -4 $O/pytest_new.log | cut -c1-300
timeout 600 python -u tools/nk_stress.py 12000 300 > $O/nk_stress.log 2>&1; echo "rc=$?" >> $O/nk_stress.log; tail -4 $O/nk_stress.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; python - <<PY
import json
for l in open("$O/bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        r = d["roofline"]; nk = d["secondary"]["nuthkaab"]; v = d["secondary"]["variogram"]; va = d["secondary"]["variogram_c5a"]
        print("headline", d["ms_per_step"], r["kernel_ms"], r["frac"], "| caller planes", r.get("kernel_ms_caller_planes"), r.get("frac_caller_planes"))
        print("nk", nk["ms_per_iteration"], nk["ms_per_iteration_whole_fit"], nk["routes"], nk["roofline"]["frac"], nk["roofline"].get("frac_at_survey_bytes"))
        print("vario", v["matheron_pass_Gpairs_s"], v["dowd_exact_median_Gpairs_s"], va["matheron_pass_Gpairs_s"], va["dowd_exact_median_Gpairs_s"])
        for row in d["secondary"].get("terrain_sets", []):
            print("  set", row)
PY
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
tail -4 $O/pytest_all.log | cut -c1-300
