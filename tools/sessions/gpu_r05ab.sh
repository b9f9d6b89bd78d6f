#!/bin/bash
# round-5, session ab: the round's END STATE -- A/B of the hillshade-only launch (one reciprocal square root), whole GPU suite, bench.py
TAG=${1:-r05ab}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
for m in 4 1; do
  echo "== mask $m (Florinsky)"; timeout 120 python tools/ab_libs.py --mask $m --fit 2 --reps 7 --rounds 3 base=xdem_amd/csrc/libxdemhip_base.so new=xdem_amd/csrc/libxdemhip.so 2>&1 | grep -v amdgpu.ids | tail -3
done > $O/ab_small_sets.txt 2>&1
cat $O/ab_small_sets.txt
timeout 330 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_all.log 2>&1
echo "pytest rc=$?" >> $O/pytest_all.log
tail -4 $O/pytest_all.log | cut -c1-300
timeout 200 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; python - $O <<'PY'
import json, sys
for l in open(sys.argv[1] + "/bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        r = d["roofline"]; nk = d["secondary"]["nuthkaab"]; v = d["secondary"]["variogram"]; va = d["secondary"]["variogram_c5a"]
        print("headline", d["ms_per_step"], r["kernel_ms"], r["frac"], "| caller planes", r.get("kernel_ms_caller_planes"), r.get("frac_caller_planes"))
        print("nk", nk["ms_per_iteration"], nk["ms_per_iteration_whole_fit"], nk["routes"], nk["roofline"]["frac"], nk["roofline"].get("frac_at_survey_bytes"))
        print("vario", v["matheron_pass_Gpairs_s"], v["dowd_exact_median_Gpairs_s"], va["matheron_pass_Gpairs_s"], va["dowd_exact_median_Gpairs_s"])
        for name, row in d["secondary"]["terrain_sets"]["sets"].items():
            print(f"  set {name:66.66s} {row['kernel_ms_median']:8.3f} ms  {row['frac_of_hbm_peak']}")
PY
