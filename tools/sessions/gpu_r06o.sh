#!/bin/bash
# round 6, session o: whole GPU suite + the default bench line on the library with the byte-wide bin cache and the dh-only prediction
O=gpurun_out/r06o; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 3300 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -4 $O/pytest_all.log | cut -c1-300
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06o/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('frac',r['frac'],'caller',r.get('frac_caller_planes'),'ms',r['kernel_ms'],r.get('kernel_ms_caller_planes'),'clock',r.get('clock_GHz'),r.get('clock_GHz_caller_planes'))
s=d['secondary']; n=s['nuthkaab']; print('nk',n['ms_per_iteration'],n['ms_per_iteration_whole_fit'],n.get('ms_per_iteration_settled'),n.get('settled_roofline_frac'),n['roofline']['frac'],n['routes'])
print('vario B',s['variogram']['dowd_exact_median_Gpairs_s'],s['variogram']['matheron_pass_Gpairs_s'],'A',s['variogram_c5a']['dowd_exact_median_Gpairs_s'],s['variogram_c5a']['matheron_pass_Gpairs_s'])
for k,v in s['terrain_sets']['sets'].items(): print(k[:40], v['kernel_ms_median'], v['frac_of_hbm_peak'], (v.get('issue') or {}).get('issue_frac'))
PY
