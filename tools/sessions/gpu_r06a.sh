#!/bin/bash
# round 6, session a: what the box's sysfs offers for clock / power sampling; baseline bench line of the round-5 library with the sampler
O=gpurun_out/r06a; mkdir -p $O
export PYTHONUNBUFFERED=1
{
for d in /sys/class/drm/card*/device; do
  echo "== $d"; cat $d/vendor 2>&1; ls $d | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk pp_dpm_socclk gpu_busy_percent mem_busy_percent current_compute_partition current_memory_partition; do echo "-- $f"; cat $d/$f 2>&1 | head -12; done
  for h in $d/hwmon/hwmon*; do echo "== $h"; ls $h | tr '\n' ' '; echo; for f in $h/*_input $h/*_average $h/*_cap $h/*_label; do echo "$f: $(cat $f 2>&1)"; done; done
done
which rocm-smi amd-smi rocprofv3
} > $O/sysfs.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06a/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('frac',r['frac'],'caller',r.get('frac_caller_planes'),'ms',r['kernel_ms'],r.get('kernel_ms_caller_planes'))
print(json.dumps(r.get('gpu_state_during_timed_steps'))[:1500])
print(json.dumps(r['caller_planes'].get('gpu_state_during_timed_steps'))[:800])
s=d['secondary']; print('nk',s['nuthkaab']['ms_per_iteration'],'vario B',s['variogram']['dowd_exact_median_Gpairs_s'],s['variogram']['matheron_pass_Gpairs_s'],'A',s['variogram_c5a']['dowd_exact_median_Gpairs_s'])
for k,v in s['terrain_sets']['sets'].items(): print(k, v['kernel_ms_median'], v['frac_of_hbm_peak'])
PY
head -60 $O/sysfs.txt
