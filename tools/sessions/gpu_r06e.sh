#!/bin/bash
# round 6, session e: piece sizes of the scattered backing and a scattered input raster; bench line with the in-kernel clock probe
O=gpurun_out/r06e; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python tools/piece_probe.py --pieces 2,8,32,128 > $O/piece_probe.txt 2>&1; echo "piece rc=$?"; grep -v "^/opt" $O/piece_probe.txt | tail -20
timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06e/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('frac',r['frac'],'caller',r.get('frac_caller_planes'),'ms',r['kernel_ms'],r.get('kernel_ms_caller_planes'),'clock',r.get('clock_GHz'),r.get('clock_GHz_caller_planes'),'power',r.get('power_W'))
print(json.dumps(r['gpu_state_during_timed_steps'].get('shader_clock_under_load')))
print(r['gpu_state_during_timed_steps'].get('sysfs_note'))
PY
