#!/bin/bash
# round 6: the headline terrain leg of bench.py (driver's steps / warm-up, no secondary legs) on whatever box this call lands on -- one row of
# profiles/r06_box_table.txt per call (the same binary over many boxes)
TAG=${1:-x}
O=gpurun_out/r06box; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-end-to-end > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"
python - <<PY
import json, hashlib
d=json.loads(open('gpurun_out/r06box/bench_$TAG.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$TAG', 'lib md5', hashlib.md5(open('xdem_amd/csrc/libxdemhip.so','rb').read()).hexdigest()[:8], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], '| caller planes', r.get('kernel_ms_caller_planes'), r.get('frac_caller_planes'), '| clock GHz', r.get('clock_GHz'), r.get('clock_GHz_caller_planes'))
PY
