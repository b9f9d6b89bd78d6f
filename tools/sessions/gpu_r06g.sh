#!/bin/bash
# round 6, session g: predicted brackets with the extrapolated rule -- NK suite + the partitioned tests, the settled sequence under rocprofv3,
# the whole bench line
O=gpurun_out/r06g; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_nk.log 2>&1; echo "nk suite rc=$?"; tail -4 $O/pytest_nk.log | cut -c1-300
timeout 1200 python -m pytest tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider -x -k "nuth or nk" > $O/pytest_dist_nk.log 2>&1; echo "dist nk rc=$?"; tail -6 $O/pytest_dist_nk.log | cut -c1-300
NK_SETTLED=1 XDEMHIP_DEBUG=1 timeout 200 python -u tools/nk_trace.py 20000 6 > $O/steps_settled_pred.log 2>&1; grep -E "step 20000|routes|falls" $O/steps_settled_pred.log | cut -c1-220
cd /tmp && export TMPDIR=/tmp
NK_SETTLED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_pred -o pred -- python -u $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 3 > $GRAFT_REPO_ROOT/$O/trace_pred.log 2>&1; echo "trace rc=$?"
NK_SETTLED=1 NK_HOOKED=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_hooked -o hooked -- python -u $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 3 > $GRAFT_REPO_ROOT/$O/trace_hooked.log 2>&1; echo "trace hooked rc=$?"; grep -E "step 20000|routes" $GRAFT_REPO_ROOT/$O/trace_hooked.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python tools/trace_sequence.py $O/trace_pred 14 > $O/sequence_pred.txt 2>&1; tail -18 $O/sequence_pred.txt | cut -c1-150
python tools/trace_sequence.py $O/trace_hooked 40 > $O/sequence_hooked.txt 2>&1; tail -44 $O/sequence_hooked.txt | cut -c1-150
find $O -name '*.csv' -size +3M -delete
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06g/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('frac',r['frac'],'caller',r.get('frac_caller_planes'),'ms',r['kernel_ms'],r.get('kernel_ms_caller_planes'),'clock',r.get('clock_GHz'),r.get('clock_GHz_caller_planes'))
s=d['secondary']; n=s['nuthkaab']; print('nk',n['ms_per_iteration'],n['ms_per_iteration_whole_fit'],n.get('ms_per_iteration_settled'),n.get('settled_roofline_frac'),n['routes'])
print('vario B',s['variogram']['dowd_exact_median_Gpairs_s'],s['variogram']['matheron_pass_Gpairs_s'],'A',s['variogram_c5a']['dowd_exact_median_Gpairs_s'])
for k,v in s['terrain_sets']['sets'].items(): print(k, v['kernel_ms_median'], v['frac_of_hbm_peak'])
PY
