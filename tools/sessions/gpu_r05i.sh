#!/bin/bash
# round-5: A/B of the per-class record in the run-length counting pass (libxdemhip.so against libxdemhip_va0.so = -DXD_NO_CLSREC), then the variogram tests
TAG=${1:-r05i}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2; do
  for lib in libxdemhip.so libxdemhip_va0.so; do
    XD_LIB=$GRAFT_REPO_ROOT/xdem_amd/csrc/$lib PROBE_CFG=0,0 timeout 300 python -u tools/vario_runs_probe.py > $O/probe_${lib}_$rep.log 2>&1
    echo "$lib rep $rep:"; grep -E "run-length|Matheron" $O/probe_${lib}_$rep.log | cut -c1-150
  done
done
timeout 900 python -X faulthandler -m pytest tests/test_variogram_gpu.py -q -m gpu --maxfail=6 > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-200
