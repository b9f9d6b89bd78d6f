#!/bin/bash
# round 6, session ra: the two-pass Nuth-Kaab route retired (one-pass + plain): Nuth-Kaab / binning / partitioned GPU tests, and what the
# plain route costs per C3 step (option nk_fused = 0; the mean statistic; the same under a 1-rank RCCL hook)
O=gpurun_out/r06ra; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests/test_nuthkaab_gpu.py tests/test_binning_gpu.py tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_nk.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_nk.log | cut -c1-300
for cfg in "" "NK_FUSED=0" "NK_FUSED=0 NK_STAT=mean" "NK_STAT=mean" "NK_HOOKED=1" "NK_HOOKED=1 NK_FUSED_DIST=0"; do
  echo "== $cfg"; env $cfg timeout 300 python tools/nk_trace.py 20000 5 2>&1 | grep -E "step 20000|routes" | tail -4
done | tee $O/plain_route_cost.txt
