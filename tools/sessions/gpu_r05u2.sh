#!/bin/bash
# round-5, sessions u2 (run with tag r05u: 12 all-reduces per step) and v (10): the one-pass step on partitioned plans (10 all-reduces per step) -- the new tests first, the tests of the hooked
# plans that now take the route, the shared-GPU probe of one-pass vs two-pass on 2 ranks
TAG=${1:-r05u}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1 XDEMHIP_DEBUG=1
timeout 300 python -m pytest tests/test_nuthkaab_gpu.py -q -m gpu -p no:cacheprovider -x -s -k "hooked_plan or sharded_reduction_path" > $O/pytest_a.log 2>&1; echo "a rc=$?"; tail -25 $O/pytest_a.log | cut -c1-400
timeout 500 python -m pytest tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider -x -s -k "one_pass_step or partitioned_row_blocks" > $O/pytest_b.log 2>&1; echo "b rc=$?"; tail -25 $O/pytest_b.log | cut -c1-400
timeout 300 python -u tools/nk_dist_probe.py 20000 2 5 > $O/nk_dist_probe.log 2>&1; echo "probe rc=$?"; grep -v "^\[xdemhip\]" $O/nk_dist_probe.log | tail -8 | cut -c1-400
