#!/bin/bash
# round-5 third session: the tests that failed in r05b (rules 2 / 3 at a size the one-pass step answers; contexts closing their plans),
# phases of the exact Dowd route under the non-default variogram conventions
TAG=${1:-r05c}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -X faulthandler -m pytest tests/test_nuthkaab_gpu.py tests/test_terrain_gpu.py tests/test_concurrency_gpu.py -q -m gpu --maxfail=8 -k "dilating or bench_C3 or fresh_scattered or lean_kernels or decision or nan_rules" > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300
XDEMHIP_DEBUG=1 timeout 600 python -u tools/vario_conventions_probe.py 100 > $O/vario_conv.log 2> $O/vario_conv.err; echo "probe rc $?"
cat $O/vario_conv.log | cut -c1-200
grep -E "^----|sampled digit|counting|selection among|candidates \(|plain|attempt|missed" $O/vario_conv.err | cut -c1-160 | head -80
