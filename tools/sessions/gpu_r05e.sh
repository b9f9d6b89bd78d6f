#!/bin/bash
# round-5: the float32 shadow of float64-difference pair sets (WIDE kernels): its test, the variogram suite, the phases under the conventions
TAG=${1:-r05e}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -X faulthandler -m pytest tests/test_variogram_gpu.py -q -m gpu --maxfail=6 > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300
grep -E "^E  " $O/pytest.log | head -20 | cut -c1-250
XDEMHIP_DEBUG=1 timeout 600 python -u tools/vario_conventions_probe.py 100 > $O/vario_conv.log 2> $O/vario_conv.err; echo "probe rc $?"
cat $O/vario_conv.log | cut -c1-200
grep -E "^----|sampled digit|counting|selection among|candidates \(|plain|attempt|missed" $O/vario_conv.err | cut -c1-160 | tail -24
