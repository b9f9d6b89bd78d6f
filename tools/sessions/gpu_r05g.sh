#!/bin/bash
TAG=${1:-r05g}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
XDEMHIP_DEBUG=1 timeout 300 python -u tools/nk_trace.py 20000 3 > $O/nk_default.log 2>&1; grep -E "step|routes|falls|xdemhip" $O/nk_default.log | tail -12 | cut -c1-220
