#!/bin/bash
# round 6, session m: rows per workgroup chunk of the one-pass Nuth-Kaab kernel (settled steps, one process per value)
O=gpurun_out/r06m; mkdir -p $O
export PYTHONUNBUFFERED=1
for c in 256 128 64 32 16; do
  XDEMHIP_NK_CHUNK_ROWS=$c NK_SETTLED=1 timeout 200 python -u tools/nk_trace.py 20000 6 > $O/steps_chunk$c.log 2>&1; echo "chunk rows $c:"; grep -E "step 20000" $O/steps_chunk$c.log | cut -c1-120 | tail -4
done
NK_SETTLED=1 timeout 200 python -u tools/nk_trace.py 20000 6 > $O/steps_default.log 2>&1; echo "default:"; grep -E "step 20000" $O/steps_default.log | cut -c1-120 | tail -4
