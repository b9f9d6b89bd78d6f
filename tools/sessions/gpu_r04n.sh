#!/bin/bash
# round-4 session 14: spread sample with 1 / 2 / 4 B slots per lane and tile (A/B of builds), offsets of the brackets
O=gpurun_out/r04q; mkdir -p $O
export PYTHONUNBUFFERED=1
for v in "" _va1 _va4 ""; do
  XD_LIB=$GRAFT_REPO_ROOT/xdem_amd/csrc/libxdemhip$v.so XDEMHIP_DEBUG=1 PROBE_CFG=0 timeout 200 python tools/vario_runs_probe.py 9091 100 > $O/probe$v.txt 2> $O/probe$v.err
  echo "== libxdemhip$v"; grep "run-length" $O/probe$v.txt; grep -E "sampled digit|selection among|wanted rank|candidates \(" $O/probe$v.err | tail -4
done
