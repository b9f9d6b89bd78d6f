#!/bin/bash
# round 6, session aj: every step of a fresh plan's fit timed on its own
O=gpurun_out/r06aj; mkdir -p $O
export PYTHONUNBUFFERED=1
XDEMHIP_DEBUG=1 timeout 300 python -u tools/probes/nk_fit_steps_probe.py 20000 > $O/fit_steps.log 2>&1; grep -E "trial|brackets x|one-pass step \(" $O/fit_steps.log | cut -c1-230
