#!/bin/bash
# round 6, session am: PMC passes of the Nuth-Kaab data pass on the end-state library (per-workgroup sum slots instead of float64 atomics)
O=gpurun_out/r06am; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 bash tools/profile_nk.sh r06zz 20000 > $O/profile_nk.log 2>&1; echo "profile rc=$?"; tail -25 $O/profile_nk.log | cut -c1-220
