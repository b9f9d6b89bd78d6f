#!/bin/bash
# round 6, session l: the rest of the GPU suite (session k stopped at one stale test reference)
O=gpurun_out/r06l; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 3300 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_all.log 2>&1; echo "suite rc=$?"; tail -12 $O/pytest_all.log | cut -c1-300
