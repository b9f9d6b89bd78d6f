#!/bin/bash
# round 6, session rb: after the split of nuthkaab.hip (nk_geom.h, nk_onepass.h) and the retired two-pass route: the whole GPU suite
O=gpurun_out/r06rb; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 3000 python -m pytest tests/ -q -m gpu -p no:cacheprovider -s > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|routes:" $O/pytest_all.log | cut -c1-300 | tail -20
