#!/bin/bash
# another box, whole GPU suite twice in a row (flakiness check after the address-reuse finding)
O=gpurun_out/r04ab; mkdir -p $O
export PYTHONUNBUFFERED=1
for i in 1 2; do
  timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --maxfail=8 -p no:cacheprovider > $O/pytest$i.log 2>&1
  tail -2 $O/pytest$i.log | cut -c1-300
done
