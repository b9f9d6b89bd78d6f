#!/bin/bash
# round-5, session ad: the class-interval subtract of the lattice pair loops folded into the v_dot2 accumulator -- variogram GPU tests, then
# A/B of the two builds on C5 reading B (Matheron pass, exact Dowd), one process each
TAG=${1:-r05ad}
O=gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 150 python -m pytest tests/test_variogram_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_vario.log 2>&1; echo "pytest rc=$?" >> $O/pytest_vario.log; tail -3 $O/pytest_vario.log | cut -c1-300
for lib in libxdemhip_base.so libxdemhip.so; do
  echo "== $lib"; XD_LIB=$PWD/xdem_amd/csrc/$lib timeout 60 python tools/vario_c5_probe.py 0 2>/dev/null | tail -2
done > $O/ab_vario.txt 2>&1
cat $O/ab_vario.txt
