#!/bin/bash
# round 6, session ao: the public convolution (8a row a5) and per-bin lookup (8f row f3) -- GPU parity tests, rates, kernel stats
O=gpurun_out/r06ao; mkdir -p $O
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_convolution_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest_conv.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_conv.log | cut -c1-220
timeout 600 python tools/probes/conv_probe.py 16384 > $O/conv_probe.txt 2>&1; cat $O/conv_probe.txt | cut -c1-200
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o conv -- python $R/tools/probes/conv_probe.py 8192 > $R/$O/prof.log 2>&1 )
for f in $(find $O/prof -name "*kernel_stats.csv"); do cp $f $O/r06ao_conv_kernel_stats.csv; head -8 $f | cut -c1-200; done
find $O -name '*.csv' -size +2M -delete; find $O -name "*.db" -delete
