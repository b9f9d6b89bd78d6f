#!/bin/bash
# round 6, session ak: per-kernel times of one exact-Dowd call on C5 reading B (where the +35 % over the Matheron pass go)
O=gpurun_out/r06ak; mkdir -p $O
export PYTHONUNBUFFERED=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
XDEMHIP_DEBUG=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o t -- python -u $R/tools/vario_c5_probe.py 0 > $R/$O/probe.log 2>&1; echo "rc=$?"
cd $R
grep -E "pairs|Dowd|xdemhip" $O/probe.log | cut -c1-220 | tail -30
python tools/trace_sequence.py $O/trace 40 > $O/sequence.txt 2>&1; tail -44 $O/sequence.txt | cut -c1-170
find $O -name '*.csv' -size +3M -delete
