#!/bin/bash
# round-4 session: SQ counters of the counting pass in its final form (after the partial-A-tile / 32-pair-flush / design-effect changes)
O=gpurun_out/r04aa; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PROBE_CFG=0
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/$O/sq1_cfg0 -o v -- python $R/tools/vario_runs_probe.py 9091 25 > $R/$O/sq1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $R/$O/sq2_cfg0 -o v -- python $R/tools/vario_runs_probe.py 9091 25 > $R/$O/sq2.log 2>&1
cd $R
python tools/pmc_summary.py $O "pairs_kernel<float, 4"
find $O -name '*.csv' -size +2M -delete
