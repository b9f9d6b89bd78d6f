#!/bin/bash
# round-5: SQ counters of the small attribute sets' streaming kernels (why do 8-12 B/pixel launches sit at 3.6-3.9 ms whatever the fit?)
TAG=${1:-r05d}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/small_sets_pmc.py 40000"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/p1 -o s -- $CMD > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/p2 -o s -- $CMD > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -o s -- $CMD > $O/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL --output-format csv -d $O/p4 -o s -- $CMD > $O/p4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $O/p5 -o s -- $CMD > $O/p5.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_by_kernel.py $O terrain_strip > $O/summary.txt 2>&1
find $O -name '*.csv' -size +2M -delete
cat $O/summary.txt | cut -c1-400 | head -80
tail -2 $O/p1.log
