#!/bin/bash
# SQ counters of the variogram pair kernels (<= 8 per pass, every rocprofv3 under timeout); run through gpurun
TAG=${1:-r02x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for LAT in 1 0; do
  export C5_LATTICE=$LAT C5_RUNS=25
  timeout 120 python $GRAFT_REPO_ROOT/tools/vario_c5b.py 2>&1 | tail -1
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq_lat$LAT -o v -- python $GRAFT_REPO_ROOT/tools/vario_c5b.py > $OUT/sq_lat$LAT.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $OUT/sq2_lat$LAT -o v -- python $GRAFT_REPO_ROOT/tools/vario_c5b.py > $OUT/sq2_lat$LAT.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT "pairs_kernel<float, 0" 
python tools/pmc_summary.py $OUT "pairs_kernel<float, 4" | head -60
