#!/bin/bash
# SQ counters + HBM bytes of the Nuth-Kaab lean kernels and of the variogram passes (<= 8 SQ counters per pass, every rocprofv3
# under timeout); run through gpurun:  bash tools/sessions/gpu_nk_pmc.sh <tag>
TAG=${1:-r02nk}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
NK="python $GRAFT_REPO_ROOT/tools/nk_probe.py 20000 3"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/nk_sq -o v -- $NK > $OUT/nk_sq.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/nk_fetch -o v -- $NK > $OUT/nk_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/nk_write -o v -- $NK > $OUT/nk_write.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/nk_grbm -o v -- $NK > $OUT/nk_grbm.log 2>&1
export C5_LATTICE=1 C5_RUNS=25
VA="python $GRAFT_REPO_ROOT/tools/vario_c5b.py"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/va_sq -o v -- $VA > $OUT/va_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $OUT/va_sq2 -o v -- $VA > $OUT/va_sq2.log 2>&1
cd $GRAFT_REPO_ROOT
for k in "nk_dh_count_lean_kernel" "nk_bins_lean_kernel" "pairs_kernel<float, 0" "pairs_kernel<float, 4"; do echo "== $k"; python tools/pmc_summary.py $OUT "$k" | head -60; done
