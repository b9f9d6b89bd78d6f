#!/bin/bash
# round-4 session 11: parity of the selection changes; SQ counters of the run-length counting pass (C5 sampler geometry, 25 runs)
O=gpurun_out/r04n; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_variogram_gpu.py tests/test_binning_gpu.py tests/test_nuthkaab_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in 0 1; do
  export PROBE_CFG=$cfg
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/$O/sq1_cfg$cfg -o v -- python $R/tools/vario_runs_probe.py 9091 25 > $R/$O/sq1_cfg$cfg.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $R/$O/sq2_cfg$cfg -o v -- python $R/tools/vario_runs_probe.py 9091 25 > $R/$O/sq2_cfg$cfg.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $R/$O/sq3_cfg$cfg -o v -- python $R/tools/vario_runs_probe.py 9091 25 > $R/$O/sq3_cfg$cfg.log 2>&1
  tail -2 $R/$O/sq3_cfg$cfg.log
done
cd $R
python tools/pmc_summary.py $O "pairs_kernel<float, 4"
find $O -name '*.csv' -size +2M -delete
