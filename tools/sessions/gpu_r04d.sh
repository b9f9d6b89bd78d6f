#!/bin/bash
# round-4 session 4: Nuth-Kaab one-pass step after the mask rewrite (tests of the three routes, dispatch sequence, SQ counters of the
# fused kernel), window / small-set terrain tests
O=gpurun_out/r04f; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -X faulthandler -m pytest tests/test_nuthkaab_gpu.py tests/test_terrain_gpu.py -q -m gpu --maxfail=8 -k "lean or route or C3 or ext or window or generic or options or fbm or randomised or halo or strips or one_rank" > $O/pytest.log 2>&1
tail -12 $O/pytest.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/nktrace -o nk -- python $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 2 > $GRAFT_REPO_ROOT/$O/nktrace.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/nk_sq -o v -- python $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 2 > $GRAFT_REPO_ROOT/$O/nk_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/nk_sq2 -o v -- python $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 2 > $GRAFT_REPO_ROOT/$O/nk_sq2.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/nk_mem -o v -- python $GRAFT_REPO_ROOT/tools/nk_trace.py 20000 2 > $GRAFT_REPO_ROOT/$O/nk_mem.log 2>&1
cd $GRAFT_REPO_ROOT
grep "^step" $O/nktrace.log
python tools/trace_sequence.py $O/nktrace 70 > $O/nk_sequence.txt 2>&1; grep -v "copyBuffer\|select_advance\|bracket_keys\|select_reset\|rebase_shift" $O/nk_sequence.txt | tail -40
python tools/pmc_summary.py $O "nk_fused_kernel" 2>&1 | head -40
find $O -name '*.csv' -size +2M -delete
