#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03n}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/host_path_probe.py 16384 8 16 4 > $OUT/hostpath.log 2>&1; grep -v amdgpu $OUT/hostpath.log
timeout 900 python -m pytest tests/test_terrain_gpu.py tests/test_nuthkaab_gpu.py tests/test_cabi_and_host.py -x -q -k "host_path or ext_route or device_side or sharded or randomised_config or cabi or strided" > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
