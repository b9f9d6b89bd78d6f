#!/bin/bash
# round-4 session 17: new GPU tests (list_ranges binning, run-length counting pass on pdist / float64 lattice sets)
O=gpurun_out/r04u; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_binning_gpu.py tests/test_variogram_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
