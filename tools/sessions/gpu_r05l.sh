#!/bin/bash
# Round 5, session l: after session k's three failures under the non-default decision file -- those tests again under it, then the
# whole GPU suite under it once more, then the touched files under the default conventions.
set -u
mkdir -p gpurun_out/r05l
cat > /tmp/alt_decision.json <<'JSON'
{"nk_nan_rule": 3, "vario_edge": 1, "vario_diff": 1}
JSON
XDEM_THIRDPARTY_DECISION=/tmp/alt_decision.json timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r05l/pytest_alt.log 2>&1
echo "pytest alt rc=$?" >> gpurun_out/r05l/pytest_alt.log
tail -8 gpurun_out/r05l/pytest_alt.log
timeout 900 python -m pytest tests/test_nuthkaab_gpu.py tests/test_variogram_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/r05l/pytest_default.log 2>&1
echo "pytest default rc=$?" >> gpurun_out/r05l/pytest_default.log
tail -8 gpurun_out/r05l/pytest_default.log
