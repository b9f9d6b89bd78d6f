#!/bin/bash
# Round 5, session k: the whole GPU suite under a NON-default decision file (VERDICT task 7b): product and oracles both read
# $XDEM_THIRDPARTY_DECISION = {nk_nan_rule: 3, vario_edge: 1, vario_diff: 1}.
set -u
mkdir -p gpurun_out/r05k
cat > /tmp/alt_decision.json <<'JSON'
{"nk_nan_rule": 3, "vario_edge": 1, "vario_diff": 1}
JSON
export XDEM_THIRDPARTY_DECISION=/tmp/alt_decision.json
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r05k/pytest_alt.log 2>&1
echo "pytest alt rc=$?" >> gpurun_out/r05k/pytest_alt.log
tail -40 gpurun_out/r05k/pytest_alt.log
