#!/bin/bash
# Round 5, session z: the whole GPU suite of the round's end state (the one-pass step on partitioned plans included) under the NON-default decision file
set -u
mkdir -p gpurun_out/r05z
cat > /tmp/alt_decision.json <<'JSON'
{"nk_nan_rule": 3, "vario_edge": 1, "vario_diff": 1}
JSON
XDEM_THIRDPARTY_DECISION=/tmp/alt_decision.json timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r05z/pytest_alt.log 2>&1
echo "pytest alt rc=$?" >> gpurun_out/r05z/pytest_alt.log
tail -8 gpurun_out/r05z/pytest_alt.log | cut -c1-300
