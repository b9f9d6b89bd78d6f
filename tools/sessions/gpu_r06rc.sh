#!/bin/bash
# round 6, session rc: the whole GPU suite on the library after the two-pass route was retired, under the default conventions and under the
# NON-default decision file (product and oracles both follow it)
O=gpurun_out/r06rc; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_default.log 2>&1; echo "pytest default rc=$?" | tee -a $O/pytest_default.log
tail -4 $O/pytest_default.log | cut -c1-300
cat > /tmp/alt_decision.json <<'JSON'
{"nk_nan_rule": 3, "vario_edge": 1, "vario_diff": 1}
JSON
XDEM_THIRDPARTY_DECISION=/tmp/alt_decision.json timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_alt.log 2>&1; echo "pytest alt rc=$?" | tee -a $O/pytest_alt.log
tail -6 $O/pytest_alt.log | cut -c1-300
