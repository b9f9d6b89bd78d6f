"""Default conventions of the oracles for the two un-vendored third-party engines -- TEST INFRASTRUCTURE ONLY.

The product takes its defaults of `nk_nan_rule` / `vario_edge` / `vario_diff` from xdem_amd/thirdparty_decision.json (or the file
named by $XDEM_THIRDPARTY_DECISION) where oracle/pin_thirdparty.py could decide them.  The oracles follow the SAME file, so that
the whole GPU suite can be run under a non-default decision (`XDEM_THIRDPARTY_DECISION=<file> pytest -m gpu`): product and
checker switch together, and a decision cannot land on code the suite never exercised.  (Reads a JSON file; imports nothing of
the product.)"""
from __future__ import annotations

import json
import os

_BUILTIN = {"nk_nan_rule": 0, "vario_edge": 0, "vario_diff": 0}


def decided(name: str) -> int:
    path = os.environ.get("XDEM_THIRDPARTY_DECISION") or os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xdem_amd", "thirdparty_decision.json")
    try:
        return int(json.load(open(path)).get(name, _BUILTIN[name]))
    except (OSError, ValueError):
        return _BUILTIN[name]
