"""Shared pytest configuration: registers the ``gpu`` marker and puts the repo root / oracle on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def decided(name: str) -> int:
    """The third-party convention in force (oracle/_conventions.py): the built-in default unless a decision file -- the product's
    xdem_amd/thirdparty_decision.json or $XDEM_THIRDPARTY_DECISION -- says otherwise.  Tests that switch an option restore THIS value,
    so that `XDEM_THIRDPARTY_DECISION=<file> pytest -m gpu` runs the whole suite, product and oracles alike, under that decision."""
    import _conventions

    return _conventions.decided(name)


def default_conventions() -> bool:
    return all(decided(k) == 0 for k in ("nk_nan_rule", "vario_edge", "vario_diff"))
