"""Drop-in check of the Python mirror: every parameter of the reference's entry points on the three hot paths (recorded
from the reference by oracle/gen_golden_signatures.py into tests/golden/signatures.json) exists in xdem_amd's function of the
same name, in the same order and with the same plain-literal default.  The mirror may add trailing keyword parameters."""
import inspect
import json
import os

import pytest

from conftest import GOLDEN

SIG = json.load(open(os.path.join(GOLDEN, "signatures.json")))
# parameters the mirror deliberately treats differently (documented where they are handled)
ALLOWED_DEFAULT_DIFFS = {
    ("terrain", "*", "engine"),  # reference default "scipy"; here every engine name runs on the GPU, default "hip"
    ("dem", "coregister_3d", "coreg_method"),  # required upstream (docstring: "default is ... Nuth and Kaab"); optional here
}


def _mirror(module: str, name: str):
    if module == "terrain":
        from xdem_amd import terrain as m
    elif module == "spatialstats":
        from xdem_amd import spatialstats as m
    elif module in ("surfit", "window", "freq"):   # the reference's xdem.terrain.<module>, reachable here as xdem_amd.terrain.<module>
        import xdem_amd

        return getattr(getattr(xdem_amd.terrain, module), name)
    elif module == "dem":
        from xdem_amd import dem as m

        return getattr(m.DEM, name)
    else:
        from xdem_amd import coreg as m

        return getattr(m.NuthKaab, name.split(".")[1])
    return getattr(m, name)


CASES = [(mod, name) for mod in SIG for name in SIG[mod]]


@pytest.mark.parametrize("module,name", CASES)
def test_reference_parameters_are_mirrored(module, name):
    fn = _mirror(module, name)
    mine = list(inspect.signature(fn).parameters.items())
    mine_names = [n for n, _ in mine]
    if any(p.kind is inspect.Parameter.VAR_KEYWORD for _, p in mine):
        catch_all = True
    else:
        catch_all = False
    pos = -1
    for rec in SIG[module][name]:
        if rec["kind"] in ("VAR_KEYWORD", "VAR_POSITIONAL"):
            continue
        pname = rec["name"]
        if pname not in mine_names:
            assert catch_all, f"{module}.{name}: parameter '{pname}' of the reference is missing"
            continue
        i = mine_names.index(pname)
        assert i > pos, f"{module}.{name}: parameter '{pname}' is out of order"
        pos = i
        ref_default = rec["default"]
        p = mine[i][1]
        if (module, "*", pname) in ALLOWED_DEFAULT_DIFFS or (module, name, pname) in ALLOWED_DEFAULT_DIFFS:
            continue
        if ref_default == "<required>":
            assert p.default is inspect.Parameter.empty, f"{module}.{name}: '{pname}' must stay required"
        elif ref_default == "<object>":
            assert p.default is not inspect.Parameter.empty, f"{module}.{name}: '{pname}' must have a default"
        else:
            d = list(p.default) if isinstance(p.default, (tuple, list)) else p.default
            assert d == ref_default, f"{module}.{name}: default of '{pname}' is {p.default!r}, reference has {ref_default!r}"
