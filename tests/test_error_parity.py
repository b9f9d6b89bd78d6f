"""The mirror's argument validation against the reference's own reactions (exception type and message recorded by
oracle/gen_golden_errors.py into tests/golden/terrain_errors.json).  Validation happens before any GPU work, so this runs
on CPU.  Cases the reference accepts are only checked not to be rejected by the validation step."""
import json
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN

CASES = json.load(open(os.path.join(GOLDEN, "terrain_errors.json")))


@pytest.mark.parametrize("case", CASES, ids=[f"{c['function']}-{i}" for i, c in enumerate(CASES)])
def test_validation_matches_reference(case):
    from xdem_amd import terrain as t

    dem = np.arange(12 * 14, dtype=np.float32).reshape(12, 14) * 0.5
    kw = dict(case["kwargs"])
    if isinstance(kw.get("resolution"), list):
        kw["resolution"] = tuple(kw["resolution"])
    fn = getattr(t, case["function"])
    if case["raises"] is None:
        # accepted upstream: the validation step alone must accept it too (running it needs the GPU: -m gpu tests)
        attribute = kw.pop("attribute")
        t._validate(dem, attribute, kw.get("resolution", 1.0), kw.get("hillshade_altitude", 45.0), kw.get("hillshade_azimuth", 315.0),
                    kw.get("hillshade_z_factor", 1.0), kw.get("surface_fit", "Florinsky"), kw.get("curv_method", "geometric"),
                    kw.get("tri_method", "Riley"), kw.get("window_size_fractal", 13))
        return
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(Exception) as ei:
            fn(dem, **kw)
    assert type(ei.value).__name__ == case["raises"], (case, repr(ei.value))
    assert str(ei.value) == case["message"], (case["function"], case["kwargs"])


SS_CASES = json.load(open(os.path.join(GOLDEN, "spatialstats_errors.json")))


@pytest.mark.parametrize("case", SS_CASES, ids=[f"{c['function']}-{i}" for i, c in enumerate(SS_CASES)])
def test_spatialstats_refusals_match_reference(case):
    """Argument refusals of sample_empirical_variogram (spatialstats.py:1366-1400) and interp_nd_binning (292-352): exception
    type and message as the reference itself produced them (recorded by oracle/gen_golden_errors.py; the named inputs are
    rebuilt by its `build_input` recipe).  Every case is refused before any GPU work."""
    from gen_golden_errors import build_input

    from xdem_amd import spatialstats as ss

    assert case["raises"] is not None
    kw = {k: (build_input(v) if k in ("values", "coords", "df") else v) for k, v in case["kwargs"].items()}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(Exception) as ei:
            getattr(ss, case["function"])(**kw)
    assert type(ei.value).__name__ == case["raises"], (case, repr(ei.value))
    assert str(ei.value) == case["message"], (case["function"], case["kwargs"])
