"""Decides the switchable third-party conventions from fixtures recorded by oracle/pin_thirdparty.py -- wherever geoutils /
scikit-gstat were importable.  The fixtures cannot be produced in the build image (both packages are absent), so these
tests SKIP there and say so: until they run, the conventions stay "parity unpinned"  and the product keeps them
switchable ("nk_nan_rule", "vario_edge", "vario_diff")."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_geoutils_interp_points_selects_the_nan_rule():
    path = os.path.join(GOLDEN, "thirdparty_interp.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/thirdparty_interp.npz not recorded (geoutils absent here): run oracle/pin_thirdparty.py where it is importable")
    import nuthkaab_oracle as nko

    z = np.load(path)
    dem, res = z["dem"], float(z["res"])
    matches = {}
    for rule in (0, 1, 2, 3):
        ok = True
        for k in range(6):
            sx, sy = z[f"shift{k}"]
            want = z[f"vals{k}"]
            got = nko.bilinear_shifted(dem, -sy / res, sx / res, nan_rule=rule)
            ok &= np.array_equal(np.isnan(got), np.isnan(want)) and np.allclose(got, want, rtol=1e-6, atol=0, equal_nan=True)
        matches[rule] = bool(ok)
    assert sum(matches.values()) >= 1, f"no nk_nan_rule reproduces geoutils' _interp_points: {matches}"
    assert matches[0], f"the default nk_nan_rule (0) is not geoutils' convention: {matches} -- change the default"


def test_skgstat_selects_edge_and_diff_conventions():
    path = os.path.join(GOLDEN, "thirdparty_skgstat.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/thirdparty_skgstat.npz not recorded (scikit-gstat absent here): run oracle/pin_thirdparty.py where it is importable")
    import variogram_oracle as vo

    z = np.load(path)
    coords, values, edges = z["coords"], z["values"], [float(e) for e in z["edges"]]
    found = {}
    for est in ("matheron", "cressie", "dowd"):
        for right_closed in (False, True):
            for diff_f64 in (False, True):
                e, c = vo.empirical_variogram_blocks([(coords[:, 0], coords[:, 1], values)], edges, est, right_closed=right_closed,
                                                     diff_f64=diff_f64)
                same = np.array_equal(c, z[f"count_{est}_float32"]) and np.allclose(e, z[f"exp_{est}_float32"], rtol=1e-9, equal_nan=True)
                found[(est, right_closed, diff_f64)] = bool(same)
    for est in ("matheron", "cressie", "dowd"):
        assert any(v for (e_, _, _), v in found.items() if e_ == est), f"no convention reproduces scikit-gstat for {est}: {found}"
    assert all(found[(est, False, False)] for est in ("matheron", "cressie", "dowd")), \
        f"the defaults (vario_edge 0, vario_diff 0) are not scikit-gstat's conventions: {found} -- change the defaults"


def test_skgstat_equidistant_metric_space_structure():
    """RasterEquidistantMetricSpace: radii list and whether centre-disk x centre-disk pairs exist."""
    path = os.path.join(GOLDEN, "thirdparty_skgstat.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/thirdparty_skgstat.npz not recorded (scikit-gstat absent here): run oracle/pin_thirdparty.py where it is importable")
    z = np.load(path)
    if "rems_radii" not in z or "rems_centers" not in z:
        pytest.skip("this scikit-gstat version does not expose the radii / centres")
    gc, samples, ratio = z["rems_coords"], int(z["rems_samples"]), float(z["rems_ratio"])
    r0 = np.sqrt(samples / (ratio * np.pi)) * 1.0
    diag = np.hypot(np.ptp(gc[:, 0]), np.ptp(gc[:, 1]))
    mine = [0.0]
    r = r0
    while r < diag:
        mine.append(r)
        r *= np.sqrt(2)
    mine.append(diag)
    assert np.allclose(z["rems_radii"], mine, rtol=1e-12), (z["rems_radii"], mine)
    # disk x disk pairs: both endpoints inside the centre disk of one centre
    rows, cols = z["rems_rows"], z["rems_cols"]
    inside = np.zeros(rows.size, dtype=bool)
    for c in z["rems_centers"]:
        da = np.hypot(gc[rows, 0] - c[0], gc[rows, 1] - c[1])
        db = np.hypot(gc[cols, 0] - c[0], gc[cols, 1] - c[1])
        inside |= (da < r0) & (db < r0)
    assert inside.any(), "scikit-gstat holds no disk x disk pairs: drop ring 0 from the equidistant sample again"


def test_geoutils_subsample_array_draw():
    """The restated draw behind NuthKaab's default subsample and the variogram samplers (coreg.subsample_valid_mask:
    default_rng(random_state).choice(valid flat indexes, n, replace=False)) against geoutils' own subsample_array."""
    path = os.path.join(GOLDEN, "thirdparty_subsample.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/thirdparty_subsample.npz not recorded (geoutils absent here): run oracle/pin_thirdparty.py where it is importable")
    from xdem_amd import coreg

    z = np.load(path)
    arr, mask = z["arr"], z["mask"]
    for k in range(int(z["n_cases"])):
        name, subsample, seed = (str(v) for v in z[f"case{k}"])
        valid = np.isfinite(arr) & (~mask if name == "masked" else True)
        sub = float(subsample)
        sub = int(sub) if sub > 1 else sub
        got = coreg.subsample_valid_mask(valid, sub, random_state=int(seed))
        want = np.zeros(arr.shape, dtype=bool)
        want[z[f"rows{k}"], z[f"cols{k}"]] = True
        assert np.array_equal(got, want), (name, subsample, seed)


def test_defaults_follow_the_decision_file():
    """Round 4: where oracle/pin_thirdparty.py could decide a convention from the packages' own outputs it writes
    xdem_amd/thirdparty_decision.json and every context takes its defaults from there.  CPU check of the mechanism: the file, if
    present, holds values in range for known options only, `_lib.thirdparty_decision()` returns exactly them, and the built-in
    defaults are the documented ones (0 / 0 / 0: include/xdemhip.h)."""
    import json

    from xdem_amd import _lib

    assert _lib.THIRDPARTY_DEFAULTS == {"nk_nan_rule": 0, "vario_edge": 0, "vario_diff": 0}
    path = os.environ.get("XDEM_THIRDPARTY_DECISION") or os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "thirdparty_decision.json")
    got = _lib.thirdparty_decision()
    if not os.path.exists(path):
        assert got == {}
        return
    raw = json.load(open(path))
    assert set(k for k in raw if not k.startswith("_")) <= set(_lib.THIRDPARTY_DEFAULTS), raw
    assert got == {k: int(v) for k, v in raw.items() if k in _lib.THIRDPARTY_DEFAULTS}
