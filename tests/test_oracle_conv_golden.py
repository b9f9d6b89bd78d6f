"""oracle/conv_oracle.py against the outputs of the reference itself (tests/golden/conv_golden.npz + conv_errors.json, written by
oracle/gen_golden_conv.py from /root/reference): the generic convolution of SURVEY 8a row a5 in both of upstream's engines --
bit for bit, odd / even / rectangular filters, zero and sub-epsilon weights, NaN and Inf pixels -- and, for the SciPy engine,
against scipy.ndimage.convolve itself; the per-bin lookup of SURVEY 8f row f3 on the reference's own DataFrames."""
import io
import json
import os

import numpy as np
import pandas as pd
import pytest

import conv_oracle as co

HERE = os.path.dirname(__file__)
Z = np.load(os.path.join(HERE, "golden", "conv_golden.npz"))
ERRS = json.load(open(os.path.join(HERE, "golden", "conv_errors.json")))
CONV_CASES = [str(n) for n in Z["conv|names"]]

PERBIN_RUNS = {
    "1var_f32": ("1", ["q_slope"], ["slope"], "nanmedian", 0),
    "1var_f64_min30": ("1", ["q_slope64"], ["slope"], "nmad", 30),
    "1var_name_as_str": ("1", ["q_slope.ravel"], "slope", "nmad", 0),
    "2var": ("2", ["q_slope", "q_curv"], ["slope", "curv"], "nmad", 10),
    "2var_other_order": ("2", ["q_curv", "q_slope64"], ["curv", "slope"], "nanmedian", 0),
    "3var": ("3", ["q_slope", "q_curv", "q_third"], ["slope", "curv", "third"], "nmad", 5),
    "3var_min_huge": ("3", ["q_slope", "q_curv", "q_third"], ["slope", "curv", "third"], "nmad", 10**6),
}


def perbin_frame(d: str) -> pd.DataFrame:
    """nd_binning's DataFrame of the d-variable binning, rebuilt from the recorded columns."""
    cols = {c: Z[f"perbin|df{d}|{c}"] for c in ("nd", "count", "nanmedian", "nmad")}
    for v in ("slope", "curv", "third")[: int(d)]:
        lo, hi = Z[f"perbin|df{d}|{v}|left"], Z[f"perbin|df{d}|{v}|right"]
        cols[v] = np.array([pd.Interval(np.float64(a), np.float64(b), closed="left") if np.isfinite(a) else np.nan for a, b in zip(lo, hi)], dtype=object)
    return pd.DataFrame(cols)


def perbin_vars(keys):
    return [Z["perbin|" + k.split(".")[0]].ravel() if k.endswith(".ravel") else Z["perbin|" + k] for k in keys]


@pytest.mark.parametrize("name", CONV_CASES)
def test_convolution_oracle_equals_the_reference(name):
    imgs, filters = Z[f"conv|{name}|imgs"], Z[f"conv|{name}|filters"]
    for method in ("scipy", "numba"):
        want = Z[f"conv|{name}|{method}"]
        got = co.convolution(imgs, filters, method)
        assert got.dtype == np.float64 and got.shape == want.shape
        assert np.array_equal(got, want, equal_nan=True), (name, method)


def test_the_two_engines_differ_where_the_fixtures_say():
    """The cases are chosen so that the conventions matter: flip, rounding to the image dtype, the zero-weight rule, even sizes."""
    a, b = Z["conv|float32|5x5|scipy"], Z["conv|float32|5x5|numba"]
    assert not np.array_equal(a, b, equal_nan=True)
    assert np.isnan(b[0, 1]).sum() > np.isnan(a[0, 1]).sum()           # a NaN under a zero weight spreads in the Numba loop only
    assert np.all(Z["conv|float64|even_4x4|numba"][0, :, -1, :] == 0.0) and np.all(Z["conv|float64|even_4x4|numba"][0, :, :, -1] == 0.0)
    assert np.array_equal(a.astype(np.float32).astype(np.float64), a, equal_nan=True)     # SciPy engine: float32-valued


def test_scipy_engine_of_the_oracle_equals_scipy_itself():
    import scipy.ndimage

    rng = np.random.default_rng(3)
    for dt in (np.float32, np.float64):
        img = (500 + np.cumsum(rng.normal(size=(29, 33)), axis=1)).astype(dt)
        img[4, 5] = np.nan
        img[20, 2] = np.inf
        for shape in ((3, 3), (5, 5), (4, 6), (1, 8), (7, 2), (6, 6)):
            k = rng.normal(size=shape)
            k[rng.uniform(size=shape) < 0.3] = 0.0
            want = scipy.ndimage.convolve(img, k, mode="constant", cval=np.nan).astype(np.float64)
            got = co.convolution(img[None], k[None], "scipy")[0, 0]
            assert np.array_equal(got, want, equal_nan=True), (dt, shape)


@pytest.mark.parametrize("key", sorted(PERBIN_RUNS))
def test_perbin_oracle_equals_the_reference(key):
    d, vkeys, names, stat, mc = PERBIN_RUNS[key]
    got = co.get_perbin_nd_binning(perbin_frame(d), perbin_vars(vkeys), names, stat, mc)
    assert np.array_equal(got, Z[f"perbin|{key}|out"], equal_nan=True)
    assert np.isfinite(got).any() or key == "3var_min_huge"


def test_perbin_oracle_overlapping_intervals():
    hand = pd.DataFrame({"x": [pd.Interval(0.0, 5.0, closed="left"), pd.Interval(3.0, 8.0, closed="left"), pd.Interval(2.0, 4.0, closed="left")],
                         "count": [10.0, 1.0, 7.0], "val": [1.5, 2.5, 3.5]})
    x = Z["perbin|overlap|x"]
    assert np.array_equal(co.get_perbin_nd_binning(hand, [x], ["x"], "val", 0), Z["perbin|overlap|out_min0"], equal_nan=True)
    assert np.array_equal(co.get_perbin_nd_binning(hand, [x], ["x"], "val", 5), Z["perbin|overlap|out_min5"], equal_nan=True)


def test_recorded_errors_are_the_expected_kinds():
    assert ERRS["bin_without_row"]["type"] == "IndexError" and ERRS["min_count_none"]["type"] == "TypeError"
    assert ERRS["fewer_variables_than_the_binning"]["type"] == "TypeError"
    assert {k for k, v in ERRS.items() if v and v["type"] == "ValueError"} == {"empty_dataframe", "lengths_differ", "method_name", "no_count_column",
                                                                                "unknown_statistic", "unknown_variable"}
