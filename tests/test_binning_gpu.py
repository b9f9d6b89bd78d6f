"""GPU parity of the N-D binned statistics: xdem_amd.spatialstats.nd_binning (csrc/binstats.hip through the
C-ABI) vs the reference's own DataFrames (tests/golden/binning_golden.npz) and vs the CPU oracle on larger inputs.
Counts, exact medians, NMADs and interval edges: BIT-EXACT."""
import os
import warnings

import numpy as np
import pytest

import binning_oracle as bo
from conftest import GOLDEN
from test_oracle_binning_golden import CASES, flatten_like_reference, load_case

pytestmark = pytest.mark.gpu


def df_to_cols(df, nv):
    out = {"nd": df["nd"].values.astype(np.float64)}
    for c in ("count", "nanmedian", "nmad"):
        out[c] = df[c].values.astype(np.float64)
    for v in range(nv):
        col = df[f"v{v}"].values
        out[f"v{v}|left"] = np.array([iv.left if hasattr(iv, "left") else np.nan for iv in col], float)
        out[f"v{v}|right"] = np.array([iv.right if hasattr(iv, "right") else np.nan for iv in col], float)
    return out


@pytest.mark.parametrize("name", CASES)
def test_nd_binning_equals_reference_dataframe(name):
    from xdem_amd import spatialstats as ss

    z = np.load(os.path.join(GOLDEN, "binning_golden.npz"))
    values, list_var, bins = load_case(z, name)
    names = [f"v{i}" for i in range(len(list_var))]
    df = ss.nd_binning(values, list_var, names, list_var_bins=bins, statistics=["count", np.nanmedian, ss.nmad])
    assert list(df.columns) == ["nd", "count", "nanmedian", "nmad"] + names
    assert df["count"].dtype == np.float64
    got = df_to_cols(df, len(list_var))
    for key, arr in got.items():
        ref = np.asarray(z[f"{name}|{key}"], np.float64)
        assert arr.shape == ref.shape and np.array_equal(arr, ref, equal_nan=True), (name, key)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_nd_binning_vs_oracle_large(dtype):
    """2e6 samples, 3 variables of mixed dtype, 10 x 10 x 10 + all lower-dimensional binnings, NaN / Inf rows."""
    from xdem_amd import spatialstats as ss

    rng = np.random.default_rng(3)
    n = 2_000_000
    slope = rng.gamma(2.0, 8.0, n).astype(np.float32)
    curv = np.abs(rng.normal(0, 1.5, n)).astype(dtype)
    elev = rng.uniform(0, 3000, n)
    dh = (rng.normal(0, 1, n) * (0.5 + 0.05 * slope + 0.3 * curv)).astype(dtype)
    dh = np.round(dh, 2)  # plenty of ties
    dh[::1001] = np.nan
    slope[3::5003] = np.inf
    df = ss.nd_binning(dh, [slope, curv, elev], ["v0", "v1", "v2"])
    got = df_to_cols(df, 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = flatten_like_reference(bo.nd_binning_arrays(dh, [slope, curv, elev]), 3)
    assert got["count"].sum() > 3 * n
    for key, arr in got.items():
        assert np.array_equal(arr, np.asarray(ref[key], np.float64), equal_nan=True), key


def test_nd_binning_argument_rules():
    from xdem_amd import spatialstats as ss

    v = np.arange(100, dtype=np.float32)
    with pytest.raises(NotImplementedError, match="not available on the HIP engine"):
        ss.nd_binning(v, [v], ["a"], statistics=[np.nanmean])
    with pytest.raises(NotImplementedError, match="list_ranges"):
        ss.nd_binning(v, [v], ["a"], list_ranges=[0, 1])
    df = ss.nd_binning(v, [v], ["a"], list_var_bins=4, statistics=[np.nanmedian])  # count is added in front
    assert list(df.columns) == ["nd", "count", "nanmedian", "a"] and df["count"].tolist() == [25.0] * 4
    assert df["nanmedian"].tolist() == [12.0, 37.0, 62.0, 87.0]
