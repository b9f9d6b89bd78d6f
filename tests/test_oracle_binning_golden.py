"""Pins oracle/binning_oracle.py against DataFrames recorded from the reference's own nd_binning
(oracle/gen_golden_binning.py -> tests/golden/binning_golden.npz).  Bar: counts, medians, NMADs and bin edges BIT-EXACT."""
import os

import numpy as np
import pytest

import binning_oracle as bo
from conftest import GOLDEN

CASES = ["f32_3var_default", "f32_2var_bins", "f64_1var", "custom_edges", "constant_var"]


def load_case(z, name):
    values = z[f"{name}|values"]
    list_var = []
    while f"{name}|var{len(list_var)}" in z.files:
        list_var.append(z[f"{name}|var{len(list_var)}"])
    b = z[f"{name}|bins"]
    if b.ndim == 0:
        bins = None if int(b) == -1 else int(b)
    elif name == "custom_edges":
        bins = (np.asarray(b, float),)
    else:
        bins = tuple(int(v) for v in b)
    return values, list_var, bins


def flatten_like_reference(results, nv):
    """Concatenate the per-binning grids in DataFrame row order (1-D blocks, 2-D blocks, N-D block; C order each), with the
    interval bounds of every variable (NaN where the variable is not part of the binning).  The N-D block labels follow
    upstream's np.meshgrid(*intervals) ('xy' indexing, spatialstats.py:202)."""
    cols = {"nd": [], "count": [], "nanmedian": [], "nmad": []}
    left = [[] for _ in range(nv)]
    right = [[] for _ in range(nv)]
    for ids, c, m, s, edges in results:
        n = c.size
        cols["nd"].append(np.full(n, len(ids)))
        cols["count"].append(c.ravel())
        cols["nanmedian"].append(m.ravel())
        cols["nmad"].append(s.ravel())
        if len(ids) <= 2:
            grids = np.meshgrid(*[np.arange(len(e) - 1) for e in edges], indexing="ij")
        else:
            grids = np.meshgrid(*[np.arange(len(e) - 1) for e in edges])
        for v in range(nv):
            if v in ids:
                k = ids.index(v)
                g = grids[k].flatten()
                left[v].append(np.asarray(edges[k], float)[g])
                right[v].append(np.asarray(edges[k], float)[g + 1])
            else:
                left[v].append(np.full(n, np.nan))
                right[v].append(np.full(n, np.nan))
    out = {k: np.concatenate(v) for k, v in cols.items()}
    for v in range(nv):
        out[f"v{v}|left"] = np.concatenate(left[v])
        out[f"v{v}|right"] = np.concatenate(right[v])
    return out


@pytest.mark.parametrize("name", CASES)
def test_oracle_equals_reference_nd_binning(name):
    z = np.load(os.path.join(GOLDEN, "binning_golden.npz"))
    values, list_var, bins = load_case(z, name)
    with np.errstate(all="ignore"):
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = flatten_like_reference(bo.nd_binning_arrays(values, list_var, bins), len(list_var))
    for key, arr in got.items():
        ref = z[f"{name}|{key}"]
        assert arr.shape == ref.shape, (name, key)
        assert np.array_equal(np.asarray(arr, np.float64), np.asarray(ref, np.float64), equal_nan=True), (name, key)


RANGE_CASES = ["range_pair_1var", "range_list_1var", "range_equal_1var", "range_wider_than_data"]


def load_range_case(z, name):
    values, list_var = z[f"{name}|values"], [z[f"{name}|var0"]]
    r = z[f"{name}|ranges"]
    ranges = [tuple(float(x) for x in row) for row in r] if bool(z[f"{name}|ranges_is_list"]) else tuple(float(x) for x in r)
    return values, list_var, int(z[f"{name}|bins"]), ranges


@pytest.mark.parametrize("name", RANGE_CASES)
def test_oracle_equals_reference_nd_binning_with_ranges(name):
    """``list_ranges`` as upstream hands it to SciPy's ``range=`` (xdem/spatialstats.py:147): edges from the range, samples outside
    in no bin, a sample on the last edge in the last bin -- the reference's DataFrame, recorded by oracle/gen_golden_binning.py."""
    import warnings

    z = np.load(os.path.join(GOLDEN, "binning_ranges_golden.npz"))
    values, list_var, bins, ranges = load_range_case(z, name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = flatten_like_reference(bo.nd_binning_arrays(values, list_var, bins, list_ranges=ranges), 1)
    assert got["count"].sum() > 0
    for key, arr in got.items():
        ref = z[f"{name}|{key}"]
        assert arr.shape == ref.shape and np.array_equal(np.asarray(arr, np.float64), np.asarray(ref, np.float64), equal_nan=True), (name, key)


def test_oracle_heteroscedasticity_equals_reference():
    """Error map of the reference's _estimate_model_heteroscedasticity + fun(full grid) (spatialstats.py:576-631, 866-868)."""
    import warnings

    z = np.load(os.path.join(GOLDEN, "binning_golden.npz"))
    dh, slope, maxc, stable = z["het|dh"], z["het|slope"], z["het|maxc"], z["het|stable"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res, fun, scale = bo.estimate_model_heteroscedasticity(dh[stable], [slope[stable], maxc[stable]], (8, 6), min_count=30)
        err = scale * fun((slope, maxc))
        probe = scale * fun((z["het|probe_x"], z["het|probe_y"]))
    flat = flatten_like_reference(res, 2)
    assert np.array_equal(flat["count"], z["het|df_count"]) and np.array_equal(flat["nmad"], z["het|df_nmad"], equal_nan=True)
    assert np.array_equal(np.isnan(err), np.isnan(z["het|error"]))
    assert np.allclose(err, z["het|error"], rtol=1e-13, atol=0, equal_nan=True)
    assert np.allclose(probe, z["het|probe_out"], rtol=1e-13, atol=0, equal_nan=True)
