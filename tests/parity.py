"""Parity metric shared by the GPU tests and the host-simulation tests.

Bar (BASELINE.json north_star): NaN/nodata masks bit-exact; float32 outputs within 1e-6 relative of the
reference recipe.  "Relative" is taken against max(|ref|, scale_attr) where scale_attr is the 99th
percentile of |ref| over the raster -- the same magnitude-scaled comparison the reference's own tests use
(tests/test_terrain/test_terrain.py:88-102) -- because several attributes pass through zero (aspect wraps,
curvatures change sign) where a pure relative error is meaningless.  The tests additionally report the
fraction of bit-identical pixels.
"""
import numpy as np

RTOL = 1e-6


def compare(got: np.ndarray, ref: np.ndarray, rtol: float = RTOL):
    assert got.shape == ref.shape and got.dtype == ref.dtype, (got.shape, ref.shape, got.dtype, ref.dtype)
    nan_equal = np.array_equal(np.isnan(got), np.isnan(ref))
    fin = np.isfinite(ref) & np.isfinite(got)
    inf_equal = np.array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)])
    if fin.any():
        r = ref[fin].astype(np.float64)
        g = got[fin].astype(np.float64)
        scale = np.percentile(np.abs(r), 99)
        scale = scale if scale > 0 else 1.0
        err = np.abs(g - r) / np.maximum(np.abs(r), scale)
        maxerr = float(err.max())
        exact = float(np.mean(g == r))
    else:
        maxerr, exact = 0.0, 1.0
    return {"nan_equal": nan_equal, "inf_equal": inf_equal, "max_scaled_err": maxerr, "bit_exact_frac": exact}


def assert_parity(got, ref, name="", rtol: float = RTOL, min_exact: float = 0.0):
    c = compare(got, ref, rtol)
    assert c["nan_equal"], f"{name}: NaN mask differs"
    assert c["inf_equal"], f"{name}: +-Inf positions/values differ"
    assert c["max_scaled_err"] <= rtol, f"{name}: scaled error {c['max_scaled_err']:.3e} > {rtol:g}"
    assert c["bit_exact_frac"] >= min_exact, f"{name}: only {c['bit_exact_frac']:.6f} bit-exact"
    return c
