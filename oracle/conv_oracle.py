"""CPU oracle (TEST INFRASTRUCTURE ONLY -- never imported by the product) of two public functions around the hot paths:

* ``convolution(imgs, filters, method)``        xdem/spatialstats.py:2558-2594
* ``get_perbin_nd_binning(df, list_var, ...)``  xdem/spatialstats.py:425-527  

NumPy restatements, pinned by tests/test_oracle_conv_golden.py against the reference's own outputs
(tests/golden/conv_golden.npz, written by oracle/gen_golden_conv.py from /root/reference) and, for the SciPy engine, against
``scipy.ndimage.convolve`` itself, which is installed wherever the tests run.
"""
from __future__ import annotations

import itertools

import numpy as np


def _convolve_scipy(img: np.ndarray, kern: np.ndarray) -> np.ndarray:
    """scipy.ndimage.convolve(img, kern, mode="constant", cval=nan) for a rectangular kernel of any parity
    (_scipy_convolution, spatialstats.py:2512-2525): out[r, c] = sum k[a, b] img[r + M1//2 - a, c + M2//2 - b]; a double
    accumulator adds weight * value over the weights with |w| > eps, walking the flipped kernel in row-major order; the sum is
    rounded to the image dtype."""
    m1, m2 = kern.shape
    H, W = img.shape
    flipped = kern[::-1, ::-1]
    top, lft = m1 - 1 - m1 // 2, m2 - 1 - m2 // 2          # image offset of flipped tap (0, 0) is (-top, -lft)
    pad = np.full((H + m1 - 1, W + m2 - 1), np.nan, dtype=np.float64)
    pad[top : top + H, lft : lft + W] = img
    acc = np.zeros((H, W), dtype=np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        for a in range(m1):
            for b in range(m2):
                w = flipped[a, b]
                if abs(w) > np.finfo(np.float64).eps:
                    acc = acc + w * pad[a : a + H, b : b + W]
    return acc.astype(img.dtype).astype(np.float64)


def _convolve_numba(img: np.ndarray, kern: np.ndarray) -> np.ndarray:
    """The loop of _numba_convolution (spatialstats.py:2528-2555) on the image padded with (M - 1) // 2 NaNs per side
    (2582-2585): out[r, c] += pad[r + a, c + b] * k[a, b] over every tap, row-major, for r < H + 2 ((M1-1)//2) - M1 + 1 (one
    row short for an even M1: that row keeps its zeros); float64, unrounded."""
    m1, m2 = kern.shape
    H, W = img.shape
    h1, h2 = (m1 - 1) // 2, (m2 - 1) // 2
    pad = np.full((H + 2 * h1, W + 2 * h2), np.nan, dtype=img.dtype)
    pad[h1 : h1 + H, h2 : h2 + W] = img
    rows, cols = H + 2 * h1 - m1 + 1, W + 2 * h2 - m2 + 1
    out = np.zeros((H, W), dtype=np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        for a in range(m1):
            for b in range(m2):
                out[:rows, :cols] = out[:rows, :cols] + pad[a : a + rows, b : b + cols] * kern[a, b]
    return out


def convolution(imgs: np.ndarray, filters: np.ndarray, method: str = "scipy") -> np.ndarray:
    filters = np.asarray(filters, dtype=np.float64)
    one = _convolve_scipy if method.lower() == "scipy" else _convolve_numba
    out = np.zeros((imgs.shape[0], filters.shape[0]) + imgs.shape[1:], dtype=np.float64)
    for i in range(imgs.shape[0]):
        for j in range(filters.shape[0]):
            out[i, j] = one(imgs[i], filters[j])
    return out


def get_perbin_nd_binning(df, list_var, list_var_names, statistic_name: str, min_count=0) -> np.ndarray:
    """spatialstats.py:484-527 on a DataFrame whose variable columns already hold pd.Interval: masks
    ``var >= left & var < right`` per sorted unique interval and variable, the product of the intervals walked in
    itertools.product order, a bin's FIRST row of the DataFrame written where its count exceeds min_count."""
    names = [list_var_names] if isinstance(list_var_names, str) else list(list_var_names)
    sub = df[df.nd == len(names)] if "nd" in df.columns else df
    out = np.full(np.shape(list_var[0]), np.nan)
    ivs = [np.unique(sub[n].values) for n in names]
    member = [[(np.asarray(v) >= iv.left) & (np.asarray(v) < iv.right) for iv in u] for v, u in zip(list_var, ivs)]
    for combo in itertools.product(*[range(len(u)) for u in ivs]):
        inside = np.logical_and.reduce([member[k][j] for k, j in enumerate(combo)])
        if not inside.any():
            continue
        row = np.logical_and.reduce([(sub[n] == ivs[k][j]).values for k, (n, j) in enumerate(zip(names, combo))])
        if sub["count"].values[row][0] > min_count:
            out[inside] = sub[statistic_name].values[row][0]
    return out
