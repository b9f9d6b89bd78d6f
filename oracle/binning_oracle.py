"""CPU oracle of the N-D binned statistics -- TEST INFRASTRUCTURE ONLY.

Restates what ``xdem.spatialstats.nd_binning`` (xdem/spatialstats.py:91-216) computes through
``scipy.stats.binned_statistic / _2d / _dd`` for the statistics count, ``np.nanmedian`` and geoutils' ``nmad``:
joint finite filter (140-143), SciPy's bin edges (``_binned_statistic.py:_bin_edges``: data range as float, +-0.5
when degenerate, ``np.linspace`` in the sample dtype) and bin numbers (``_bin_numbers``: ``np.digitize`` + the
rightmost-edge rounding rule), then the per-bin statistic on the value-dtype array of the bin.

Pinned against outputs of the reference's own nd_binning recorded in tests/golden/binning_golden.npz
(oracle/gen_golden_binning.py).  ``nmad`` lives in geoutils (absent here): its published definition
``nfact * nanmedian(|x - nanmedian(x)|)`` is restated; parity of that one formula is unpinned.
"""
from __future__ import annotations

import itertools

import numpy as np


def nmad(data, nfact: float = 1.4826):
    arr = np.asarray(data)
    return nfact * np.nanmedian(np.abs(arr - np.nanmedian(arr)))


def bin_edges(sample: np.ndarray, bins: list, rng=None) -> tuple[list[np.ndarray], list[int]]:
    """(edges per dimension, SciPy's rounding decimals) for a (N, D) sample matrix; ``rng`` = SciPy's ``range=`` after
    binned_statistic's wrapping (one (start, stop) pair per dimension) or None."""
    edges_dtype = sample.dtype if np.issubdtype(sample.dtype, np.floating) else np.dtype(float)
    if rng is None:
        smin = np.atleast_1d(np.array(sample.min(axis=0), float))
        smax = np.atleast_1d(np.array(sample.max(axis=0), float))
    else:
        if len(rng) != sample.shape[1]:
            raise ValueError(f"range given for {len(rng)} dimensions; {sample.shape[1]} required")
        smin = np.array([float(r[0]) for r in rng])
        smax = np.array([float(r[1]) for r in rng])
        if np.any(smax < smin):
            raise ValueError("range: start must be <= stop")
    edges, decimals = [], []
    for i in range(sample.shape[1]):
        if smin[i] == smax[i]:
            smin[i], smax[i] = smin[i] - 0.5, smax[i] + 0.5
        if np.isscalar(bins[i]):
            e = np.linspace(smin[i], smax[i], int(bins[i]) + 1, dtype=edges_dtype)
        else:
            e = np.asarray(np.asarray(bins[i], float), edges_dtype)
        edges.append(e)
        decimals.append(int(-np.log10(np.diff(e).min())) + 6)
    return edges, decimals


def bin_numbers(sample: np.ndarray, edges: list[np.ndarray], decimals: list[int]) -> np.ndarray:
    """Flattened C-order bin id over the core bins, -1 for outliers."""
    n, nd = sample.shape
    flat = np.zeros(n, np.int64)
    ok = np.ones(n, bool)
    for i in range(nd):
        idx = np.digitize(sample[:, i], edges[i])
        on_edge = (sample[:, i] >= edges[i][-1]) & (np.around(sample[:, i], decimals[i]) == np.around(edges[i][-1], decimals[i]))
        idx[on_edge] -= 1
        nb = len(edges[i]) - 1
        ok &= (idx >= 1) & (idx <= nb)
        flat = flat * nb + (idx - 1)
    flat[~ok] = -1
    return flat


def binned_stats(values: np.ndarray, cols: list[np.ndarray], bins: list, rng=None):
    """count / nanmedian / nmad per bin (float64 arrays shaped like the bin grid) + edges, inputs already filtered."""
    sample = np.atleast_2d(cols).T
    edges, decimals = bin_edges(sample, bins, rng)
    shape = tuple(len(e) - 1 for e in edges)
    ids = bin_numbers(sample, edges, decimals)
    nb = int(np.prod(shape))
    count = np.zeros(nb)
    med = np.full(nb, np.nan)
    nm = np.full(nb, np.nan)
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    start = np.searchsorted(sid, np.arange(nb), side="left")
    stop = np.searchsorted(sid, np.arange(nb), side="right")
    for b in range(nb):
        if stop[b] > start[b]:
            v = values[order[start[b]:stop[b]]]
            count[b] = v.size
            med[b] = np.nanmedian(v)
            nm[b] = nmad(v)
    return count.reshape(shape), med.reshape(shape), nm.reshape(shape), edges


def nd_binning_arrays(values, list_var, list_var_bins=None, list_ranges=None):
    """All binnings nd_binning performs, as a list of (var_ids, count, median, nmad, edges) in its order: every 1-D,
    every 2-D combination, then the N-D one when there are more than two variables.  ``list_ranges`` goes to every binning
    as upstream hands it to SciPy (xdem/spatialstats.py:147, 176, 190; binned_statistic wraps a 2-element range into a list)."""
    nv = len(list_var)
    r1 = list_ranges
    if r1 is not None and len(r1) == 2:
        r1 = [r1]
    if list_var_bins is None:
        list_var_bins = (10,) * nv
    elif isinstance(list_var_bins, (int, np.integer)):
        list_var_bins = (list_var_bins,) * nv
    values = np.asarray(values).ravel()
    list_var = [np.asarray(v).ravel() for v in list_var]
    valid = np.logical_and.reduce([np.isfinite(values)] + [np.isfinite(v) for v in list_var])
    values = values[valid]
    list_var = [v[valid] for v in list_var]
    out = []
    for i in range(nv):
        out.append(((i,),) + binned_stats(values, [list_var[i]], [list_var_bins[i]], r1))
    if nv > 1:
        for i1, i2 in itertools.combinations(range(nv), 2):
            out.append(((i1, i2),) + binned_stats(values, [list_var[i1], list_var[i2]], [list_var_bins[i1], list_var_bins[i2]], list_ranges))
    if nv > 2:
        out.append((tuple(range(nv)),) + binned_stats(values, list_var, list(list_var_bins), list_ranges))
    return out


# ---- heteroscedasticity inference (xdem/spatialstats.py:237-421, 530-631) ----------------------------------------------
def interp_table(centres_per_bin: list[np.ndarray], stat: np.ndarray, count: np.ndarray, min_count):
    """Regular-grid interpolant (scipy RegularGridInterpolator) of a binned statistic given per-bin centre coordinates:
    bins under ``min_count`` dropped, linear griddata inside the hull, nearest-neighbour outside (on the grid, then on
    the grid extended by one node per side), linear interpolation in between -- the documented stages of
    interp_nd_binning."""
    from scipy.interpolate import RegularGridInterpolator, griddata

    stat = np.array(stat, float)
    if min_count is not None:
        stat[count < min_count] = np.nan
    ok = np.isfinite(stat)
    axes = [np.array(sorted(np.unique(c[ok]))) for c in centres_per_bin]
    grid_pts = tuple(m.flatten() for m in np.meshgrid(*axes, indexing="ij"))
    g = griddata(tuple(c[ok] for c in centres_per_bin), stat[ok], grid_pts, method="linear")
    f = np.isfinite(g)
    g = griddata(tuple(p[f] for p in grid_pts), g[f], grid_pts, method="nearest")
    ext = [np.concatenate([[a[0] - 1], a, [a[-1] + 1]]) for a in axes]
    ext_pts = tuple(m.flatten() for m in np.meshgrid(*ext, indexing="ij"))
    ge = griddata(grid_pts, g, ext_pts, method="nearest").reshape([len(a) for a in ext])
    return RegularGridInterpolator(tuple(ext), ge, method="linear", bounds_error=False, fill_value=None)


def estimate_model_heteroscedasticity(dvalues, list_var, list_var_bins=None, min_count=100, fac_spread_outliers=7):
    """(binning results, unscaled interpolant, scale factor): error = scale * interpolant(vars)."""
    import pandas as pd

    res = nd_binning_arrays(dvalues, list_var, list_var_bins)
    nv = len(list_var)
    ids, count, _, nm, edges = [r for r in res if len(r[0]) == nv][-1]
    mids = [pd.IntervalIndex.from_breaks(e, closed="left").mid.values for e in edges]  # (float64 midpoints, base.py / pandas)
    mesh = np.meshgrid(*mids, indexing="ij") if nv <= 2 else np.meshgrid(*mids)
    fun = interp_table([m.flatten() for m in mesh], nm.flatten(), count.flatten(), min_count)
    with np.errstate(all="ignore"):
        z = np.asarray(dvalues).ravel() / fun(tuple(np.asarray(v).ravel() for v in list_var))
    if fac_spread_outliers is not None:
        z[np.abs(z) > fac_spread_outliers * nmad(z)] = np.nan
    return res, fun, nmad(z)
