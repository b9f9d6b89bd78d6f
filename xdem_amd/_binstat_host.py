"""Host half of a binned statistic the device does not evaluate: ``scipy.stats.binned_statistic_dd``'s treatment of the `statistic`
argument (scipy/stats/_binned_statistic.py, the version-stable part: vectorised answers for the names / NumPy function objects it
knows, a per-bin call for any other callable) on bin numbers the GPU produced.  Used by ``nd_binning(statistics=[...])``
(xdem/spatialstats.py:143-157, 172-195: upstream hands every statistic to SciPy) and by ``NuthKaab(bin_statistic=<callable>)``.
The values of a bin reach the callable in SAMPLE ORDER and in the values' dtype, each bin as a fresh array, as upstream's per-bin
lists do; empty bins receive ``statistic([])``, or NaN if that raises."""
from __future__ import annotations

import warnings

import numpy as np

KNOWN_NAMES = ("mean", "median", "count", "sum", "std", "min", "max")


def _is(stat, name: str, fn) -> bool:
    return (isinstance(stat, str) and stat == name) or stat is fn


def binned_statistic_host(stat, b: np.ndarray, v: np.ndarray, n_bins: int) -> np.ndarray:
    """float64[n_bins]: `stat` of the values `v` (1-D, finite, in sample order) grouped by their bin numbers `b` (same length, each
    in [0, n_bins)).  `stat`: one of SciPy's names, or a callable of a 1-D array."""
    if not callable(stat) and stat not in KNOWN_NAMES:
        raise ValueError(f"invalid statistic {stat!r}")
    b = np.asarray(b).astype(np.intp, copy=False)
    counts = np.bincount(b, minlength=n_bins).astype(np.int64)
    nz = counts > 0
    if _is(stat, "count", None):
        return counts.astype(np.float64)
    if _is(stat, "sum", np.sum):
        return np.bincount(b, weights=v, minlength=n_bins).astype(np.float64)
    if _is(stat, "mean", np.mean) or _is(stat, "std", np.std):
        out = np.full(n_bins, np.nan, dtype=np.float64)
        flatsum = np.bincount(b, weights=v, minlength=n_bins)   # (float64 accumulation in sample order, as SciPy's _bincount)
        if _is(stat, "mean", np.mean):
            out[nz] = flatsum[nz] / counts[nz]
        else:
            delta = v - flatsum[b] / counts[b]
            out[nz] = np.sqrt(np.bincount(b, weights=delta * np.conj(delta), minlength=n_bins)[nz] / counts[nz])
        return out
    order = np.argsort(b, kind="stable")   # (stable: every bin keeps its values in sample order)
    starts = np.concatenate(([0], np.cumsum(counts)))
    vs = v[order]
    if _is(stat, "min", np.min) or _is(stat, "max", np.max):
        out = np.full(n_bins, np.nan, dtype=np.float64)
        red = np.minimum if _is(stat, "min", np.min) else np.maximum
        if nz.any():
            out[nz] = red.reduceat(vs, starts[:-1][nz])
        return out
    if _is(stat, "median", np.median):
        out = np.full(n_bins, np.nan, dtype=np.float64)
        for k in np.flatnonzero(nz):   # SciPy: values sorted within the bin, (v[floor(mid)] + v[ceil(mid)]) / 2 in the values' dtype
            seg = np.sort(vs[starts[k]:starts[k + 1]])
            mid = (seg.size - 1) / 2
            out[k] = (seg[int(np.floor(mid))] + seg[int(np.ceil(mid))]) / 2
        return out
    with np.errstate(invalid="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        try:
            null = stat([])
        except Exception:
            null = np.nan
    out = np.full(n_bins, null, dtype=np.float64)
    for k in np.flatnonzero(nz):
        # (a fresh array per bin, as upstream builds one: NumPy's vectorised reductions peel to the buffer's alignment, so a slice
        #  at an odd offset can sum in another order than the same values at the start of an allocation)
        out[k] = stat(vs[starts[k]:starts[k + 1]].copy())
    return out
