"""CPU oracle for the Nuth & Kaab (2011) inner loop -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

NumPy restatement (own code) of the reference's raster-raster Nuth-Kaab path:

* auxiliary variables, once per fit: ``np.gradient`` -> slope tangent and aspect
  (xdem/coreg/affine.py:412-474), zero-slope removal ``np.isclose(slope_tan, 0)`` (affine.py:578-579),
  valid mask (xdem/coreg/base.py:650-661);
* per iteration (affine.py:477-536): elevation difference at the shifted position, ``nanmedian`` vertical
  shift, ``y = dh / slope_tan``, initial guess ``(3 nanstd(y)/sqrt 2, 0, nanmean(y))`` (affine.py:381-384),
  72-bin aspect binning with ``np.nanmedian`` per bin through ``scipy.stats.binned_statistic``
  (xdem/spatialstats.py:143-157; SciPy 1.15 ``_binned_statistic.py``: edges = linspace(min, max, 73) cast to
  the sample dtype, ``np.digitize``, points on the last edge moved into the last bin), bin mid-points
  (xdem/coreg/base.py:1027), ``scipy.optimize.curve_fit`` of ``a cos(b - x) + c`` (affine.py:340-355,
  base.py:1038-1045), offsets update and stop rule ``i > 1 and stat < tol`` (affine.py:102-147, 526-534).

PINNED against the reference: tests/test_oracle_golden.py compares ``aux_vars`` / ``bin_medians`` /
``bin_fit`` / ``iterate`` with vectors recorded from the reference's own functions (oracle/gen_golden_nk.py).

PARITY UNPINNED for one piece: the reference interpolates the shifted DEM with geoutils' ``_interp_points``
(geoutils==0.2.5, un-vendored, absent here).  The convention used by this oracle AND by the HIP kernel:
bilinear on the pixel grid, sample position (row - shift_y / res_y, col + shift_x / res_x), float64 weights,
result rounded to the DEM dtype, NaN if any of the four taps is non-finite or outside the raster.
One more platform note: the reference's aspect is ``np.arctan2`` on float32 (libm ``atan2f``, up to 1 ulp
off); the oracle and the kernel use the correctly rounded float32 of the float64 arctangent.
"""
from __future__ import annotations

import numpy as np

import _conventions


def gradient_unit(dem: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """np.gradient(dem) with unit spacing in the DEM dtype: central differences, one-sided at the borders."""
    dt = dem.dtype
    gy = np.empty_like(dem)
    gx = np.empty_like(dem)
    two = dt.type(2.0)
    with np.errstate(invalid="ignore"):
        if dem.shape[0] > 1:
            gy[1:-1] = (dem[2:] - dem[:-2]) / two
            gy[0] = dem[1] - dem[0]
            gy[-1] = dem[-1] - dem[-2]
        if dem.shape[1] > 1:
            gx[:, 1:-1] = (dem[:, 2:] - dem[:, :-2]) / two
            gx[:, 0] = dem[:, 1] - dem[:, 0]
            gx[:, -1] = dem[:, -1] - dem[:, -2]
    return gy, gx


def aux_vars(ref: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """slope tangent and aspect (radians) of affine.py:433-438, then zero slopes -> NaN (affine.py:578-579)."""
    dt = ref.dtype
    gy, gx = gradient_unit(ref)
    with np.errstate(invalid="ignore"):
        slope_tan = np.sqrt(gx**2 + gy**2)
        aspect = np.arctan2(-gx.astype(np.float64), gy.astype(np.float64)).astype(dt)
        aspect = aspect + dt.type(np.pi)
        slope_tan[np.abs(slope_tan) <= 1e-8] = np.nan  # np.isclose(x, 0): atol 1e-8 (rtol * 0 = 0)
    return slope_tan, aspect


def bilinear_shifted(img: np.ndarray, dr: float, dc: float, nan_rule: int | None = None) -> np.ndarray:
    """bilinear(img)(row + dr, col + dc) on the full grid, float64 weights, result in img's dtype.  ``nan_rule`` = the
    switchable nodata convention of the kernel (geoutils' own rule is unpinned, see header): 0 "4tap" -- NaN if any of the
    four taps is non-finite or outside, zero weights included (except a zero-weight tap beyond the last row / column: nodes on
    the upper edge keep their value); 1 "weighted" -- taps with zero weight are ignored; 2
    "dilate3x3" -- NaN if the 3 x 3 neighbourhood of the nearest pixel holds a non-finite pixel or leaves the raster; 3
    "dilate_cross" -- the same with the 4-connected cross."""
    if nan_rule is None:   # the decided convention (oracle/_conventions.py: the product's default, 0 unless a decision file says otherwise)
        nan_rule = _conventions.decided("nk_nan_rule")
    H, W = img.shape
    rr = np.arange(H, dtype=np.float64)[:, None] + dr
    cc = np.arange(W, dtype=np.float64)[None, :] + dc
    r0 = np.floor(rr)
    c0 = np.floor(cc)
    fr = rr - r0
    fc = cc - c0
    r0 = r0.astype(np.int64)
    c0 = c0.astype(np.int64)
    # zero-weight taps: ignored by rule 1 everywhere; by the other rules only where the tap would leave the raster (a node
    # exactly on the upper edge is returned by every linear interpolator)
    need_r1 = (fr != 0) if nan_rule == 1 else ((fr != 0) | (r0 + 1 < H))
    need_c1 = (fc != 0) if nan_rule == 1 else ((fc != 0) | (c0 + 1 < W))
    r1 = np.where(need_r1, r0 + 1, r0)
    c1 = np.where(need_c1, c0 + 1, c0)
    ok = np.broadcast_to((r0 >= 0) & (r1 < H), (H, W)) & np.broadcast_to((c0 >= 0) & (c1 < W), (H, W))
    r0c, r1c = np.clip(r0, 0, H - 1), np.clip(r1, 0, H - 1)
    c0c, c1c = np.clip(c0, 0, W - 1), np.clip(c1, 0, W - 1)
    t = img.astype(np.float64)
    with np.errstate(invalid="ignore"):
        v00 = t[r0c, c0c]
        v01 = t[r0c, c1c]
        v10 = t[r1c, c0c]
        v11 = t[r1c, c1c]
        top = v00 + fc * (v01 - v00)
        bot = v10 + fc * (v11 - v10)
        val = top + fr * (bot - top)
        finite = np.isfinite(v00) & np.isfinite(v01) & np.isfinite(v10) & np.isfinite(v11)
        good = ok & finite
        if nan_rule >= 2:
            bad = ~np.isfinite(img)
            pad = np.ones((H + 2, W + 2), dtype=bool)
            pad[1:-1, 1:-1] = bad
            dil = np.zeros((H, W), dtype=bool)
            for a in range(3):
                for b in range(3):
                    if nan_rule == 2 or a == 1 or b == 1:  # rule 3: the 4-connected cross (SciPy's default dilation structure)
                        dil |= pad[a : a + H, b : b + W]
            rn = np.floor(rr + 0.5).astype(np.int64)
            cn = np.floor(cc + 0.5).astype(np.int64)
            inside = np.broadcast_to((rn >= 0) & (rn < H), (H, W)) & np.broadcast_to((cn >= 0) & (cn < W), (H, W))
            near_bad = np.where(inside, dil[np.clip(rn, 0, H - 1), np.clip(cn, 0, W - 1)], True)
            good = good & ~near_bad
        return np.where(good, val, np.nan).astype(img.dtype)


def shifted_dh(ref: np.ndarray, tba: np.ndarray, shift_x: float, shift_y: float, res: tuple[float, float],
               nan_rule: int | None = None) -> np.ndarray:
    """ref - bilinear(tba)(row - shift_y/res_y, col + shift_x/res_x) on the full grid (stated convention, see header)."""
    return ref - bilinear_shifted(tba, -shift_y / res[1], shift_x / res[0], nan_rule)


def bin_edges(x: np.ndarray, n_bins: int) -> np.ndarray:
    """SciPy's _bin_edges for an integer bin count: linspace(min, max, n+1) in double, cast to the sample dtype."""
    smin, smax = float(x.min()), float(x.max())
    if smin == smax:
        smin, smax = smin - 0.5, smax + 0.5
    return np.linspace(smin, smax, n_bins + 1, dtype=x.dtype)


def bin_index(x: np.ndarray, edges: np.ndarray) -> np.ndarray:
    """0-based bin of each sample (-1 / n for outliers): np.digitize + the 'on the last edge' rule of SciPy."""
    idx = np.digitize(x, edges)
    dedges_min = np.diff(edges).min()
    decimal = int(-np.log10(dedges_min)) + 6
    on_edge = (x >= edges[-1]) & (np.around(x, decimal) == np.around(edges[-1], decimal))
    idx = idx - on_edge.astype(idx.dtype)
    return idx - 1


def bin_medians(x: np.ndarray, y: np.ndarray, n_bins: int = 72):
    """(edges, counts int64[n], medians float64[n] with NaN for empty bins) of binned_statistic(x, y, np.nanmedian, n)."""
    edges = bin_edges(x, n_bins)
    b = bin_index(x, edges)
    counts = np.bincount(b[(b >= 0) & (b < n_bins)], minlength=n_bins).astype(np.int64)
    med = np.full(n_bins, np.nan)
    order = np.lexsort((y, b))
    bs, ys = b[order], y[order]
    start = np.searchsorted(bs, np.arange(n_bins), side="left")
    for k in range(n_bins):
        n = counts[k]
        if n:
            seg = ys[start[k] : start[k] + n]
            lo, hi = seg[(n - 1) // 2], seg[n // 2]
            # np.nanmedian of an even-sized float array: mean of the two middle values in the array dtype
            med[k] = lo if n % 2 else np.mean(np.array([lo, hi], dtype=y.dtype))
    return edges, counts, med


def bin_means(x: np.ndarray, y: np.ndarray, n_bins: int = 72):
    """(edges, counts, means) of binned_statistic(x, y, np.nanmean, n): SciPy calls the statistic on ``y[bin == k]`` for
    every bin (callable path, _binned_statistic.py:648-657) -- NumPy's mean in the dtype of y; empty bins give NaN."""
    edges = bin_edges(x, n_bins)
    b = bin_index(x, edges)
    counts = np.bincount(b[(b >= 0) & (b < n_bins)], minlength=n_bins).astype(np.int64)
    means = np.full(n_bins, np.nan)
    for k in range(n_bins):
        if counts[k]:
            means[k] = np.nanmean(y[b == k])
    return edges, counts, means


def fit_func(xx, a, b, c):
    """affine.py:340-355."""
    return a * np.cos(b - xx) + c


def bin_fit(dh: np.ndarray, slope_tan: np.ndarray, aspect: np.ndarray, n_bins: int = 72):
    """_nuth_kaab_bin_fit (affine.py:358-409) for the default bin_and_fit: returns (easting, northing, c), details."""
    import scipy.optimize

    with np.errstate(divide="ignore", invalid="ignore"):
        y = dh / slope_tan
    p0 = (3 * np.nanstd(y) / (2**0.5), 0.0, np.nanmean(y))
    ok = np.isfinite(y) & np.isfinite(aspect)
    edges, counts, med = bin_medians(aspect[ok], y[ok], n_bins)
    mids = ((edges[:-1] + edges[1:]) / 2).astype(np.float64) if edges.dtype == np.float64 else (
        0.5 * (edges[:-1].astype(np.float64) + edges[1:].astype(np.float64)))
    good = np.isfinite(med)
    if not good.any():
        raise ValueError("Only NaN values after binning, did you pass the right bin edges?")
    popt, _ = scipy.optimize.curve_fit(fit_func, mids[good], med[good], p0=p0, absolute_sigma=True)
    a, b, c = popt
    return (a * np.sin(b), a * np.cos(b), c), {"edges": edges, "counts": counts, "medians": med, "mids": mids,
                                                "p0": np.array(p0, dtype=np.float64), "popt": popt}


def iteration_step(offsets, ref, tba, valid, slope_tan, aspect, res, n_bins: int = 72, nan_rule: int | None = None):
    """_nuth_kaab_iteration_step (affine.py:477-536) on the full grid restricted to `valid`."""
    dh = shifted_dh(ref, tba, offsets[0], offsets[1], res, nan_rule)[valid]
    vshift = np.nanmedian(dh)
    dh = dh - vshift
    ok = np.isfinite(dh)
    if not ok.any():
        raise ValueError("The subsample contains no more valid values.")
    (e, n, _), det = bin_fit(dh[ok], slope_tan[valid][ok], aspect[valid][ok], n_bins)
    new = (offsets[0] + e * res[0], offsets[1] + n * res[1], float(vshift))
    det["n_valid"] = int(ok.sum())
    det["vshift"] = float(vshift)
    return new, float(np.sqrt(e**2 + n**2)), det


def nuth_kaab(ref: np.ndarray, tba: np.ndarray, inlier_mask: np.ndarray | None, res: tuple[float, float],
              tolerance: float = 0.001, max_iterations: int = 10, n_bins: int = 72, nan_rule: int | None = None):
    """nuth_kaab (affine.py:539-609) with subsample == 1: returns ((east, north, vertical), n_valid0, trace).  `nan_rule`: the nodata
    convention of the interpolator (None = the decided one)."""
    slope_tan, aspect = aux_vars(ref)
    if inlier_mask is None:
        inlier_mask = np.ones(ref.shape, dtype=bool)
    valid = inlier_mask & np.isfinite(ref) & np.isfinite(tba) & np.isfinite(slope_tan) & np.isfinite(aspect)
    if not valid.any():
        raise ValueError("There is no valid points common to the input and auxiliary data.")
    offsets = (0.0, 0.0, 0.0)
    trace = []
    for i in range(max_iterations):
        offsets, stat, det = iteration_step(offsets, ref, tba, valid, slope_tan, aspect, res, n_bins, nan_rule)
        trace.append((offsets, stat, det))
        if i > 1 and stat < tolerance:
            break
    return offsets, int(valid.sum()), trace
