"""Nuth & Kaab (2011) co-registration on MI355X -- host-side mirror of ``xdem.coreg.NuthKaab`` for the
raster-raster hot path.

Same constructor arguments, ``fit`` contract and ``meta`` outputs as the reference class
(``xdem/coreg/affine.py:2386-2541``) and the same iteration driver (``affine.py:102-147, 477-609``); every pass
over the grids (gradient, shifted elevation difference, exact ``nanmedian`` vertical shift, 72-bin aspect binning
with exact per-bin ``nanmedian``) runs in ``csrc/nuthkaab.hip`` through ``xdemhip_nk_create`` / ``xdemhip_nk_step``.
The 72-point ``scipy.optimize.curve_fit`` stays on the host exactly as upstream (``xdem/coreg/base.py:1038-1045``).

Scope: two rasters on the same grid given as arrays (+ resolution); ``bin_before_fit=True`` (default) or ``False``;
``bin_statistic`` = ``np.nanmedian`` (default: exact medians on the GPU), ``np.nanmean`` (per-bin sums on the GPU) or any other
callable (the GPU hands y and the bin ids of every pixel back and the callable runs on the host per bin, exactly as
``scipy.stats.binned_statistic`` calls it upstream -- single-process fits only).  ``subsample=1`` uses all valid pixels (the
BASELINE configuration); any other value a random subset drawn once by the restated rule of geoutils' ``subsample_array``
(``subsample_valid_mask``).  Point-cloud inputs are outside the hot path and raise ``NotImplementedError``.
"""
from __future__ import annotations

import ctypes
import logging
import warnings
from typing import Any, Callable

import numpy as np

from . import _lib


def _nuth_kaab_fit_func(xx, *params):
    """y(x) = a * cos(b - x) + c  (xdem/coreg/affine.py:340-355)."""
    return params[0] * np.cos(params[1] - xx) + params[2]


def _bin_statistic_id(bin_statistic) -> int:
    """How ``NuthKaab(bin_statistic=...)`` (affine.py:2404) is evaluated: 0 = exact medians on the GPU (np.nanmedian, the reference
    default), 1 = per-bin sums on the GPU (np.nanmean), 2 = any other callable -- the GPU produces y and the bin ids of every pixel, the
    callable runs on the host over each bin's values in raster order, as ``scipy.stats.binned_statistic`` calls it upstream
    (``NKPlan._step_callable``; whole-raster plans only)."""
    if any(bin_statistic is f for f in (np.nanmedian, np.median)) or bin_statistic in ("median", "nanmedian"):
        return 0
    if any(bin_statistic is f for f in (np.nanmean, np.mean)) or bin_statistic in ("mean", "nanmean"):
        return 1
    if callable(bin_statistic):
        return 2
    raise TypeError("bin_statistic must be a callable (np.nanmedian, np.nanmean, or any function of a 1-D array)")


class HaloTooSmall(_lib.XdemHipError):
    """A step of a partitioned (row-block) plan would sample tba outside the rank's halo rows: re-create the plan with a
    deeper halo (``xdem_amd.dist.nuth_kaab_row_blocks`` does)."""


class NKPlan:
    """Device-resident state of one fit (``xdemhip_nk_plan``)."""

    def __init__(self, ref: np.ndarray, tba: np.ndarray, inlier_mask: np.ndarray | None, ctx: _lib.Context | None = None,
                 group=None, block: tuple[int, int, int, int, int] | None = None):
        """``group``: a torch.distributed process group (or "world") to shard the grid passes over its ranks by row
        block; histograms / counts are all-reduced (exact).  Two layouts:

        * replicated (``block=None``): every rank passes the same full rasters and works on its rows of them;
        * partitioned (``block=(H, row_begin, row_end, halo_top, halo_bottom)``): ``ref`` / ``tba`` / ``inlier_mask`` hold only
          raster rows ``[row_begin - halo_top, row_end + halo_bottom)`` of an ``H``-row raster -- the rank's row block plus
          halo rows copied from its neighbours (``xdem_amd.dist.nuth_kaab_row_blocks`` builds and exchanges them).  One
          halo row serves the gradient; the bilinear taps need ``floor(|shift_y| / res_y) + 1`` more."""
        self.ctx = ctx or _lib.default_context()
        self.handle = None
        self.group = None
        self.ctx.adopt(self)   # (closed with the context if the caller never closes it: the plan holds a pointer to the context)
        prev_group = getattr(self.ctx, "_group", None)   # hooks the caller had on the context before this plan
        try:
            self._create(ref, tba, inlier_mask, group, block)
        except BaseException:
            # a creation that fails after the reduction hook went in must not leave it on the (usually process-wide) context
            # -- a later single-process call on it would enter a collective alone -- and must not take away hooks the caller
            # had installed before either: the previous state comes back
            if group is not None:
                self.ctx.set_allreduce(prev_group)
            if self.handle:
                self.ctx._L.xdemhip_nk_destroy(self.handle)
                self.handle = None
            raise

    def _create(self, ref, tba, inlier_mask, group, block) -> None:
        h = ctypes.c_void_p()
        nv = ctypes.c_int64()
        L = self.ctx._L
        if block is not None:
            if group is not None:
                self.ctx.set_allreduce(group)  # (the creation pass already counts the valid pixels globally)
            H, rb, re_, ht, hb = (int(v) for v in block)

            def create(rp, tp, ip, dt, nrows, W, space):
                if nrows != (re_ - rb) + ht + hb:
                    raise ValueError(f"block arrays hold {nrows} rows, expected {(re_ - rb) + ht + hb}")
                return L.xdemhip_nk_create_block(self.ctx.handle, rp, tp, ip, dt, H, W, rb, re_, ht, hb, space, ctypes.byref(h),
                                                 ctypes.byref(nv))
        else:
            def create(rp, tp, ip, dt, nrows, W, space):
                return L.xdemhip_nk_create(self.ctx.handle, rp, tp, ip, dt, nrows, W, space, ctypes.byref(h), ctypes.byref(nv))
        if hasattr(ref, "is_cuda"):
            # device-resident rasters (torch CUDA/HIP tensors, same dtype, contiguous): no host copies; the caller keeps them alive
            if not (ref.is_cuda and tba.is_cuda and ref.is_contiguous() and tba.is_contiguous() and ref.dtype == tba.dtype
                    and ref.shape == tba.shape and ref.dim() == 2):
                raise ValueError("device inputs must be contiguous 2D CUDA tensors of the same shape and dtype")
            import torch

            self.dtype = np.dtype({torch.float32: np.float32, torch.float64: np.float64}[ref.dtype])
            self.shape = tuple(ref.shape)
            if inlier_mask is not None and not hasattr(inlier_mask, "is_cuda"):
                # (a host mask next to device rasters, e.g. the random subsample drawn by nuth_kaab)
                inlier_mask = torch.from_numpy(np.ascontiguousarray(inlier_mask, dtype=np.uint8)).to(ref.device)
            self._keep = (ref, tba, inlier_mask)
            inl_ptr = None
            if inlier_mask is not None:
                if not (inlier_mask.is_cuda and inlier_mask.dtype == torch.uint8 and inlier_mask.is_contiguous()):
                    raise ValueError("device inlier mask must be a contiguous uint8 CUDA tensor")
                inl_ptr = inlier_mask.data_ptr()
            torch.cuda.current_stream(ref.device).synchronize()
            self.ctx.check(create(ref.data_ptr(), tba.data_ptr(), inl_ptr, _lib.F32 if self.dtype == np.float32 else _lib.F64,
                                  self.shape[0], self.shape[1], _lib.DEVICE))
        else:
            ref = np.ascontiguousarray(ref)
            tba = np.ascontiguousarray(tba)
            if ref.shape != tba.shape or ref.ndim != 2:
                raise ValueError("ref and tba must be 2D arrays of the same shape")
            if ref.dtype != tba.dtype or ref.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
                dt = np.float64 if np.float64 in (ref.dtype, tba.dtype) else np.float32
                ref, tba = ref.astype(dt), tba.astype(dt)
            self.dtype = ref.dtype
            self.shape = ref.shape
            inl = None
            if inlier_mask is not None:
                inl = np.ascontiguousarray(inlier_mask, dtype=np.uint8)
            self.ctx.check(create(ref.ctypes.data, tba.ctypes.data, inl.ctypes.data if inl is not None else None,
                                  _lib.F32 if self.dtype == np.float32 else _lib.F64, ref.shape[0], ref.shape[1], _lib.HOST))
        self.handle = h
        self.n_valid = int(nv.value)
        self.group = group
        if group is not None and block is None:
            import torch.distributed as dist

            from .dist import row_block

            pg = None if group == "world" else group
            r0, r1 = row_block(self.shape[0], dist.get_world_size(pg), dist.get_rank(pg))
            self.ctx.set_allreduce(group)
            self.ctx.check(self.ctx._L.xdemhip_nk_set_rows(self.handle, r0, r1, ctypes.byref(nv)))
            self.n_valid = int(nv.value)

    def step(self, shift_x: float, shift_y: float, res: tuple[float, float], n_bins: int = 72) -> dict[str, Any]:
        n_bins = getattr(self, "_n_custom_bins", None) or int(n_bins)
        if getattr(self, "_bin_callable", None) is not None:
            return self._step_callable(shift_x, shift_y, res, n_bins)
        edges = np.empty(n_bins + 1, dtype=np.float64)
        counts = np.empty(n_bins, dtype=np.int64)
        med = np.empty(n_bins, dtype=np.float64)
        vshift, ymean, ystd = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        nv = ctypes.c_int64()
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)
        rc = self.ctx._L.xdemhip_nk_step(self.handle, float(shift_x), float(shift_y), float(res[0]), float(res[1]), int(n_bins),
                                         ctypes.byref(vshift), ctypes.byref(nv), ctypes.byref(ymean), ctypes.byref(ystd),
                                         edges.ctypes.data_as(dp), counts.ctypes.data_as(ip), med.ctypes.data_as(dp))
        if rc != _lib.OK:
            self._raise_step_error(rc)
        return {"vshift": vshift.value, "n_valid": int(nv.value), "y_mean": ymean.value, "y_std": ystd.value,
                "edges": edges, "counts": counts, "medians": med}

    def _step_callable(self, shift_x: float, shift_y: float, res: tuple[float, float], n_bins: int) -> dict[str, Any]:
        """The step under a ``bin_statistic`` that is neither the median nor the mean: ``xdemhip_nk_step_values`` returns y =
        (dh - vshift) / slope_tan and the aspect-bin id of every pixel (raster order); the callable is applied per bin exactly as
        ``scipy.stats.binned_statistic`` does it upstream (xdem/spatialstats.py:143-157 after the finite-value filter of
        ``nd_binning``, spatialstats.py:122-131): empty bins receive ``statistic([])``, or NaN if that raises."""
        if self.group is not None:
            raise NotImplementedError("a bin_statistic other than np.nanmedian / np.nanmean needs all values of a bin in one process: "
                                      "not available for partitioned (group=...) fits")
        if getattr(self, "_vals", None) is None:
            self._vals = (np.empty(self.shape, dtype=self.dtype), np.empty(self.shape, dtype=np.uint16))
        y, bins = self._vals
        edges = np.empty(n_bins + 1, dtype=np.float64)
        vshift, ymean, ystd = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        nv = ctypes.c_int64()
        dp = ctypes.POINTER(ctypes.c_double)
        rc = self.ctx._L.xdemhip_nk_step_values(self.handle, float(shift_x), float(shift_y), float(res[0]), float(res[1]), int(n_bins),
                                                ctypes.byref(vshift), ctypes.byref(nv), ctypes.byref(ymean), ctypes.byref(ystd),
                                                edges.ctypes.data_as(dp), y.ctypes.data, bins.ctypes.data, _lib.HOST)
        if rc != _lib.OK:
            self._raise_step_error(rc)
        ok = (bins != 0xFFFF) & np.isfinite(y)
        b, v = bins[ok], y[ok]
        counts = np.bincount(b, minlength=n_bins).astype(np.int64)
        # the callable per bin, SciPy's way (shared with nd_binning: xdem_amd/_binstat_host.py -- sample order within a bin, a fresh
        # array per bin, statistic([]) for empty bins, SciPy's own vectorised answers for np.sum / np.std / np.min / np.max)
        from ._binstat_host import binned_statistic_host

        stat = binned_statistic_host(self._bin_callable, b, v, n_bins)
        return {"vshift": vshift.value, "n_valid": int(nv.value), "y_mean": ymean.value, "y_std": ystd.value,
                "edges": edges, "counts": counts, "medians": stat}

    def _raise_step_error(self, rc: int) -> None:
        msg = self.ctx._L.xdemhip_last_error(self.ctx.handle).decode()
        if "no more valid values" in msg:
            raise ValueError(
                "The subsample contains no more valid values. This can happen is the horizontal shift to "
                "correct is very large, or if the algorithm diverged. To ensure all possible points can "
                "be used at any iteration step, use subsample=1."
            )
        if "halo too small" in msg:
            raise HaloTooSmall(msg)
        raise _lib.XdemHipError(f"libxdemhip status {rc}: {msg}")

    def step_fit(self, shift_x: float, shift_y: float, res: tuple[float, float]) -> dict[str, Any]:
        """One iteration without binning (``bin_before_fit=False``): vertical shift, valid count, the p0 ingredients and the
        ten least-squares sums of ``y = A cos x + B sin x + c`` over every valid point (``xdemhip_nk_step_fit``)."""
        sums = np.empty(10, dtype=np.float64)
        vshift, ymean, ystd = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        nv = ctypes.c_int64()
        dp = ctypes.POINTER(ctypes.c_double)
        rc = self.ctx._L.xdemhip_nk_step_fit(self.handle, float(shift_x), float(shift_y), float(res[0]), float(res[1]),
                                             ctypes.byref(vshift), ctypes.byref(nv), ctypes.byref(ymean), ctypes.byref(ystd),
                                             sums.ctypes.data_as(dp))
        if rc != _lib.OK:
            self._raise_step_error(rc)
        return {"vshift": vshift.value, "n_valid": int(nv.value), "y_mean": ymean.value, "y_std": ystd.value, "sums": sums}

    def route_counts(self) -> dict[str, int]:
        """Steps answered so far by the one-pass step and by the plain route (``xdemhip_nk_route_counts``); how many one-pass steps took
        predicted brackets (``xdemhip_nk_predict_counts``)."""
        a, c = ctypes.c_int64(), ctypes.c_int64()
        self.ctx.check(self.ctx._L.xdemhip_nk_route_counts(self.handle, ctypes.byref(a), ctypes.byref(c)))
        p, d, m = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self.ctx.check(self.ctx._L.xdemhip_nk_predict_counts(self.handle, ctypes.byref(p), ctypes.byref(d), ctypes.byref(m)))
        return {"onepass": int(a.value), "plain": int(c.value), "predicted": int(p.value),
                "predicted_dh_only": int(d.value), "predict_missed": int(m.value)}

    def set_bin_edges(self, edges) -> None:
        """Explicit aspect-bin edges (``bin_sizes={"aspect": edges}`` upstream); ``None`` restores SciPy's automatic edges."""
        dp = ctypes.POINTER(ctypes.c_double)
        if edges is None:
            self.ctx.check(self.ctx._L.xdemhip_nk_set_bin_edges(self.handle, None, 0, 0))
            self._n_custom_bins = None
            return
        e = np.ascontiguousarray(edges, dtype=np.float64).ravel()
        # SciPy's rightmost-edge rule rounds to `decimal` digits, from the smallest spacing of the dtype-cast edges
        decimal = int(-np.log10(np.diff(e.astype(self.dtype)).min())) + 6
        self.ctx.check(self.ctx._L.xdemhip_nk_set_bin_edges(self.handle, e.ctypes.data_as(dp), int(e.size), decimal))
        self._n_custom_bins = int(e.size) - 1

    def set_statistic(self, bin_statistic) -> None:
        """Statistic of the aspect bins: ``np.nanmedian`` / ``np.median`` (exact selection, the default) or ``np.nanmean`` /
        ``np.mean`` (per-bin sums and counts); ``step`` then returns the bin means under ``"medians"``."""
        sid = _bin_statistic_id(bin_statistic)
        self._bin_callable = bin_statistic if sid == 2 else None
        self.ctx.check(self.ctx._L.xdemhip_nk_set_statistic(self.handle, 0 if sid == 2 else sid))

    def aux(self):
        """(slope_tan, aspect, valid) copied back to the host (tests / debugging)."""
        st = np.empty(self.shape, dtype=self.dtype)
        asp = np.empty(self.shape, dtype=self.dtype)
        valid = np.empty(self.shape, dtype=np.uint8)
        self.ctx.check(self.ctx._L.xdemhip_nk_get_aux(self.handle, st.ctypes.data, asp.ctypes.data, valid.ctypes.data))
        return st, asp, valid.astype(bool)

    def subsample(self, ranks: np.ndarray) -> int:
        """Keep as inliers exactly the valid pixels whose rank -- position among the plan's valid pixels in raster order -- is in
        ``ranks`` (distinct, in [0, n_valid)): what ``flatnonzero(valid)[ranks]`` selects, formed on the device
        (``xdemhip_nk_subsample``).  Whole-raster plans of one process only.  Returns the new number of valid pixels."""
        ranks = np.ascontiguousarray(ranks, dtype=np.int64)
        nv = ctypes.c_int64()
        self.ctx.check(self.ctx._L.xdemhip_nk_subsample(self.handle, ranks.ctypes.data, int(ranks.size), _lib.HOST, ctypes.byref(nv)))
        self.n_valid = int(nv.value)
        return self.n_valid

    def close(self) -> None:
        if getattr(self, "handle", None):
            if getattr(self.ctx, "handle", None):   # (a context that is already gone took its plans with it)
                self.ctx._L.xdemhip_nk_destroy(self.handle)
                if getattr(self, "group", None) is not None:
                    self.ctx.set_allreduce(None)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def binned_median(x: np.ndarray, y: np.ndarray, n_bins: int = 72, ctx: _lib.Context | None = None):
    """``scipy.stats.binned_statistic(x, y, np.nanmedian, n_bins)`` + counts on the GPU (xdem/spatialstats.py:143-157).
    Returns (edges float64[n+1], counts int64[n], medians float64[n])."""
    x = np.ascontiguousarray(x).ravel()
    y = np.ascontiguousarray(y).ravel()
    dt = np.float64 if np.float64 in (x.dtype, y.dtype) else np.float32
    x, y = x.astype(dt, copy=False), y.astype(dt, copy=False)
    ctx = ctx or _lib.default_context()
    edges = np.empty(n_bins + 1, dtype=np.float64)
    counts = np.empty(n_bins, dtype=np.int64)
    med = np.empty(n_bins, dtype=np.float64)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)
    ctx.check(ctx._L.xdemhip_binned_median(ctx.handle, x.ctypes.data, y.ctypes.data, _lib.F32 if dt == np.float32 else _lib.F64,
                                           x.size, int(n_bins), edges.ctypes.data_as(dp), counts.ctypes.data_as(ip),
                                           med.ctypes.data_as(dp)))
    return edges, counts, med


def _bin_fit_from_step(det: dict[str, Any], fit_optimizer: Callable[..., Any], dtype) -> tuple[float, float, float]:
    """The host half of ``_nuth_kaab_bin_fit`` (affine.py:381-409, base.py:1022-1045): p0, bin mids, curve_fit."""
    p0 = (3 * det["y_std"] / (2**0.5), 0.0, det["y_mean"])
    edges = det["edges"]  # sample-dtype values held in float64, like pd.IntervalIndex.from_breaks stores them
    mids = 0.5 * (edges[:-1] + edges[1:])  # pd.IntervalIndex.mid (base.py:1027)
    med = det["medians"]
    ok = np.isfinite(med) & np.isfinite(mids)
    if np.all(~ok):
        raise ValueError("Only NaN values after binning, did you pass the right bin edges?")
    results = fit_optimizer(f=_nuth_kaab_fit_func, xdata=mids[ok], ydata=med[ok], p0=p0, sigma=None, absolute_sigma=True)
    a, b, c = results[0]
    return a * np.sin(b), a * np.cos(b), c


def _fit_from_sums(det: dict[str, Any]) -> tuple[float, float, float]:
    """``bin_before_fit=False``: the least-squares optimum of ``a cos(b - x) + c`` over all points, which is what upstream's
    ``curve_fit`` converges to from its p0 (affine.py:381-409, base.py:975-989), from the normal equations of the linear
    form ``y = A cos x + B sin x + c`` (A = a cos b = northing, B = a sin b = easting)."""
    n, sc, ss, scc, sss, scs, sy, syc, sys_, _ = det["sums"]
    m = np.array([[scc, scs, sc], [scs, sss, ss], [sc, ss, n]], dtype=np.float64)
    rhs = np.array([syc, sys_, sy], dtype=np.float64)
    # (least squares, not `solve`: a degenerate system -- every point at one aspect -- has a minimum-norm answer, as an
    # optimiser started from p0 would return some finite point instead of raising LinAlgError)
    a_, b_, c_ = np.linalg.lstsq(m, rhs, rcond=None)[0]
    return float(b_), float(a_), float(c_)


def _check_unbinned_optimizer(fit_optimizer, bin_before_fit: bool) -> None:
    """``bin_before_fit=False`` is solved in closed form from ten sums over the grid (the optimum plain
    ``scipy.optimize.curve_fit`` converges to); upstream would call ``fit_optimizer`` on every point (base.py:975-989), which
    a robust or bounded optimiser answers differently -- refuse those instead of silently ignoring them."""
    import scipy.optimize

    if not bin_before_fit and fit_optimizer is not None and fit_optimizer is not scipy.optimize.curve_fit:
        raise NotImplementedError(
            "bin_before_fit=False is solved from least-squares sums on the GPU, which equals scipy.optimize.curve_fit only; "
            "a custom fit_optimizer would need every point on the host. Use bin_before_fit=True with it."
        )


def subsample_valid_mask(valid_mask: np.ndarray, subsample: float | int, random_state=None) -> np.ndarray:
    """Boolean mask of a random subsample of the valid pixels (``_get_subsample_on_valid_mask``, xdem/coreg/base.py:577-617).
    The draw itself is geoutils' ``subsample_array`` (un-vendored, absent here); its published rule is restated --
    ``rng = default_rng(random_state)``, ``n = int(subsample * n_valid)`` for 0 < subsample <= 1 else ``int(subsample)``,
    capped at n_valid, ``rng.choice(flat valid indices, n, replace=False)`` -- **parity unpinned**."""
    n_valid = int(np.count_nonzero(valid_mask))
    if subsample == 1 and n_valid > 0:
        return valid_mask
    ranks = subsample_ranks(n_valid, subsample, random_state)
    valids = np.flatnonzero(valid_mask.ravel())
    out = np.zeros(valid_mask.size, dtype=bool)
    out[valids[ranks]] = True
    return out.reshape(valid_mask.shape)


def subsample_ranks(n_valid: int, subsample: float | int, random_state=None) -> np.ndarray:
    """The draw of ``subsample_valid_mask`` as RANKS among the valid pixels: ``rng.choice(valids, n, replace=False)`` is
    ``valids[rng.choice(len(valids), n, replace=False)]`` -- NumPy draws the positions from the population SIZE and indexes the array
    with them (same generator stream, same order; CPU test) -- so the ranks need only the count of valid pixels, not the mask:
    the device turns them into pixels (``NKPlan.subsample``) and the mask stays where it is."""
    if n_valid == 0:
        raise ValueError(
            "There is no valid points common to the input and auxiliary data (bias variables, or "
            "derivatives required for this method, for example slope, aspect, etc)."
        )
    if subsample <= 0:
        raise ValueError("`subsample` must be > 0")
    npoints = int(subsample * n_valid) if subsample <= 1 else int(subsample)
    npoints = min(npoints, n_valid)
    rng = np.random.default_rng(random_state)
    return rng.choice(n_valid, npoints, replace=False)


def _iterate(plan: "NKPlan", res, tolerance, max_iterations, bin_sizes, fit_optimizer, bin_before_fit: bool, initial_offsets=(0.0, 0.0)):
    """``_iterate_method`` (affine.py:102-147) around one plan: stop when i > 1 and the horizontal step falls below the tolerance."""
    offsets = (float(initial_offsets[0]), float(initial_offsets[1]), 0.0)
    for i in range(max_iterations):
        if bin_before_fit:
            det = plan.step(offsets[0], offsets[1], res, bin_sizes if isinstance(bin_sizes, (int, np.integer)) else 72)
            east, north, _ = _bin_fit_from_step(det, fit_optimizer, plan.dtype)
        else:
            det = plan.step_fit(offsets[0], offsets[1], res)
            east, north, _ = _fit_from_sums(det)
        offsets = (offsets[0] + east * res[0], offsets[1] + north * res[1], float(det["vshift"]))
        stat = float(np.sqrt(east**2 + north**2))
        if logging.getLogger().getEffectiveLevel() <= logging.INFO:
            logging.info("      Iteration #%d - Offset: %s; Magnitude: %s", i + 1, offsets, stat)
        if i > 1 and stat < tolerance:
            logging.info("   Last offset was below the residual offset threshold of %s -> stopping", tolerance)
            break
    return offsets


_HOST_DRAW = False   # tests: True sends nuth_kaab's random subsample through the host-mask form (the only form before round 6's end)


def _shared_seed(random_state, group):
    """One seed for every rank of `group` (rank 0's choice), so that all ranks draw the same random subsample."""
    import torch.distributed as dist

    pg = None if group == "world" else group
    box = [random_state if random_state is not None else int(np.random.SeedSequence().entropy % (2**63))]
    dist.broadcast_object_list(box, src=dist.get_global_rank(pg, 0) if pg is not None else 0, group=pg)
    return box[0]


def nuth_kaab(ref_elev: np.ndarray, tba_elev: np.ndarray, inlier_mask: np.ndarray | None, res: tuple[float, float],
              tolerance: float = 0.001, max_iterations: int = 10, bin_sizes=72,
              fit_optimizer: Callable[..., Any] | None = None, ctx: _lib.Context | None = None, group=None,
              subsample: float | int = 1, random_state=None, bin_statistic=np.nanmedian, bin_before_fit: bool = True,
              initial_offsets: tuple[float, float] = (0.0, 0.0)):
    """Array-level entry mirroring ``nuth_kaab`` (xdem/coreg/affine.py:539-609) for two rasters.
    ``subsample != 1`` restricts every iteration to a random subset of the valid pixels (drawn once, affine.py:581-593; with
    ``group`` every rank draws with rank 0's seed); ``group`` (torch.distributed process group or "world") shards every grid
    pass over the ranks by row block; ``bin_sizes``: number of aspect bins or an array of bin edges; ``bin_before_fit=False``
    fits all points instead of the bin medians; ``initial_offsets`` = (easting, northing) the iteration starts from.

    Returns ((easting, northing, vertical) offsets in georeferenced units, subsample_final)."""
    import scipy.optimize

    _check_unbinned_optimizer(fit_optimizer, bin_before_fit)
    fit_optimizer = fit_optimizer or scipy.optimize.curve_fit
    logging.info("Running Nuth and Kääb (2011) coregistration")
    plan = NKPlan(ref_elev, tba_elev, inlier_mask, ctx, group)
    if subsample != 1 and plan.n_valid > 0 and group is None and not _HOST_DRAW:
        # valid = inlier & finite ref / tba / slope / aspect (base.py:650-661), as the aux pass just established it; the draw needs its
        # COUNT only (subsample_ranks), the plan turns the drawn ranks into pixels on the device: no mask back to the host, no second
        # plan (20000^2 host arrays, default subsample: 0.95 s -> profiles/r06_nk_end_to_end.txt)
        plan.subsample(subsample_ranks(plan.n_valid, subsample, random_state))
    elif subsample != 1 and plan.n_valid > 0:
        # partitioned fits (and the test switch _HOST_DRAW): the mask travels, the draw indexes it on the host, a second plan takes it
        valid = plan.aux()[2]
        plan.close()
        if group is not None:
            random_state = _shared_seed(random_state, group)
        plan = NKPlan(ref_elev, tba_elev, subsample_valid_mask(valid, subsample, random_state), ctx, group)
    try:
        if plan.n_valid == 0:
            raise ValueError(
                "There is no valid points common to the input and auxiliary data (bias variables, or "
                "derivatives required for this method, for example slope, aspect, etc)."
            )
        plan.set_statistic(bin_statistic)
        if not isinstance(bin_sizes, (int, np.integer)):
            plan.set_bin_edges(bin_sizes)
        offsets = _iterate(plan, res, tolerance, max_iterations, bin_sizes, fit_optimizer, bin_before_fit, initial_offsets)
        return offsets, plan.n_valid
    finally:
        plan.close()


def apply_translation(elev: np.ndarray, shift_x: float, shift_y: float, shift_z: float, resolution, resample: bool = True,
                      ctx: _lib.Context | None = None) -> np.ndarray:
    """Apply a pure translation to a DEM array (``Coreg.apply`` for ``shift_x / shift_y / shift_z``): with
    ``resample=True`` the shifted DEM is bilinearly resampled onto its original grid
    (``_apply_matrix_rst`` case 2 + ``_reproject_horizontal_shift_samecrs``, xdem/coreg/base.py:1522-1570, 1615-1655):
    ``out(r, c) = elev(r + shift_y / res_y, c - shift_x / res_x) + shift_z``.  Without resampling only ``shift_z`` is
    added (the reference then just moves the geotransform)."""
    arr = np.ascontiguousarray(elev.filled(np.nan) if isinstance(elev, np.ma.MaskedArray) else elev)
    if arr.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        arr = arr.astype(np.float32)
    from .spatialstats import _count_finite   # (np.isfinite over the raster on the library's host threads: 7 ms against 0.1 s at 20000^2)

    if _count_finite(arr)[0] == 0:
        raise ValueError("Input DEM has all nans.")
    if not resample:
        return arr + arr.dtype.type(shift_z)
    res = (float(resolution), float(resolution)) if np.isscalar(resolution) else (float(resolution[0]), float(resolution[1]))
    ctx = ctx or _lib.default_context()
    out = np.empty_like(arr)
    ctx.check(ctx._L.xdemhip_shift_bilinear(ctx.handle, arr.ctypes.data, _lib.F32 if arr.dtype == np.float32 else _lib.F64,
                                            arr.shape[0], arr.shape[1], float(shift_y) / res[1], -float(shift_x) / res[0],
                                            float(shift_z), out.ctypes.data, _lib.HOST))
    return out


class NuthKaab:
    """Nuth and Kaab (2011) coregistration: horizontal and vertical translations by iterative slope/aspect alignment.

    Constructor mirrors ``xdem.coreg.NuthKaab.__init__`` (xdem/coreg/affine.py:2397-2456).  ``fit`` takes the two DEMs
    as arrays on the same grid plus ``resolution`` (the reference gets it from the raster transform); estimated shifts
    land in ``self.meta["outputs"]["affine"]`` as ``shift_x = -easting``, ``shift_y = -northing``,
    ``shift_z = vertical * vertical_shift`` (affine.py:2526-2530).
    """

    def __init__(self, max_iterations: int = 10, offset_threshold: float = 0.001, bin_before_fit: bool = True,
                 fit_optimizer: Callable[..., Any] | None = None, bin_sizes: int = 72,
                 bin_statistic: Callable[[np.ndarray], Any] = np.nanmedian, subsample: int | float = 5e5,
                 vertical_shift: bool = True, initial_shift=None) -> None:
        import scipy.optimize

        _bin_statistic_id(bin_statistic)  # (np.nanmedian / np.nanmean on the GPU, any other callable over the GPU's y values; not a callable: TypeError)
        _check_unbinned_optimizer(fit_optimizer, bin_before_fit)
        if isinstance(bin_sizes, dict):  # upstream's {"aspect": n | edges} form (base.py:957-966)
            if list(bin_sizes) != ["aspect"]:
                raise ValueError("The keys of `bin_sizes` must be ['aspect'] for NuthKaab.")
            bin_sizes = bin_sizes["aspect"]
        if not isinstance(bin_sizes, (int, np.integer)):
            bin_sizes = np.asarray(bin_sizes, dtype=np.float64)
            if bin_sizes.ndim != 1 or bin_sizes.size < 2 or np.any(np.diff(bin_sizes) <= 0):
                raise ValueError("bin_sizes must be a number of bins or a 1-D array of increasing bin edges.")
            if bin_sizes.size > 129:
                raise NotImplementedError("explicit aspect-bin edges: at most 128 bins (one histogram sweep on the GPU); pass a "
                                          "number of bins (up to 1024) for a finer uniform binning.")
        if initial_shift is not None:
            # same checks as AffineCoreg.__init__ (affine.py:1813-1829)
            if not (isinstance(initial_shift, tuple) and len(initial_shift) in (2, 3)
                    and all(isinstance(val, (float, int)) for val in initial_shift)):
                raise ValueError("Argument `initial_shift` must be a tuple of exactly two or three numerical values.")
            if len(initial_shift) == 2:
                initial_shift += (0,)
            elif initial_shift[2] != 0:
                import warnings

                initial_shift = (*initial_shift[:2], 0)
                warnings.warn("Initial shift in altitude is currently work in progress.", category=UserWarning)
        self.vertical_shift = vertical_shift
        self.meta: dict[str, Any] = {
            "inputs": {
                "iterative": {"max_iterations": max_iterations, "tolerance": offset_threshold},
                "fitorbin": {"fit_or_bin": "bin_and_fit" if bin_before_fit else "fit", "fit_func": _nuth_kaab_fit_func,
                             "fit_optimizer": fit_optimizer or scipy.optimize.curve_fit,
                             "bin_sizes": int(bin_sizes) if isinstance(bin_sizes, (int, np.integer)) else bin_sizes,
                             "bin_statistic": bin_statistic},
                "random": {"subsample": subsample, "random_state": None},
                "affine": {"apply_vshift": vertical_shift, **({"initial_shift": initial_shift} if initial_shift is not None else {})},
            },
            "outputs": {},
        }

    @staticmethod
    def _resolution(resolution, transform) -> tuple[float, float]:
        """(x, y) pixel size from ``resolution`` (scalar or pair) or from an affine ``transform`` (object with ``.a`` / ``.e`` or
        a 6-tuple ``(a, b, c, d, e, f)``), which is how the reference's array interface carries it."""
        if resolution is None and transform is not None:
            a, e = (transform.a, transform.e) if hasattr(transform, "a") else (transform[0], transform[4])
            return (abs(float(a)), abs(float(e)))
        if resolution is None:
            raise ValueError("'transform' must be given if both DEMs are array-like.")  # (base.py:204; or pass resolution=)
        return (float(resolution), float(resolution)) if np.isscalar(resolution) else (float(resolution[0]), float(resolution[1]))

    def fit(self, reference_elev: np.ndarray, to_be_aligned_elev: np.ndarray, inlier_mask: np.ndarray | None = None,
            bias_vars=None, weights=None, subsample: int | float | None = None, transform=None, crs=None, area_or_point=None,
            z_name: str | None = None, random_state=None, resolution: float | tuple[float, float] | None = None, **kwargs: Any) -> "NuthKaab":
        """Estimate the x/y/z offset between two DEMs given as arrays on the same grid (``Coreg.fit``, base.py:2250-2368:
        same parameters in the same order; the grid spacing comes from ``transform`` as upstream or from ``resolution``).
        ``bias_vars`` / ``weights`` belong to other methods and must stay ``None``; ``crs`` / ``area_or_point`` / ``z_name``
        are accepted for call compatibility (one shared grid is assumed, georeferencing stays with the caller)."""
        if bias_vars is not None or weights is not None:
            raise NotImplementedError("bias_vars / weights are not used by NuthKaab.")
        if subsample is not None:
            self.meta["inputs"]["random"]["subsample"] = subsample
        if random_state is not None:
            self.meta["inputs"]["random"]["random_state"] = random_state
        res = self._resolution(resolution, transform)
        ref = np.asarray(reference_elev.filled(np.nan) if isinstance(reference_elev, np.ma.MaskedArray) else reference_elev)
        tba = np.asarray(to_be_aligned_elev.filled(np.nan) if isinstance(to_be_aligned_elev, np.ma.MaskedArray) else to_be_aligned_elev)
        it = self.meta["inputs"]["iterative"]
        fb = self.meta["inputs"]["fitorbin"]
        # initial_shift (base.py:2307-2313, 2358-2366): upstream translates the reference by (-sx, -sy), fits, and adds (sx, sy)
        # back to the estimated shift -- i.e. the search starts from shift_x = sx.  Here the iteration itself starts from
        # the offsets (-sx, -sy), which samples the to-be-aligned DEM once at the shifted position instead of resampling
        # it onto the translated grid first.
        init = self.meta["inputs"]["affine"].get("initial_shift")
        (east, north, vert), n_final = nuth_kaab(ref, tba, inlier_mask, res, tolerance=it["tolerance"],
                                                 max_iterations=it["max_iterations"], bin_sizes=fb["bin_sizes"],
                                                 fit_optimizer=fb["fit_optimizer"],
                                                 subsample=self.meta["inputs"]["random"]["subsample"],
                                                 random_state=self.meta["inputs"]["random"]["random_state"],
                                                 bin_statistic=fb["bin_statistic"], bin_before_fit=fb["fit_or_bin"] == "bin_and_fit",
                                                 initial_offsets=(-init[0], -init[1]) if init is not None else (0.0, 0.0))
        self.meta["outputs"]["affine"] = {"shift_x": -east, "shift_y": -north, "shift_z": vert * self.vertical_shift}
        self.meta["outputs"]["random"] = {"subsample_final": n_final}
        return self

    def apply(self, elev: np.ndarray, resolution: float | tuple[float, float] | None = None, resample: bool = True, *,
              bias_vars=None, resampling: str = "bilinear", transform=None, crs=None, z_name: str = "z"):
        """Apply the estimated translation to a DEM array on the fit grid (``Coreg.apply``, translation case; upstream's
        keywords are keyword-only here).  With ``transform=`` the call returns ``(array, transform)`` like upstream's array
        interface -- for ``resample=False`` the shift goes into the returned transform and only the vertical shift into the
        array; with ``resolution=`` it returns the array alone."""
        if bias_vars is not None:
            raise NotImplementedError("bias_vars is not used by NuthKaab.")
        if resampling != "bilinear":
            raise NotImplementedError("the GPU resampler is bilinear (the reference default).")
        a = self.meta["outputs"]["affine"]
        res = self._resolution(resolution, transform)
        out = apply_translation(elev, a["shift_x"], a["shift_y"], a["shift_z"], res, resample)
        if transform is None:
            return out
        if resample:
            return out, transform
        t = (transform.a, transform.b, transform.c, transform.d, transform.e, transform.f) if hasattr(transform, "a") else tuple(transform)
        shifted = (t[0], t[1], t[2] + a["shift_x"], t[3], t[4], t[5] + a["shift_y"])
        return out, (type(transform)(*shifted) if hasattr(transform, "a") else shifted)

    def fit_and_apply(self, reference_elev, to_be_aligned_elev, inlier_mask=None, bias_vars=None, weights=None, subsample=None,
                      transform=None, crs=None, area_or_point=None, z_name: str = "z", resample: bool = True,
                      resampling: str = "bilinear", random_state=None, fit_kwargs=None, apply_kwargs=None):
        """``Coreg.fit_and_apply`` (base.py:2482-2590): fit, then apply to the to-be-aligned elevations."""
        fit_kwargs = dict(fit_kwargs or {})
        apply_kwargs = dict(apply_kwargs or {})
        self.fit(reference_elev, to_be_aligned_elev, inlier_mask=inlier_mask, bias_vars=bias_vars, weights=weights, subsample=subsample,
                 transform=transform, crs=crs, area_or_point=area_or_point, z_name=z_name, random_state=random_state, **fit_kwargs)
        resolution = apply_kwargs.pop("resolution", fit_kwargs.get("resolution"))
        return self.apply(to_be_aligned_elev, resolution, resample, bias_vars=bias_vars, resampling=resampling, transform=transform,
                          crs=crs, z_name=z_name, **apply_kwargs)

    def copy(self) -> "NuthKaab":
        """Identical, independent copy (base.py:1999-2006)."""
        import copy as _copy

        new = self.__new__(type(self))
        new.__dict__ = {k: _copy.deepcopy(v) for k, v in self.__dict__.items()}
        return new

    @property
    def is_affine(self) -> bool:
        return True

    def to_translations(self) -> tuple[float, float, float]:
        """(x, y, z) translations of the estimated transform (base.py: ``to_translations`` of affine methods)."""
        m = self.to_matrix()
        return (float(m[0, 3]), float(m[1, 3]), float(m[2, 3]))

    def to_rotations(self) -> tuple[float, float, float]:
        """Rotations of the estimated transform: none, Nuth and Kaab is a pure translation."""
        return (0.0, 0.0, 0.0)

    def to_matrix(self) -> np.ndarray:
        """4x4 translation matrix (affine.py:2532-2541)."""
        m = np.diag(np.ones(4, dtype=float))
        m[0, 3] += self.meta["outputs"]["affine"]["shift_x"]
        m[1, 3] += self.meta["outputs"]["affine"]["shift_y"]
        m[2, 3] += self.meta["outputs"]["affine"]["shift_z"]
        return m
