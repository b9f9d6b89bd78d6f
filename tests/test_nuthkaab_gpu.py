"""GPU parity tests of the Nuth-Kaab path: HIP kernels (through the C-ABI) vs the CPU oracle and the vectors recorded
from the reference's own functions.  Integer work (valid counts, bin membership counts) and selections (medians) are
bit-exact; fitted shifts within 1e-6 relative."""
import os

import numpy as np
import pytest

import nuthkaab_oracle as nko
from conftest import GOLDEN, decided, default_conventions

pytestmark = pytest.mark.gpu


def _moments_close(a, b, onepass=True):
    """nanmean / nanstd of y.  They only seed the 72-point curve fit (p0, xdem/coreg/affine.py:386) and the reference itself
    forms them in float32 (np.nanmean / np.nanstd of a float32 array: pairwise float32 sums, ~1e-7 of the spread).  The plain
    route accumulates float64 atomics (last bits depend on the order of the additions); the one-pass step of round 4 (option
    "nk_fused") sums y^ = (dh - v^) / slope_tan in float32 pieces with a first-order correction in (v^ - vshift): 2e-6 of
    the spread, the reference's own accuracy class."""
    tol = 2e-6 * abs(b["y_std"]) if onepass else 1e-10 * abs(b["y_std"]) + 1e-11 * abs(b["y_mean"]) + 1e-13
    return abs(a["y_mean"] - b["y_mean"]) <= tol and abs(a["y_std"] - b["y_std"]) <= tol



@pytest.fixture(scope="module")
def coreg():
    from xdem_amd import coreg as c

    return c


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(GOLDEN, "nk_golden.npz"))


def _pair(shape=(200, 260), seed=42, dtype=np.float32, res=10.0):
    from xdem_amd.synth import fbm_numpy

    ref = fbm_numpy(shape, seed=seed, std=150.0, dtype=dtype)
    tba = ref - nko.shifted_dh(ref, ref, 1.7 * res, -0.6 * res, (res, res))
    tba = (tba + 2.0 + np.random.default_rng(seed + 1).normal(scale=0.05, size=shape)).astype(dtype)
    hole = fbm_numpy(shape, seed=seed + 2, hurst=1.0, mean=0.0, std=1.0)
    tba[hole < np.percentile(hole, 20)] = np.nan
    inlier = np.ones(shape, dtype=bool)
    inlier[:4] = False
    ref = ref.copy()
    ref[50:54, 60:70] = 321.0  # flat patch: zero slope -> excluded
    ref[100, 100] = np.nan
    return ref, tba, inlier, res


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_aux_vars_bit_exact(coreg, dtype):
    ref, tba, inlier, res = _pair(dtype=dtype)
    plan = coreg.NKPlan(ref, tba, inlier)
    st, asp, valid = plan.aux()
    st_o, asp_o = nko.aux_vars(ref)
    valid_o = inlier & np.isfinite(ref) & np.isfinite(tba) & np.isfinite(st_o) & np.isfinite(asp_o)
    assert np.array_equal(st, st_o, equal_nan=True)
    if dtype == np.float32:
        assert np.array_equal(asp, asp_o, equal_nan=True)
    else:  # float64 arctangent: libm vs ocml may differ in the last bit
        assert np.allclose(asp, asp_o, rtol=1e-15, atol=1e-15, equal_nan=True)
    assert np.array_equal(valid, valid_o) and plan.n_valid == int(valid_o.sum())
    plan.close()


def test_aux_vs_reference_fixture(coreg, z):
    dem = z["T8|float32|dem"]
    plan = coreg.NKPlan(dem, dem, None)
    st, asp, _ = plan.aux()
    assert np.array_equal(st, z["T8|float32|slope_tan"], equal_nan=True)
    ref = z["T8|float32|aspect"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isnan(asp), np.isnan(ref))
    assert np.max(np.abs(asp[fin].astype(np.float64) - ref[fin])) <= 2 * np.spacing(np.float32(6.3))
    plan.close()


@pytest.mark.parametrize("shift", [(0.0, 0.0), (17.3, -5.1), (-3.25, 40.0)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_step_matches_oracle(coreg, shift, dtype):
    ref, tba, inlier, res = _pair(dtype=dtype)
    plan = coreg.NKPlan(ref, tba, inlier)
    det = plan.step(shift[0], shift[1], (res, res), 72)
    st, asp = nko.aux_vars(ref)
    valid = inlier & np.isfinite(ref) & np.isfinite(tba) & np.isfinite(st) & np.isfinite(asp)
    dh = nko.shifted_dh(ref, tba, shift[0], shift[1], (res, res))[valid]
    vshift = np.nanmedian(dh)
    assert det["vshift"] == float(vshift)  # selection: bit-exact
    dh = dh - vshift
    ok = np.isfinite(dh)
    assert det["n_valid"] == int(ok.sum())
    with np.errstate(all="ignore"):
        y = dh[ok] / st[valid][ok]
    a = asp[valid][ok]
    if dtype == np.float64:  # device arctangent may differ in the last bit: compare on the device's own aspect
        a = plan.aux()[1][valid][ok]
    edges, counts, med = nko.bin_medians(a, y, 72)
    assert np.array_equal(det["counts"], counts)            # integer work: bit-exact
    assert np.array_equal(det["edges"], edges.astype(np.float64))
    assert np.array_equal(det["medians"], med, equal_nan=True)  # selections: bit-exact
    assert np.isclose(det["y_mean"], np.nanmean(y.astype(np.float64)), rtol=1e-9, atol=1e-12)
    assert np.isclose(det["y_std"], np.nanstd(y.astype(np.float64)), rtol=1e-7)
    plan.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_step_mean_statistic_matches_oracle(coreg, dtype):
    """NuthKaab(bin_statistic=np.nanmean): counts and edges bit-exact; the bin means (float64 sums on the device, NumPy's
    pairwise float32 / float64 mean upstream) within 1e-6 relative -- the float bar of the task, stated here."""
    ref, tba, inlier, res = _pair(dtype=dtype)
    plan = coreg.NKPlan(ref, tba, inlier)
    plan.set_statistic(np.nanmean)
    det = plan.step(17.3, -5.1, (res, res), 72)
    aspect_dev = plan.aux()[1]
    st, asp = nko.aux_vars(ref)
    valid = inlier & np.isfinite(ref) & np.isfinite(tba) & np.isfinite(st) & np.isfinite(asp)
    dh = nko.shifted_dh(ref, tba, 17.3, -5.1, (res, res))[valid]
    vshift = np.nanmedian(dh)
    assert det["vshift"] == float(vshift)
    dh = dh - vshift
    ok = np.isfinite(dh)
    with np.errstate(all="ignore"):
        y = dh[ok] / st[valid][ok]
    a = asp[valid][ok] if dtype == np.float32 else aspect_dev[valid][ok]
    edges, counts, means = nko.bin_means(a, y, 72)
    assert np.array_equal(det["counts"], counts) and np.array_equal(det["edges"], edges.astype(np.float64))
    assert np.array_equal(np.isnan(det["medians"]), np.isnan(means))
    fin = np.isfinite(means)
    np.testing.assert_allclose(det["medians"][fin], means[fin], rtol=1e-6, atol=1e-6 * float(np.nanstd(y)))
    # back to the median: same plan, exact again
    plan.set_statistic(np.nanmedian)
    det2 = plan.step(17.3, -5.1, (res, res), 72)
    assert np.array_equal(det2["medians"], nko.bin_medians(a, y, 72)[2], equal_nan=True)
    plan.close()
    with pytest.raises(TypeError):
        coreg.NuthKaab(bin_statistic=0.5)   # (not a callable)


def _p75(v):
    return np.percentile(v, 75) if len(v) else np.nan


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_step_callable_statistic_equals_scipy_binned_statistic(coreg, dtype):
    """NuthKaab(bin_statistic=<any callable>) (affine.py:2404; nd_binning hands it to scipy.stats.binned_statistic,
    spatialstats.py:143-157): the GPU returns y and the bin id of every pixel (xdemhip_nk_step_values), the callable runs on the host
    over each bin's values in raster order.  Against SciPy's own binned_statistic on the oracle's y: bit for bit, empty bins included
    (SciPy fills them with statistic([]), or NaN if that raises), for an order statistic, a sum, a spread and a lambda -- and the
    median / mean through the same door equal the GPU's own exact medians / the NumPy means."""
    import scipy.stats

    ref, tba, inlier, res = _pair(dtype=dtype)
    plan = coreg.NKPlan(ref, tba, inlier)
    aspect_dev = plan.aux()[1]
    st, asp = nko.aux_vars(ref)
    valid = inlier & np.isfinite(ref) & np.isfinite(tba) & np.isfinite(st) & np.isfinite(asp)
    dh = nko.shifted_dh(ref, tba, 17.3, -5.1, (res, res))[valid]
    vshift = np.nanmedian(dh)
    dh = dh - vshift
    ok = np.isfinite(dh)
    with np.errstate(all="ignore"):
        y = dh[ok] / st[valid][ok]
    a = asp[valid][ok] if dtype == np.float32 else aspect_dev[valid][ok]
    plan.set_statistic(np.nanmedian)
    exact = plan.step(17.3, -5.1, (res, res), 72)
    for nb in (72, 500):
        edges = nko.bin_medians(a, y, nb)[0]
        for f in (np.nanmax, np.sum, np.std, np.min, np.max, np.nanstd, np.nansum, _p75, lambda v: float(np.median(v)) if len(v) else np.nan, lambda v: float(np.mean(v, dtype=np.float64)) if len(v) else np.nan):
            plan.set_statistic(f)
            det = plan.step(17.3, -5.1, (res, res), nb)
            want = scipy.stats.binned_statistic(a, y, statistic=f, bins=edges).statistic
            cnt = scipy.stats.binned_statistic(a, y, statistic="count", bins=edges).statistic
            assert det["vshift"] == float(vshift) and det["n_valid"] == int(ok.sum())
            assert np.array_equal(det["edges"], edges.astype(np.float64)) and np.array_equal(det["counts"], cnt.astype(np.int64))
            assert np.array_equal(det["medians"], np.asarray(want, dtype=np.float64), equal_nan=True), (nb, f)
            if nb == 72 and f.__name__ == "<lambda>" and "median" in f.__code__.co_names:
                assert np.array_equal(det["medians"], exact["medians"], equal_nan=True)
    assert plan.route_counts()["plain"] >= 18
    # the C-ABI entry with DEVICE buffers (what a caller holding torch tensors passes) returns the same arrays as with host buffers
    import ctypes

    import torch

    L, dp = plan.ctx._L, ctypes.POINTER(ctypes.c_double)
    outs = {}
    for space in ("host", "device"):
        if space == "host":
            yb, bb = np.empty(ref.shape, dtype=dtype), np.empty(ref.shape, dtype=np.uint16)
            yp, bp = yb.ctypes.data, bb.ctypes.data
        else:
            yb = torch.empty(ref.shape, dtype=torch.float32 if dtype == np.float32 else torch.float64, device="cuda")
            bb = torch.empty(ref.shape, dtype=torch.int16, device="cuda")
            yp, bp = yb.data_ptr(), bb.data_ptr()
        e = np.empty(73)
        v = [ctypes.c_double() for _ in range(3)]
        nv = ctypes.c_int64()
        rc = L.xdemhip_nk_step_values(plan.handle, 17.3, -5.1, res, res, 72, ctypes.byref(v[0]), ctypes.byref(nv), ctypes.byref(v[1]), ctypes.byref(v[2]),
                                      e.ctypes.data_as(dp), yp, bp, coreg._lib.HOST if space == "host" else coreg._lib.DEVICE)
        assert rc == 0
        if space == "device":
            torch.cuda.synchronize()
            yb, bb = yb.cpu().numpy(), bb.cpu().numpy().view(np.uint16)
        outs[space] = (yb, bb, e.copy(), v[0].value, nv.value)
    assert np.array_equal(outs["host"][0], outs["device"][0], equal_nan=True) and np.array_equal(outs["host"][1], outs["device"][1])
    assert np.array_equal(outs["host"][2], outs["device"][2]) and outs["host"][3:] == outs["device"][3:] and outs["host"][3] == float(vshift)
    # (NaN / 0xFFFF exactly where the pixel has no dh; every other pixel carries the oracle's y)
    has = np.isfinite(outs["host"][0])
    assert int(has.sum()) == int(ok.sum()) and np.array_equal(outs["host"][0][has], y) and np.all(outs["host"][1][~has] == 0xFFFF)
    # explicit edges that reach beyond [0, 2 pi): the outer bins stay empty -> statistic([]) where it exists (np.sum: 0), NaN otherwise
    if dtype == np.float32:
        edges = np.linspace(-1.5, 7.5, 19)
        plan.set_bin_edges(edges)
        for f in (np.sum, np.nanmax, _p75):
            plan.set_statistic(f)
            det = plan.step(17.3, -5.1, (res, res))
            want = scipy.stats.binned_statistic(a, y, statistic=f, bins=edges.astype(np.float32)).statistic
            cnt = scipy.stats.binned_statistic(a, y, statistic="count", bins=edges.astype(np.float32)).statistic
            assert (cnt == 0).any() and np.array_equal(det["counts"], cnt.astype(np.int64))
            assert np.array_equal(det["medians"], np.asarray(want, dtype=np.float64), equal_nan=True), f
        assert plan.step(17.3, -5.1, (res, res))["medians"][0] != plan.step(17.3, -5.1, (res, res))["medians"][0]   # (NaN: _p75 of an empty bin)
    plan.close()


def test_class_api_callable_statistic(coreg):
    """The class with a callable statistic: a trimmed mean recovers the construction like the median does."""
    from xdem_amd.synth import fbm_numpy

    def trimmed(v):
        lo, hi = np.percentile(v, [10, 90])
        return float(np.mean(v[(v >= lo) & (v <= hi)]))

    n, res = 600, 10.0
    ref = fbm_numpy((n, n), seed=8, std=150.0)
    tba = (np.roll(ref, (2, -1), (0, 1)) + 1.25).astype(np.float32)
    nk = coreg.NuthKaab(bin_statistic=trimmed, subsample=1, offset_threshold=0.0, max_iterations=8)
    nk.fit(ref, tba, None, resolution=res)
    a = nk.meta["outputs"]["affine"]
    med = coreg.NuthKaab(subsample=1, offset_threshold=0.0, max_iterations=8).fit(ref, tba, None, resolution=res).meta["outputs"]["affine"]
    for k in ("shift_x", "shift_y", "shift_z"):
        assert abs(a[k] - med[k]) < 0.05 * res, (k, a[k], med[k])


def test_class_api_mean_statistic_recovers_shift(coreg):
    from xdem_amd.synth import fbm_numpy

    n, res = 600, 10.0
    ref = fbm_numpy((n, n), seed=8, std=150.0)
    tba = (np.roll(ref, (2, -1), (0, 1)) + 1.25).astype(np.float32)
    nk = coreg.NuthKaab(bin_statistic=np.nanmean, subsample=1, offset_threshold=0.0, max_iterations=8)
    nk.fit(ref, tba, None, resolution=res)
    a = nk.meta["outputs"]["affine"]
    # tba(x) = ref(x + (1, -2) px) + 1.25: shift_x = -easting etc. (same construction as test_class_api_recovers_shift)
    med = coreg.NuthKaab(subsample=1, offset_threshold=0.0, max_iterations=8).fit(ref, tba, None, resolution=res).meta["outputs"]["affine"]
    for k in ("shift_x", "shift_y", "shift_z"):
        assert abs(a[k] - med[k]) < 0.05 * res, (k, a[k], med[k])


@pytest.mark.parametrize("n", [1000, 200000, 5001])
def test_binned_median_vs_reference_fixture(coreg, z, n):
    k = f"T5|{n}"
    aspect, slope_tan, dh = z[k + "|aspect"], z[k + "|slope_tan"], z[k + "|dh"]
    y = dh / slope_tan
    edges, counts, med = coreg.binned_median(aspect, y, 72)
    assert np.array_equal(counts, z[k + "|count"])
    assert np.array_equal(edges[:-1], z[k + "|left"]) and np.array_equal(edges[1:], z[k + "|right"])
    assert np.array_equal(med, z[k + "|nanmedian"], equal_nan=True)


def test_binned_median_edge_cases(coreg):
    rng = np.random.default_rng(0)
    # empty bins, duplicates, NaNs, even / odd counts, a single point
    x = np.concatenate([rng.uniform(0, 1, 500), rng.uniform(5, 6, 501)]).astype(np.float32)
    y = rng.normal(size=x.size).astype(np.float32)
    y[::7] = y[0]
    y[3] = np.nan
    x[11] = np.nan
    e, c, m = coreg.binned_median(x, y, 72)
    ok = np.isfinite(x) & np.isfinite(y)
    e0, c0, m0 = nko.bin_medians(x[ok], y[ok], 72)
    assert np.array_equal(c, c0) and np.array_equal(m, m0, equal_nan=True) and np.array_equal(e, e0.astype(np.float64))
    assert (c == 0).any() and np.isnan(m[c == 0]).all()
    e, c, m = coreg.binned_median(np.array([2.0], np.float32), np.array([7.0], np.float32), 4)
    assert c.sum() == 1 and np.nanmax(m) == 7.0
    for nb in (1, 150, 300):  # more bins than one LDS sweep holds
        e, c, m = coreg.binned_median(x, y, nb)
        e0, c0, m0 = nko.bin_medians(x[ok], y[ok], nb)
        assert np.array_equal(c, c0) and np.array_equal(m, m0, equal_nan=True)


def test_full_fit_vs_oracle_and_reference(coreg, z):
    """T9: the reference's own `nuth_kaab` loop around a stand-in interpolator, recorded for each of the four nodata conventions
    (round 6: oracle/gen_golden_nk.py) -- the fixture of the DECIDED rule is the one compared (rule 0 keeps the first round's keys)."""
    from conftest import decided

    rule = decided("nk_nan_rule")
    pre = "T9|" if rule == 0 else f"T9|rule{rule}|"
    ref, tba, inlier, res = z["T9|ref"], z["T9|tba"], z["T9|inlier"], float(z["T9|res"])
    for tol in ("0.0", "0.001"):
        offsets, n_final = coreg.nuth_kaab(ref, tba, inlier, (res, res), tolerance=float(tol), max_iterations=10)
        o_off, o_n, _ = nko.nuth_kaab(ref, tba, inlier, (res, res), tolerance=float(tol), max_iterations=10)
        assert n_final == o_n == int(z[f"{pre}{tol}|subsample_final"])
        # Every grid quantity of a step is bit-exact (test_step_matches_oracle).  The final offsets additionally go
        # through scipy's Levenberg-Marquardt on 72 points, which near convergence (amplitude a -> 0, phase b
        # undetermined) is sensitive to the last bits of its start value p0 -- accumulated in float64 on the GPU,
        # float32 by NumPy.  Agreement is therefore asserted at the method's own convergence threshold (1e-3 px).
        assert np.allclose(offsets, o_off, rtol=0, atol=1e-3 * res)
        assert np.allclose(offsets, z[f"{pre}{tol}|offsets"], rtol=0, atol=1e-3 * res)  # the reference's own loop
        assert abs(offsets[2] - o_off[2]) < 1e-3  # (medians are exact per step; the offsets they are taken at differ as above)


def test_class_api_recovers_shift(coreg):
    ref, tba, inlier, res = _pair(shape=(300, 400))
    nk = coreg.NuthKaab(subsample=1).fit(ref, tba, inlier, resolution=res)
    out = nk.meta["outputs"]["affine"]
    assert abs(out["shift_x"] / res - 1.7) < 0.1 and abs(out["shift_y"] / res + 0.6) < 0.1 and abs(out["shift_z"] + 2.0) < 0.1
    assert nk.meta["outputs"]["random"]["subsample_final"] > 0
    assert nk.to_matrix().shape == (4, 4)
    # random subsample (reference default: 5e5 points): a subset of the valid pixels drawn once; reproducible with a seed
    full = nk.meta["outputs"]["affine"]
    nks = coreg.NuthKaab(subsample=0.5).fit(ref, tba, inlier, resolution=res, random_state=42)
    sub = nks.meta["outputs"]["affine"]
    n_all = nk.meta["outputs"]["random"]["subsample_final"]
    assert nks.meta["outputs"]["random"]["subsample_final"] == int(0.5 * n_all)
    assert abs(sub["shift_x"] - full["shift_x"]) < 0.05 * res and abs(sub["shift_y"] - full["shift_y"]) < 0.05 * res
    again = coreg.NuthKaab(subsample=0.5).fit(ref, tba, inlier, resolution=res, random_state=42).meta["outputs"]["affine"]
    # (same pixels, same exact medians; the curve_fit start values come from float64 atomics whose order varies)
    assert all(abs(again[k] - sub[k]) < 1e-6 for k in sub)
    assert coreg.NuthKaab().fit(ref, tba, inlier, resolution=res).meta["outputs"]["random"]["subsample_final"] == n_all  # 5e5 > valid
    m = coreg.subsample_valid_mask(np.array([[True, False, True], [True, True, False]]), 2, random_state=0)
    assert m.sum() == 2 and not m[0, 1] and not m[1, 2]
    with pytest.raises(ValueError, match="no valid points"):
        coreg.NuthKaab(subsample=1).fit(ref, np.full_like(tba, np.nan), None, resolution=res)


def test_device_side_subsample_equals_the_host_draw(coreg):
    """``nuth_kaab(subsample != 1)`` draws RANKS among the valid pixels on the host and lets the plan turn them into pixels
    (``xdemhip_nk_subsample``): the valid mask, the steps and the fit must be those of the host form -- the mask copied back,
    ``rng.choice(np.flatnonzero(valid), n, replace=False)``, a second plan with that mask (``coreg._HOST_DRAW``)."""
    from xdem_amd import _lib

    ref, tba, inlier, res = _pair(shape=(517, 333))   # (neither a multiple of 16 nor of the kernels' 4096-pixel tiles)
    plan = coreg.NKPlan(ref, tba, inlier)
    try:
        valid0 = plan.aux()[2]
        n0 = plan.n_valid
        assert n0 == int(valid0.sum())
        # all of them: nothing changes
        assert plan.subsample(np.random.default_rng(5).permutation(n0)) == n0
        assert np.array_equal(plan.aux()[2], valid0)
        ranks = coreg.subsample_ranks(n0, 0.3, 7)
        assert plan.subsample(ranks) == ranks.size == int(0.3 * n0)
        valid1 = plan.aux()[2]
        assert np.array_equal(valid1, coreg.subsample_valid_mask(valid0, 0.3, 7))
        # a second draw is among the pixels still valid
        ranks2 = coreg.subsample_ranks(plan.n_valid, 1000, 11)
        assert plan.subsample(ranks2) == 1000
        want = np.zeros(valid1.size, dtype=bool)
        want[np.flatnonzero(valid1.ravel())[ranks2]] = True
        assert np.array_equal(plan.aux()[2].ravel(), want)
        with pytest.raises(_lib.XdemHipError, match="outside"):
            plan.subsample(np.array([0, 1000], dtype=np.int64))
        with pytest.raises(_lib.XdemHipError, match="1 <= k"):
            plan.subsample(np.arange(1001, dtype=np.int64))
    finally:
        plan.close()
    # the steps of a subsampled plan: device form against a plan created with the host-drawn mask (one-pass route at this size)
    ref, tba, inlier, res = _pair(shape=(2304, 2304), seed=3)
    a = coreg.NKPlan(ref, tba, inlier)
    try:
        valid0 = a.aux()[2]
        a.subsample(coreg.subsample_ranks(a.n_valid, 0.5, 21))
        b = coreg.NKPlan(ref, tba, coreg.subsample_valid_mask(valid0, 0.5, 21))
        try:
            assert a.n_valid == b.n_valid
            for sx, sy in ((0.0, 0.0), (7.0, -3.0), (16.5, -6.2)):
                da, db = a.step(sx, sy, (res, res), 72), b.step(sx, sy, (res, res), 72)
                assert da["n_valid"] == db["n_valid"] and da["vshift"] == db["vshift"]
                assert np.array_equal(da["counts"], db["counts"]) and np.array_equal(da["medians"], db["medians"], equal_nan=True)
                assert np.array_equal(da["edges"], db["edges"])
        finally:
            b.close()
    finally:
        a.close()
    # ... and the whole call
    got = {}
    for host in (False, True):
        coreg._HOST_DRAW = host
        try:
            got[host] = coreg.nuth_kaab(ref, tba, inlier, (res, res), subsample=0.25, random_state=99, max_iterations=5)
        finally:
            coreg._HOST_DRAW = False
    assert got[False][1] == got[True][1] > 0
    assert np.allclose(got[False][0], got[True][0], rtol=0, atol=1e-6)


@pytest.mark.parametrize("shape", [(128, 200), (2304, 2304)])
def test_sharded_reduction_path_single_rank(coreg, shape):
    """The multi-GPU path (row range + all-reduce hook through torch.distributed) on a 1-rank NCCL group: must give
    exactly the single-process results.  (Multi-rank sums of the same integer histograms are exercised on CPU/gloo.)
    The larger shape takes the bracketed-selection route (sample / counter / candidate reductions through the hook), forced
    for the 72 aspect bins too (selection mode 3; mode 0 leaves them to the plain passes at this size)."""
    import os

    import torch.distributed as dist

    from xdem_amd import _lib

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        _lib.default_context().set_option("selection", 3 if shape[0] > 1000 else 0)
        ref, tba, inlier, res = _pair(shape)
        base = coreg.NKPlan(ref, tba, inlier)
        want = base.step(7.0, -3.0, (res, res), 72)
        base.close()
        ctx = _lib.default_context()
        h0, d0 = ctx.reduction_calls()
        plan = coreg.NKPlan(ref, tba, inlier, group="world")
        got = plan.step(7.0, -3.0, (res, res), 72)
        assert plan.n_valid == int((inlier & np.isfinite(ref) & np.isfinite(tba) & np.isfinite(nko.aux_vars(ref)[0])).sum())
        plan.close()
        # round 3: with an RCCL group the library's device arrays are all-reduced in place through the device-side hook -- no
        # staging through the host, no host synchronisation per reduction
        h1, d1 = ctx.reduction_calls()
        assert d1 - d0 >= 5 and h1 == h0, (h0, h1, d0, d1)
        # ... and the host-staged form (what gloo groups use) gives the same
        plan = coreg.NKPlan(ref, tba, inlier)
        ctx.set_allreduce("world", device_side=False)
        try:
            r0, r1 = 0, ref.shape[0]
            import ctypes

            nv = ctypes.c_int64()
            ctx.check(ctx._L.xdemhip_nk_set_rows(plan.handle, r0, r1, ctypes.byref(nv)))
            got_host = plan.step(7.0, -3.0, (res, res), 72)
        finally:
            ctx.set_allreduce(None)
            plan.close()
        h2, d2 = ctx.reduction_calls()
        assert h2 > h1 and d2 == d1
        assert got_host["vshift"] == got["vshift"] and np.array_equal(got_host["medians"], got["medians"], equal_nan=True)
        for k in ("vshift", "n_valid"):
            assert got[k] == want[k]
        assert np.array_equal(got["counts"], want["counts"]) and np.array_equal(got["medians"], want["medians"], equal_nan=True)
        assert np.array_equal(got["edges"], want["edges"])
    finally:
        _lib.default_context().set_option("selection", 0)
        if created:
            dist.destroy_process_group()


def test_row_sharding_is_exact(coreg):
    """Two row shards processed one after the other with a hook that accumulates them = the unsharded result
    (integer histograms add exactly): emulates 2 ranks on one GPU."""
    import ctypes

    ref, tba, inlier, res = _pair((120, 180))
    whole = coreg.NKPlan(ref, tba, inlier)
    want = whole.step(5.0, 2.0, (res, res), 72)
    whole.close()
    # Each "rank" runs the same sequence of reductions; record rank A's contributions, then replay them into rank B.
    H = ref.shape[0]
    logs = {0: [], 1: []}
    state = {"rank": 0, "i": 0}
    CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p)

    def hook(ptr, count, kind, user):
        a = np.frombuffer((ctypes.c_uint64 * count).from_address(ptr), dtype=np.float64 if kind == 1 else np.uint64)
        r = state["rank"]
        if r == 0:
            logs[0].append(a.copy())
            # rank 0 alone cannot know the global value yet: feed it the recorded global result of a previous full run
            if state.get("replay"):
                a[:] = state["replay"][state["i"]]
        else:
            other = logs[0][state["i"]]
            if kind in (0, 1):
                a[:] = a + other
            elif kind == 2:
                a[:] = np.minimum(a, other)
            else:
                a[:] = np.maximum(a, other)
            logs[1].append(a.copy())
        state["i"] += 1
        return 0

    cb = CB(hook)
    ctx = coreg._lib.default_context()

    def run(rank, rows):
        state["rank"], state["i"] = rank, 0
        plan = coreg.NKPlan(ref, tba, inlier, ctx)
        ctx.check(ctx._L.xdemhip_set_allreduce(ctx.handle, ctypes.cast(cb, ctypes.c_void_p), None))
        nv = ctypes.c_int64()
        ctx.check(ctx._L.xdemhip_nk_set_rows(plan.handle, rows[0], rows[1], ctypes.byref(nv)))
        out = plan.step(5.0, 2.0, (res, res), 72)
        ctx.check(ctx._L.xdemhip_set_allreduce(ctx.handle, None, None))
        plan.close()
        return out

    # pass 1: rank 0's local contributions are only meaningful once it sees global values, and its later passes depend
    # on them (selection prefixes).  Iterate: run B with A's log to get global values, replay them into A, repeat until
    # A's log is stable (2 rounds suffice: every reduction only depends on earlier global values).
    for _ in range(14):
        logs[0], logs[1] = [], []
        run(0, (0, H // 2))
        got = run(1, (H // 2, H))
        if state.get("replay") is not None and all(np.array_equal(x, y) for x, y in zip(state["replay"], logs[1])):
            break
        state["replay"] = [x.copy() for x in logs[1]]
    assert got["vshift"] == want["vshift"] and got["n_valid"] == want["n_valid"]
    assert np.array_equal(got["counts"], want["counts"]) and np.array_equal(got["medians"], want["medians"], equal_nan=True)


def test_apply_translation_matches_oracle_and_realigns(coreg):
    """f1: out(r,c) = elev(r + sy/res, c - sx/res) + sz; applying the fitted shift re-aligns the pair."""
    ref, tba, inlier, res = _pair((160, 220))
    for sx, sy, sz in ((17.0, -6.0, 2.0), (0.0, 0.0, -1.5), (-23.5, 4.25, 0.0)):
        got = coreg.apply_translation(tba, sx, sy, sz, res)
        # oracle: shifted_dh(ref=0, tba, E, N) = -tba(row - N/res, col + E/res)  with E = -sx, N = -sy
        want = -nko.shifted_dh(np.zeros_like(tba), tba, -sx, -sy, (res, res)) + tba.dtype.type(sz)
        assert np.array_equal(got, want.astype(tba.dtype), equal_nan=True)
    nk = coreg.NuthKaab(subsample=1).fit(ref, tba, inlier, resolution=res)
    aligned = nk.apply(tba, res)
    ok = np.isfinite(aligned) & np.isfinite(ref)
    before = np.isfinite(tba) & np.isfinite(ref)
    assert np.nanstd((ref - aligned)[ok]) < np.nanstd((ref - tba)[before])  # (a rough fBm is not bilinear-invertible)
    assert abs(np.nanmedian((ref - aligned)[ok])) < 0.05
    with pytest.raises(ValueError, match="all nans"):
        coreg.apply_translation(np.full((4, 4), np.nan, np.float32), 1, 1, 1, 1.0)


def test_C3_full_size_properties(coreg):
    """BASELINE config[2] size: 20000^2 float32 pair, 20 % NaN, exactly 10 iterations on every valid pixel.
    The oracle cannot sort 3e8 points in test time, so size-independent properties are checked instead."""
    from xdem_amd.synth import fbm_numpy

    n, t, res = 20000, 2500, 10.0
    tile = fbm_numpy((t, t), seed=42, std=150.0)
    yy, xx = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32), indexing="ij")
    ref = np.tile(tile, (n // t, n // t)) + 40.0 * np.sin(xx * np.float32(2 * np.pi / n)) + 40.0 * np.cos(yy * np.float32(2 * np.pi / n))
    del xx, yy
    ref = ref.astype(np.float32)
    # tba(x) = ref(x + (2, -1) px) + 2 m : an integer-pixel shift keeps the construction exact
    tba = np.roll(ref, (1, -2), axis=(0, 1)) + np.float32(2.0)
    hole = np.tile(fbm_numpy((t, t), seed=44, hurst=1.0, mean=0.0, std=1.0), (n // t, n // t))
    tba[hole < np.percentile(hole[:t, :t], 20)] = np.nan
    del hole
    plan = coreg.NKPlan(ref, tba, None)
    # (1) integer work: the valid count equals NumPy's (gradient-based aux variables are finite wherever ref is)
    gy0 = np.isfinite(tba)
    assert plan.n_valid <= int(gy0.sum()) and plan.n_valid > 0.75 * n * n
    # (2) one step: counts partition the valid set, medians finite, edges monotone
    d0 = plan.step(0.0, 0.0, (res, res), 72)
    assert d0["counts"].sum() == d0["n_valid"] <= plan.n_valid
    assert np.all(np.diff(d0["edges"]) > 0) and np.isfinite(d0["medians"]).all()
    # (3) determinism: the same step twice is bit-identical (integer histograms, no floating reductions in the medians)
    d1 = plan.step(0.0, 0.0, (res, res), 72)
    assert d1["vshift"] == d0["vshift"] and np.array_equal(d1["medians"], d0["medians"]) and np.array_equal(d1["counts"], d0["counts"])
    plan.close()
    # (4) the full 10-iteration fit recovers the construction: offsets (-2 px, +1 px... in georeferenced units) and -2 m
    offsets, n_final = coreg.nuth_kaab(ref, tba, None, (res, res), tolerance=0.0, max_iterations=10)
    assert n_final > 0.75 * n * n
    assert abs(offsets[0] / res + 2.0) < 0.02 and abs(offsets[1] / res + 1.0) < 0.02 and abs(offsets[2] + 2.0) < 0.02


def test_rows_beyond_2g_pixels_match_a_small_plan(coreg):
    """46400^2 device-resident pair whose only valid rows are the LAST 3000 (pixel offsets > 2^31): every step output
    equals that of a plan built on just those rows (tba NaN on the first two of them, as above them in the big raster)."""
    import torch

    from xdem_amd.synth import fbm_torch

    n, k, res = 46400, 3000, 10.0
    ref = fbm_torch(n, n, "cuda", seed=5)
    tba = torch.full((n, n), float("nan"), device="cuda", dtype=torch.float32)
    tba[n - k + 2:] = torch.roll(ref[n - k + 2:], shifts=-2, dims=1) + 1.5
    tba[n - k + 2:, ::7] = float("nan")
    big = coreg.NKPlan(ref, tba, None)
    small = coreg.NKPlan(ref[n - k:].contiguous(), tba[n - k:].contiguous(), None)
    assert big.n_valid == small.n_valid > 0.8 * (k - 3) * n
    # whole-pixel shifts: the bilinear weights are exact whatever the absolute row index, so everything is bit-identical
    # (a fractional shift rounds row + shift differently at row 46000 than at row 3000 -- last-ulp weights, as upstream)
    for (sx, sy) in ((0.0, 0.0), (10.0, -20.0)):
        a, b = big.step(sx, sy, (res, res), 72), small.step(sx, sy, (res, res), 72)
        assert a["n_valid"] == b["n_valid"] and np.array_equal(a["counts"], b["counts"])
        assert a["vshift"] == b["vshift"] and np.array_equal(a["edges"], b["edges"])
        bad = np.flatnonzero(a["medians"] != b["medians"])
        assert bad.size == 0, (bad, a["medians"][bad], b["medians"][bad], a["counts"][bad])
        assert _moments_close(a, b)
    a, b = big.step(1.25, -0.5, (res, res), 72), small.step(1.25, -0.5, (res, res), 72)
    assert np.array_equal(a["counts"], b["counts"])
    np.testing.assert_allclose(a["medians"], b["medians"], rtol=0, atol=1e-4)
    big.close()
    small.close()


def test_randomised_steps_vs_oracle(coreg):
    """Seeded sweep of one Nuth-Kaab iteration step: raster shapes, dtypes, NaN / inlier patterns, sub- and multi-pixel
    shifts, resolutions, numbers of aspect bins (5, 72, 150 -- the last needs two LDS sweeps) -- vertical shift, valid count,
    bin edges, per-bin counts and medians bit-exact against the oracle."""
    rng = np.random.default_rng(99)
    from xdem_amd.synth import fbm_numpy

    for trial in range(24):
        H, W = int(rng.integers(12, 220)), int(rng.integers(12, 257))
        dtype = rng.choice([np.float32, np.float64])
        res = float(rng.choice([1.0, 2.5, 30.0]))
        base = fbm_numpy((256, 256), seed=int(rng.integers(0, 1000)), std=float(rng.choice([5.0, 200.0])))[:H, :W]
        ref = base.astype(dtype)
        tba = (np.roll(base, (int(rng.integers(-2, 3)), int(rng.integers(-2, 3))), (0, 1)) + rng.normal(0, 0.2, (H, W)) + 1.0).astype(dtype)
        tba[rng.uniform(size=(H, W)) < float(rng.choice([0.0, 0.05, 0.4]))] = np.nan
        if rng.uniform() < 0.3:
            ref[rng.integers(0, H), :] = np.nan
        inlier = None if rng.uniform() < 0.4 else (rng.uniform(size=(H, W)) < 0.8)
        nb = int(rng.choice([5, 72, 150]))
        sx, sy = float(rng.uniform(-3, 3) * res), float(rng.uniform(-3, 3) * res)
        plan = coreg.NKPlan(ref, tba, inlier)
        st, asp = nko.aux_vars(ref)
        valid = (np.ones((H, W), bool) if inlier is None else inlier) & np.isfinite(ref) & np.isfinite(tba) & np.isfinite(st) & np.isfinite(asp)
        dh = nko.shifted_dh(ref, tba, sx, sy, (res, res))[valid]
        ok = np.isfinite(dh)
        if not ok.any():
            with pytest.raises(ValueError, match="no more valid values"):
                plan.step(sx, sy, (res, res), nb)
            plan.close()
            continue
        det = plan.step(sx, sy, (res, res), nb)
        vshift = np.nanmedian(dh)
        assert det["vshift"] == float(vshift), (trial, H, W, dtype)
        dh = dh - vshift
        assert det["n_valid"] == int(ok.sum())
        with np.errstate(all="ignore"):
            y = dh[ok] / st[valid][ok]
        a = asp[valid][ok]
        if dtype == np.float64:
            a = plan.aux()[1][valid][ok]
        edges, counts, med = nko.bin_medians(a, y, nb)
        plan.close()
        assert np.array_equal(det["counts"], counts), (trial, nb)
        assert np.array_equal(det["edges"], edges.astype(np.float64)), (trial, nb)
        assert np.array_equal(det["medians"], med, equal_nan=True), (trial, nb, H, W, dtype)


def test_unbinned_fit_mode_vs_oracle(coreg):
    """NuthKaab(bin_before_fit=False) -- the mode of the reference's synthetic tests (tests/test_coreg/test_affine.py:163-239):
    the step returns the least-squares sums over every valid point; the fitted offsets equal curve_fit on all points (oracle:
    scipy.optimize.curve_fit exactly as xdem/coreg/base.py:975-989 calls it) to 1e-6."""
    import scipy.optimize

    ref, tba, inlier, res = _pair((150, 210))
    plan = coreg.NKPlan(ref, tba, inlier)
    det = plan.step_fit(4.0, -6.0, (res, res))
    plan.close()
    st, asp = nko.aux_vars(ref)
    valid = inlier & np.isfinite(ref) & np.isfinite(tba) & np.isfinite(st) & np.isfinite(asp)
    dh = nko.shifted_dh(ref, tba, 4.0, -6.0, (res, res))
    ok = valid & np.isfinite(dh)
    vs = np.nanmedian(dh[ok])
    assert det["vshift"] == float(vs) and det["n_valid"] == int(ok.sum())
    y = ((dh[ok] - vs) / st[ok]).astype(np.float64)
    x = asp[ok].astype(np.float64)
    assert np.isclose(det["y_mean"], y.mean(), rtol=1e-9) and np.isclose(det["y_std"], y.std(), rtol=1e-6)
    p0 = (3 * np.nanstd(y) / (2**0.5), 0.0, np.nanmean(y))
    (a, b, c), _ = scipy.optimize.curve_fit(nko.fit_func, x, y, p0=p0, absolute_sigma=True)
    east, north, vert = coreg._fit_from_sums(det)
    # to 1e-6 of the fitted amplitude (both sides stop a Levenberg-Marquardt iteration at ftol = xtol = 1e-8: the small third
    # parameter is not known to 1e-6 of ITSELF -- under nodata rule 3 the two differed by 6e-6 of c = 3e-7 of a)
    assert np.allclose([east, north, vert], [a * np.sin(b), a * np.cos(b), c], rtol=1e-6, atol=1e-6 * abs(a))
    # and the class recovers a synthetic shift in this mode, with an initial shift
    nk = coreg.NuthKaab(bin_before_fit=False, subsample=1, initial_shift=(5.0, -5.0))
    nk.fit(ref, tba, inlier, resolution=res)
    nb = coreg.NuthKaab(subsample=1).fit(ref, tba, inlier, resolution=res)
    for k in ("shift_x", "shift_y"):
        assert abs(nk.meta["outputs"]["affine"][k] - nb.meta["outputs"]["affine"][k]) < 0.1 * res


def test_explicit_bin_edges_vs_oracle(coreg):
    """bin_sizes as an array of edges (scipy.stats.binned_statistic's `bins` sequence): counts / medians exact vs the oracle."""
    import scipy.stats

    ref, tba, inlier, res = _pair((140, 190))
    edges = np.array([0.3, 1.0, 1.5, 2.9, 3.0, 4.4, 6.2])
    plan = coreg.NKPlan(ref, tba, inlier)
    plan.set_bin_edges(edges)
    det = plan.step(2.0, 1.0, (res, res))
    plan.set_bin_edges(None)
    auto = plan.step(2.0, 1.0, (res, res), 72)
    plan.close()
    assert len(auto["counts"]) == 72 and len(det["counts"]) == 6
    st, asp = nko.aux_vars(ref)
    valid = inlier & np.isfinite(ref) & np.isfinite(tba) & np.isfinite(st) & np.isfinite(asp)
    dh = nko.shifted_dh(ref, tba, 2.0, 1.0, (res, res))
    ok = valid & np.isfinite(dh)
    y = (dh[ok] - np.nanmedian(dh[ok])) / st[ok]
    med = scipy.stats.binned_statistic(asp[ok], y, statistic=np.nanmedian, bins=edges.astype(np.float32))[0]
    cnt = scipy.stats.binned_statistic(asp[ok], y, statistic="count", bins=edges.astype(np.float32))[0]
    assert np.array_equal(det["counts"], cnt.astype(np.int64))
    assert np.array_equal(det["medians"], med.astype(np.float64), equal_nan=True)


@pytest.mark.parametrize("rule", [0, 1, 2, 3])
def test_nan_rules_of_the_bilinear_taps(coreg, rule):
    """Context option "nk_nan_rule": the three nodata conventions of the bilinear taps (geoutils' own is unpinned), step and
    translation resample, each bit-exact against the oracle's implementation of the same rule.  Rule 1 keeps the last row /
    column at integer shifts (the ADVICE item of round 1)."""
    ctx = coreg._lib.default_context()
    ref, tba, inlier, res = _pair((110, 130))
    try:
        ctx.set_option("nk_nan_rule", rule)
        for sx, sy in ((0.0, 0.0), (res * 2.0, -res * 1.0), (3.3, -7.1)):
            plan = coreg.NKPlan(ref, tba, inlier)
            det = plan.step(sx, sy, (res, res), 72)
            plan.close()
            st, asp = nko.aux_vars(ref)
            valid = inlier & np.isfinite(ref) & np.isfinite(tba) & np.isfinite(st) & np.isfinite(asp)
            dh = nko.shifted_dh(ref, tba, sx, sy, (res, res), nan_rule=rule)
            ok = valid & np.isfinite(dh)
            assert det["n_valid"] == int(ok.sum()) and det["vshift"] == float(np.nanmedian(dh[ok]))
            out = coreg.apply_translation(tba, sx, sy, 0.5, (res, res))
            want = nko.bilinear_shifted(tba, sy / res, -sx / res, nan_rule=rule) + np.float32(0.5)
            assert np.array_equal(out, want, equal_nan=True)
        if rule == 1:
            out = coreg.apply_translation(tba, 0.0, 0.0, 0.0, (res, res))
            assert np.array_equal(out, tba, equal_nan=True)   # zero shift = identity, last row and column included
    finally:
        ctx.set_option("nk_nan_rule", decided("nk_nan_rule"))


@pytest.mark.parametrize("rule", [0, 1, 2, 3])
def test_block_plan_halo_depth_per_nan_rule(coreg, rule):
    """A partitioned plan (row block + halo rows) reads exactly the rows the single-GPU step reads: with the dilation rules (2,
    3) that is one row more than the four bilinear taps once the fractional row shift reaches one half.
    Two blocks of one raster, each stepped alone: too thin a halo is refused (HaloTooSmall, not a silently different dh), a
    sufficient one gives block counts of finite dh that add up to the unpartitioned step's."""
    ctx = coreg._lib.default_context()
    ref, tba, inlier, res = _pair((160, 130))
    H, cut = ref.shape[0], 77
    sy = -res * 2.6    # dr = +2.6 rows: taps in rows floor(r + 2.6) .. + 1 -> 3 halo rows; nearest row round(r + 2.6) +- 1 -> 4
    need = 4 if rule >= 2 else 3

    def blocks(halo):
        out = []
        for rb, re_ in ((0, cut), (cut, H)):
            ht, hb = min(halo, rb), min(halo, H - re_)
            sl = slice(rb - ht, re_ + hb)
            out.append(coreg.NKPlan(ref[sl], tba[sl], inlier[sl], block=(H, rb, re_, ht, hb)))
        return out

    try:
        ctx.set_option("nk_nan_rule", rule)
        full = coreg.NKPlan(ref, tba, inlier)
        want = full.step(3.3, sy, (res, res), 72)["n_valid"]
        full.close()
        for halo in (need - 1, need):
            plans = blocks(halo)
            try:
                if halo < need:
                    with pytest.raises(coreg.HaloTooSmall):
                        plans[0].step(3.3, sy, (res, res), 72)
                else:
                    got = sum(p.step(3.3, sy, (res, res), 72)["n_valid"] for p in plans)
                    assert got == want
            finally:
                for p in plans:
                    p.close()
    finally:
        ctx.set_option("nk_nan_rule", decided("nk_nan_rule"))


def test_advice_round2_contracts(coreg):
    """(i) explicit bin edges: the C entry refuses an n_bins that does not match them (its outputs are sized by n_bins);
    (ii) a partitioned plan whose creation fails leaves no reduction hook on the context; (iii) the un-binned fit refuses
    optimisers it cannot honour and survives a degenerate system."""
    import ctypes

    import scipy.optimize

    ref, tba, inlier, res = _pair((140, 150))
    plan = coreg.NKPlan(ref, tba, inlier)
    plan.set_bin_edges(np.array([0.5, 1.0, 2.0, 4.0]))
    L, dp, ip = plan.ctx._L, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)
    e, c, m = np.empty(73), np.empty(72, dtype=np.int64), np.empty(72)
    v = [ctypes.c_double() for _ in range(3)]
    nv = ctypes.c_int64()
    rc = L.xdemhip_nk_step(plan.handle, 0.0, 0.0, res, res, 72, ctypes.byref(v[0]), ctypes.byref(nv), ctypes.byref(v[1]),
                           ctypes.byref(v[2]), e.ctypes.data_as(dp), c.ctypes.data_as(ip), m.ctypes.data_as(dp))
    assert rc == -1   # XDEMHIP_EINVAL
    assert len(plan.step(0.0, 0.0, (res, res))["counts"]) == 3   # the wrapper passes n_edges - 1
    plan.close()
    with pytest.raises(NotImplementedError, match="128 bins"):
        coreg.NuthKaab(bin_sizes=np.linspace(0, 6.3, 200))
    # (ii) hook hygiene: creation fails (wrong row count) after set_allreduce
    import torch.distributed as dist

    created = False
    if not dist.is_initialized():
        import os

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29641")
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        with pytest.raises(ValueError, match="block arrays hold"):
            coreg.NKPlan(ref[:50], tba[:50], inlier[:50], group="world", block=(140, 0, 60, 0, 2))
        assert getattr(coreg._lib.default_context(), "_hook", None) is None
        ok = coreg.NKPlan(ref, tba, inlier)     # a plain single-process plan still steps (no collective entered)
        assert ok.step(0.0, 0.0, (res, res), 72)["n_valid"] > 0
        ok.close()
    finally:
        if created:
            dist.destroy_process_group()
    # (iii)
    with pytest.raises(NotImplementedError, match="bin_before_fit=False"):
        coreg.NuthKaab(bin_before_fit=False, fit_optimizer=scipy.optimize.least_squares)
    with pytest.raises(NotImplementedError, match="bin_before_fit=False"):
        coreg.nuth_kaab(ref, tba, inlier, (res, res), bin_before_fit=False, fit_optimizer=lambda **k: None)
    s = np.array([5.0, 5 * np.cos(1.0), 5 * np.sin(1.0), 5 * np.cos(1.0) ** 2, 5 * np.sin(1.0) ** 2, 5 * np.cos(1.0) * np.sin(1.0),
                  10.0, 10 * np.cos(1.0), 10 * np.sin(1.0), 20.0])   # five points, all at aspect 1.0, y = 2: singular normal equations
    east, north, c0 = coreg._fit_from_sums({"sums": s})
    assert np.isfinite([east, north, c0]).all() and abs(north * np.cos(1.0) + east * np.sin(1.0) + c0 - 2.0) < 1e-9


def test_lean_route_equals_plain_route_at_scale():
    """At BASELINE's C3 scale the oracle is too slow, but the product has two independent implementations of a step: the queued
    route (lean dh / bin kernels, bracketed selections on samples and candidates, aspect-bin cache) and the plain one (generic
    kernels, y / bin-id arrays, full digit passes; option "selection" = 1).  Every output of every step must be identical, also
    when a step is repeated after others (cache reuse) -- 12000^2 pair with noise and 20 % gaps."""
    import torch

    from xdem_amd import _lib, coreg
    from xdem_amd.synth import fbm_torch

    m = 12000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    ref = fbm_torch(m, m, dev, seed=42)
    tba = torch.roll(ref, shifts=(1, -2), dims=(0, 1)) + 2.0 + 0.5 * torch.randn((m, m), device=dev, generator=g)
    hole = fbm_torch(m, m, dev, seed=44)
    tba[hole < torch.quantile(hole[::16, ::16].flatten(), 0.2)] = float("nan")
    del hole
    torch.cuda.synchronize()
    ctx = _lib.Context(0)
    try:
        res = {}
        # three implementations: the one-pass step (default); the plain route (option "nk_fused" = 0: stored dh, generic kernels) with
        # bracketed selections over the stored arrays; the same with plain digit passes only (option "selection" = 1)
        for name, mode, fused, route in (("onepass", 0, 1, "onepass"), ("plain, bracketed", 0, 0, "plain"), ("plain", 1, 0, "plain")):
            ctx.set_option("selection", mode)
            ctx.set_option("nk_fused", fused)
            plan = coreg.NKPlan(ref, tba, None, ctx)
            res[name] = [plan.step(sx, sy, (10.0, 10.0), 72) for (sx, sy) in ((0.0, 0.0), (3.0, -4.0), (13.7, 21.3), (3.0, -4.0))]
            assert plan.route_counts()[route] == 4, (name, plan.route_counts())   # every step answered by the route under test
            plan.close()
        # the one-pass step with its sample brackets at a fixed fraction of the rule (option "nk_narrow"; default: adaptive):
        # full width answers every step itself; a quarter may miss and hand a step to the plain route -- exact either way
        ctx.set_option("selection", 0)
        ctx.set_option("nk_fused", 1)
        for k in (0, 1, 2):
            ctx.set_option("nk_narrow", k)
            plan = coreg.NKPlan(ref, tba, None, ctx)
            res[f"narrow{k}"] = [plan.step(sx, sy, (10.0, 10.0), 72) for (sx, sy) in ((0.0, 0.0), (3.0, -4.0), (13.7, 21.3), (3.0, -4.0))]
            rc = plan.route_counts()
            assert rc["onepass"] + rc["plain"] == 4 and (k > 0 or rc["onepass"] == 4), (k, rc)
            print(f"nk_narrow = {k}: routes {rc}")
            plan.close()
        ctx.set_option("nk_narrow", -1)
        for name in ("onepass", "plain, bracketed", "narrow0", "narrow1", "narrow2"):
            for a, b in zip(res[name], res["plain"]):
                assert a["n_valid"] == b["n_valid"] and a["vshift"] == b["vshift"], name
                assert np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["medians"], b["medians"], equal_nan=True), name
                assert np.array_equal(a["edges"], b["edges"]), name
                assert _moments_close(a, b, onepass=name != "plain, bracketed"), (name, a["y_mean"], b["y_mean"], a["y_std"], b["y_std"])
            assert res[name][1]["n_valid"] == res[name][3]["n_valid"] and np.array_equal(res[name][1]["medians"], res[name][3]["medians"], equal_nan=True)
    finally:
        ctx.close()


@pytest.mark.parametrize("rule", [2, 3])
def test_dilating_nan_rules_all_routes_agree_at_scale(rule):
    """Rules 2 / 3 of the bilinear taps (the forms with a dilated nodata mask -- rule 3 is what the only hint about geoutils'
    convention favours) on a 9000^2 pair with noise, 20 % contiguous gaps and scattered single-pixel holes,
    36 aspect bins (about as many sampled values per bin as the 12000^2 / 72-bin case above: brackets the one-pass step can use):
    the one-pass step (streaming kernel + the plan's bad-bit mask) against the plain route (generic kernel, a 3 x 3 / cross
    neighbourhood read per pixel; with bracketed selections and with plain digit passes): every output of every step identical, fractional and integer
    shifts, a repeated step; and the default rule gives a DIFFERENT valid count on the same pair (the rules do differ here)."""
    import torch

    from xdem_amd import _lib, coreg
    from xdem_amd.synth import fbm_torch

    m, nbin = 9000, 36
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    ref = fbm_torch(m, m, dev, seed=42)
    tba = torch.roll(ref, shifts=(1, -2), dims=(0, 1)) + 2.0 + 0.5 * torch.randn((m, m), device=dev, generator=g)
    hole = fbm_torch(m, m, dev, seed=44)
    tba[hole < torch.quantile(hole[::16, ::16].flatten(), 0.2)] = float("nan")
    tba[torch.rand((m, m), device=dev, generator=g) < 0.01] = float("nan")
    del hole
    torch.cuda.synchronize()
    ctx = _lib.Context(0)
    steps = ((0.0, 0.0), (3.0, -4.0), (13.7, 21.3), (-10.0, 20.0), (4.999999999, -5.000000001), (3.0, -4.0))
    try:
        res = {}
        ctx.set_option("nk_nan_rule", rule)
        for name, mode, fused in (("onepass", 0, 1), ("plain, bracketed", 0, 0), ("plain", 1, 0)):
            ctx.set_option("selection", mode)
            ctx.set_option("nk_fused", fused)
            plan = coreg.NKPlan(ref, tba, None, ctx)
            try:
                res[name] = [plan.step(sx, sy, (10.0, 10.0), nbin) for (sx, sy) in steps]
                assert plan.route_counts()[name.split(",")[0]] == len(steps), (name, plan.route_counts())
            finally:
                plan.close()
        for name in ("onepass", "plain, bracketed"):
            for a, b in zip(res[name], res["plain"]):
                assert a["n_valid"] == b["n_valid"] and a["vshift"] == b["vshift"], (rule, name)
                assert np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["medians"], b["medians"], equal_nan=True), (rule, name)
                assert np.array_equal(a["edges"], b["edges"]), (rule, name)
                assert _moments_close(a, b, onepass=name == "onepass"), (rule, name)
            assert np.array_equal(res[name][1]["medians"], res[name][5]["medians"], equal_nan=True)
        ctx.set_option("nk_nan_rule", 0)
        ctx.set_option("selection", 0)
        ctx.set_option("nk_fused", 1)
        plan = coreg.NKPlan(ref, tba, None, ctx)
        d0 = plan.step(3.0, -4.0, (10.0, 10.0), nbin)
        plan.close()
        assert d0["n_valid"] > res["plain"][1]["n_valid"]   # rule 0's mask is a subset of the dilating rules' at this shift
    finally:
        ctx.close()


def test_whole_fit_stays_on_the_one_pass_route():
    """bench.py's whole-fit leg in small: the iteration converges, the pair ends aligned, dh collapses onto a few float32 values
    (differences of ~1e3 m elevations are multiples of 1.2e-4 m) -- every step must still be answered by the one-pass route (value /
    key buckets: a bucket that is ONE key needs no gathered keys), and two runs end at the same offsets."""
    import os
    import sys

    import scipy.optimize
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from xdem_amd import _lib, coreg

    dev = torch.device("cuda", 0)
    ref, tba = bench._c3_pair(dev, 9000)
    got = {}
    for form in (1, 0):   # (two runs of the one form: round 4's digit-pass selections, the other form of rounds 4-5, went in round 6)
        ctx = _lib.Context(0)
        try:
            plan = coreg.NKPlan(ref.contiguous(), tba.contiguous(), None, ctx)
            off = coreg._iterate(plan, (10.0, 10.0), 0.0, 8, 36, scipy.optimize.curve_fit, True)   # (36 bins: enough sample per bin at this size)
            routes = plan.route_counts()
            plan.close()
            assert routes["plain"] == 0 and routes["onepass"] >= 3, (form, routes)
            got[form] = off
        finally:
            ctx.close()
    # BIT FOR BIT since the end of round 6: nanmean / nanstd of y -- the p0 of the curve fit -- used to be float64 atomics over the pass's
    # workgroups in whatever order they finished; a start value that differed in its last bits moved the fitted shift by ~1e-12, the
    # next step's exact medians answered with a jump of one float32 spacing, and eight iterations of it were seen to end 1.9e-4 m
    # apart.  The sums are now added in a fixed order (nk_sums_reduce): two runs of a fit are the same numbers.
    assert tuple(float(v) for v in got[1]) == tuple(float(v) for v in got[0]), (got[1], got[0])
    assert abs(got[1][0] + 17.0) < 0.05 and abs(got[1][1] + 6.0) < 0.05 and abs(got[1][2] + 2.0) < 0.01, got[1]


def test_predicted_brackets_return_the_sampled_steps_integers():
    """Round 6: a settled one-pass step takes its 73 brackets from the previous step's exact medians moved by the Nuth-Kaab model for
    the change of the shift (no sample kernels, no digit passes over samples: option "nk_predict", default on).  A sequence of
    steps shaped like a converging fit -- large corrections first, then changes of a thousandth of a pixel -- with the option on and
    off: every integer and every median of every step identical; the settled steps really are predicted; a jump back to the start
    is sampled again; and a whole fit ends where the sampled fit ends."""
    import os
    import sys

    import scipy.optimize
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from xdem_amd import _lib, coreg

    dev = torch.device("cuda", 0)
    ref, tba = bench._c3_pair(dev, 9000)
    steps = ((0.0, 0.0), (-15.0, -5.0), (-16.8, -5.9), (-16.98, -5.99), (-17.0, -6.0), (-17.004, -6.002), (-17.0045, -6.0016),
             (-17.0046, -6.0017), (-17.0046, -6.0017), (-17.0041, -6.0020), (0.0, 0.0), (-17.0, -6.0))
    res, routes, fits = {}, {}, {}
    for on in (1, 0):
        ctx = _lib.Context(0)
        try:
            ctx.set_option("nk_predict", on)
            plan = coreg.NKPlan(ref.contiguous(), tba.contiguous(), None, ctx)
            res[on] = [plan.step(sx, sy, (10.0, 10.0), 36) for (sx, sy) in steps]
            routes[on] = plan.route_counts()
            plan.close()
            plan = coreg.NKPlan(ref.contiguous(), tba.contiguous(), None, ctx)
            fits[on] = (coreg._iterate(plan, (10.0, 10.0), 0.0, 9, 36, scipy.optimize.curve_fit, True), plan.route_counts())
            plan.close()
        finally:
            ctx.close()
    print("routes with / without prediction:", routes[1], routes[0], "whole fit:", fits[1][1], fits[0][1])
    for on in (1, 0):
        assert routes[on]["onepass"] == len(steps) and routes[on]["plain"] == 0, routes[on]
    assert routes[0]["predicted"] == 0 and routes[1]["predicted"] >= 3, routes
    assert routes[1]["predict_missed"] <= 1, routes[1]
    for k, (a, b) in enumerate(zip(res[1], res[0])):
        assert a["n_valid"] == b["n_valid"] and a["vshift"] == b["vshift"], k
        assert np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["medians"], b["medians"], equal_nan=True), k
        assert np.array_equal(a["edges"], b["edges"]), k
        assert _moments_close(a, b, onepass=True), k
    # the whole fit: predicted steps in it, the same end point as the sampled fit to what two runs of one form agree to (see
    # test_whole_fit_stays_on_the_one_pass_route)
    assert fits[1][1]["predicted"] >= 2 and fits[1][1]["plain"] == 0, fits[1][1]
    assert np.allclose(fits[1][0], fits[0][0], rtol=0, atol=1e-4), (fits[1][0], fits[0][0])


def test_bench_C3_pair_at_full_size_routes_agree():
    """The very input bench.py times (SURVEY 8d's C3: 20000^2 pair, tba = ref shifted bilinearly by (+1.7, -0.6) px + 2 m + noise,
    20 % gaps) at full size: the queued route the bench runs (EXT dh pass, lean kernels, dual bracket selections, aspect-bin cache)
    against the plain one (option "selection" = 1: generic kernels, full digit passes) -- every integer output identical for
    fractional steps, repeated steps included; and the fit the bench reports recovers the construction."""
    import os
    import sys

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from xdem_amd import _lib, coreg

    dev = torch.device("cuda", 0)
    ref, tba = bench._c3_pair(dev, 20000)
    ctx = _lib.Context(0)
    try:
        # (-17, -6): the shift the fit converges to -- the pair ALIGNED: dh = offset + small noise, and a float32 difference of ~1e3 m
        #  elevations is a multiple of their ulp: a dozen distinct values carry all the candidates of the median (ties en masse; the
        #  value-bucket selection of round 5 first sent exactly these steps to the fall-back route -- bench.py's whole-fit leg found it)
        steps = ((0.0, 0.0), (1.7, 0.6), (1.2, 0.9), (1.7, 0.6), (-17.0, -6.0), (-16.9998, -5.9996))
        res = {}
        for name, mode, fused in (("onepass", 0, 1), ("plain, bracketed", 0, 0), ("plain", 1, 0)):
            ctx.set_option("selection", mode)
            ctx.set_option("nk_fused", fused)
            plan = coreg.NKPlan(ref, tba, None, ctx)
            res[name] = [plan.step(sx, sy, (10.0, 10.0), 72) for (sx, sy) in steps]
            assert plan.route_counts()[name.split(",")[0]] == len(steps), (name, plan.route_counts())
            plan.close()
        ctx.set_option("selection", 0)
        for name in ("onepass", "plain, bracketed"):
            for a, b in zip(res[name], res["plain"]):
                assert a["n_valid"] == b["n_valid"] and a["vshift"] == b["vshift"], name
                assert np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["medians"], b["medians"], equal_nan=True), name
                assert np.array_equal(a["edges"], b["edges"]), name
                assert _moments_close(a, b, onepass=name == "onepass"), name
            assert np.array_equal(res[name][1]["medians"], res[name][3]["medians"], equal_nan=True)
        assert 0.75 * 4e8 < res["onepass"][0]["n_valid"] < 0.85 * 4e8
    finally:
        ctx.close()


@pytest.mark.parametrize("fused,n_bins", [(1, 8), (0, 72)])
@pytest.mark.parametrize("dtype,rule", [(np.float32, 0), (np.float32, 1), (np.float64, 0), (np.float32, 2), (np.float32, 3),
                                        (np.float64, 3)])
def test_lean_kernels_vs_oracle(coreg, dtype, rule, fused, n_bins):
    """The queued route (lean dh / bin kernels, bracketed selections, aspect-bin cache) only runs from 2^22 pixels on: a
    2100 x 2050 pair, selection mode 3 (bracketed route for the 72 bins whatever their sample size), against the oracle for
    shifts that exercise the row-tap table and the carried lerps -- zero, integer, negative, beyond one pixel, a hair below an
    integer (pos = i + dr rounds to the next tap row for large i) -- repeated so that the aspect-bin cache is both filled and
    reused.  Vertical shift, valid count, edges, per-bin counts and medians bit-exact.
    Round 4: the one-pass step (option "nk_fused" = 1) on the same pair with 8 aspect bins -- at 4.3 M pixels a 1/64 sample gives
    72 bins ~900 values each, brackets that hold most of a bin, more candidates than the buffers take (the step then falls
    through to the two passes, which is what the 72-bin case runs); 8 bins bracket tightly enough for the route to answer.
    Round 5: rules 2 / 3 ("dilate3x3" / "dilate_cross") run the same streaming kernels through the plan's bad-bit mask
    (nk_badbits_kernel: the neighbourhood test of the nearest pixel, evaluated once per plan) instead of the generic kernels."""
    from xdem_amd.synth import fbm_numpy

    ctx = coreg._lib.default_context()
    H, W, res = 2100, 2050, 10.0
    rng = np.random.default_rng(17)
    base = fbm_numpy((H, W), seed=7, std=150.0)
    ref = base.astype(dtype)
    tba = (np.roll(base, (1, -2), (0, 1)) + rng.normal(0, 0.3, (H, W)) + 1.5).astype(dtype)
    tba[rng.uniform(size=(H, W)) < 0.03] = np.nan
    tba[700:760, 900:1400] = np.nan
    inlier = rng.uniform(size=(H, W)) < 0.9
    st, asp = nko.aux_vars(ref)
    valid = inlier & np.isfinite(ref) & np.isfinite(tba) & np.isfinite(st) & np.isfinite(asp)
    try:
        ctx.set_option("nk_nan_rule", rule)
        ctx.set_option("selection", 3)
        ctx.set_option("nk_fused", fused)   # the one-pass step (1, default) / the plain route with bracketed selections (0)
        plan = coreg.NKPlan(ref, tba, inlier)
        if dtype == np.float64:
            asp = plan.aux()[1]
        for sx, sy in ((0.0, 0.0), (res * 2.0, -res * 1.0), (3.3, -7.1), (-13.7, 21.3), (0.0, -res * 0.9999999999999999), (3.3, -7.1)):
            det = plan.step(sx, sy, (res, res), n_bins)
            dh = nko.shifted_dh(ref, tba, sx, sy, (res, res), nan_rule=rule)[valid]
            ok = np.isfinite(dh)
            vshift = np.nanmedian(dh)
            assert det["n_valid"] == int(ok.sum()), (sx, sy)
            assert det["vshift"] == float(vshift), (sx, sy)
            with np.errstate(all="ignore"):
                y = (dh - vshift)[ok] / st[valid][ok]
            edges, counts, med = nko.bin_medians(asp[valid][ok], y, n_bins)
            assert np.array_equal(det["counts"], counts), (sx, sy)
            assert np.array_equal(det["edges"], edges.astype(np.float64)), (sx, sy)
            assert np.array_equal(det["medians"], med, equal_nan=True), (sx, sy)
            y64 = y.astype(np.float64)
            assert _moments_close(det, {"y_mean": float(y64.mean()), "y_std": float(y64.std())}, onepass=bool(fused)), (sx, sy)
        rc_ = plan.route_counts()
        assert rc_["onepass" if fused else "plain"] == 6 and rc_["plain" if fused else "onepass"] == 0, rc_
        plan.close()
    finally:
        ctx.set_option("nk_nan_rule", decided("nk_nan_rule"))
        ctx.set_option("selection", 0)
        ctx.set_option("nk_fused", 1)


def test_ext_lists_equal_the_aspect_reading_route_and_fall_back(coreg, capfd, monkeypatch):
    """The one-pass step takes min / max aspect from the plan's lists of extreme-aspect pixels and reads a masked copy of the
    reference DEM instead of mask + aspect; the plain route (option "nk_fused" = 0) reads both for every pixel.  (i) Steps of
    both routes are identical field by field; (ii) when every listed pixel loses its dh -- here: the only rows whose aspects
    reach the ends of [0, 2 pi) are VALID (tba is finite at every other row of the top half) but lose their dh at every shift
    (the lower tap row is missing) -- the first step notices (counter [6] of its device block; one "falls through" line under
    XDEMHIP_DEBUG), the plain route answers it, the plan gives its lists up (no second attempt), and the results stay
    identical: bin edges from the aspects of the pixels that do keep a dh."""
    from xdem_amd.synth import fbm_numpy

    ctx = coreg._lib.default_context()
    H, W, res, nb = 2200, 2048, 10.0, 8    # (8 bins: the one-pass step wants ~460 k pixels per bin)
    rng = np.random.default_rng(23)
    base = fbm_numpy((H, W), seed=9, std=120.0)
    # bottom half tilted towards one side: its aspects stay within [1.29, 1.87], the extremes all come from the top half
    tilt = np.zeros((H, W), dtype=np.float32)
    tilt[H // 2:] = (np.arange(W, dtype=np.float32) * 20.0)[None, :]
    ref = (base + tilt).astype(np.float32)
    tba_full = (np.roll(ref, (1, 0), (0, 1)) + rng.normal(0, 0.2, (H, W)).astype(np.float32) + 0.7).astype(np.float32)
    tba_full[rng.uniform(size=(H, W)) < 0.02] = np.nan
    tba_comb = tba_full.copy()
    tba_comb[1: H // 2 + 8: 2] = np.nan      # top half: every other row of tba missing -> valid pixels without a dh at any shift
    keys = ("vshift", "n_valid")
    steps = ((0.0, 0.0), (0.0, -12.1), (0.0, -12.1))   # (shifts along the rows: the tilt does not enter dh)
    monkeypatch.setenv("XDEMHIP_DEBUG", "1")
    for name, tba in (("full", tba_full), ("comb", tba_comb)):
        got, routes, fell = {}, {}, {}
        for fused in (1, 0):
            ctx.set_option("nk_fused", fused)
            try:
                capfd.readouterr()
                plan = coreg.NKPlan(ref, tba, None)
                got[fused] = [plan.step(sx, sy, (res, res), nb) for sx, sy in steps]
                routes[fused] = plan.route_counts()
                valid, asp = plan.aux()[2].astype(bool), plan.aux()[1]
                plan.close()
                fell[fused] = capfd.readouterr().err.count("one-pass step falls through")
            finally:
                ctx.set_option("nk_fused", 1)
        assert routes[0]["plain"] == 3 and routes[0]["onepass"] == 0 and fell[0] == 0, (name, routes, fell)
        if name == "full":   # (i) the lists answer every step
            assert routes[1]["onepass"] == 3 and routes[1]["plain"] == 0 and fell[1] == 0, (routes, fell)
        else:                # (ii) the first step notices, hands over, and the plan does not try its lists again
            assert routes[1]["onepass"] == 0 and routes[1]["plain"] == 3 and fell[1] == 1, (routes, fell)
        for a, b in zip(got[1], got[0]):
            assert all(a[k] == b[k] for k in keys), name
            # (the two moments are float64 atomics on the plain route: their last bits depend on the order of the additions)
            assert _moments_close(a, b, onepass=name == "full"), name
            assert np.array_equal(a["edges"], b["edges"]) and np.array_equal(a["counts"], b["counts"]), name
            assert np.array_equal(a["medians"], b["medians"], equal_nan=True), name
        if name == "comb":
            # the premise of (ii): valid pixels of the top half reach aspect extremes the surviving half does not -- and the edges
            # are the survivors'
            top = np.where(valid[: H // 2], asp[: H // 2], np.nan)
            bot = np.where(valid[H // 2 + 10:], asp[H // 2 + 10:], np.nan)
            assert np.nanmin(top) < np.nanmin(bot) and np.nanmax(top) > np.nanmax(bot)
            assert got[1][0]["edges"][0] > float(np.nanmin(top)) and got[1][0]["edges"][-1] < float(np.nanmax(top))


def test_device_side_reductions_cost_little():
    """SURVEY 8e / round-2 review: the sharded step used to make ~25 host round trips (D2H + sync + hook + H2D + sync per digit
    pass).  With the device-side hook on a 1-rank RCCL group a 12000^2 step must stay within 25 % of the hook-less step
    (the reductions are enqueued on the library's stream; what remains on the host are the route agreements).  Both steps take the
    plain route (option "nk_fused" = 0), the one with a reduction per key digit."""
    import os
    import time

    import torch
    import torch.distributed as dist

    from xdem_amd import _lib, coreg
    from xdem_amd.synth import fbm_torch

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        m = 12000
        ref = fbm_torch(m, m, "cuda", seed=42)
        tba = torch.roll(ref, shifts=(1, -2), dims=(0, 1)) + 1.0 + 0.3 * torch.randn((m, m), device="cuda")
        tba[torch.rand((m, m), device="cuda") < 0.1] = float("nan")
        torch.cuda.synchronize()
        ctx = _lib.default_context()
        ctx.set_option("nk_fused", 0)
        times = {}
        res = {}
        for mode in ("plain", "hooked"):
            plan = coreg.NKPlan(ref, tba, None, ctx, group="world" if mode == "hooked" else None)
            plan.step(0.0, 0.0, (10.0, 10.0), 72)
            h0, d0 = ctx.reduction_calls()
            t0 = time.perf_counter()
            for i in range(3):
                res[mode] = plan.step(3.0 + i, -4.0, (10.0, 10.0), 72)
            times[mode] = (time.perf_counter() - t0) / 3
            h1, d1 = ctx.reduction_calls()
            plan.close()
            if mode == "hooked":
                assert h1 == h0 and d1 > d0
        assert res["plain"]["vshift"] == res["hooked"]["vshift"]
        assert np.array_equal(res["plain"]["medians"], res["hooked"]["medians"], equal_nan=True)
        print(f"step plain {times['plain'] * 1e3:.2f} ms, through the device-side hook {times['hooked'] * 1e3:.2f} ms")
        assert times["hooked"] < 1.25 * times["plain"] + 0.5e-3
    finally:
        _lib.default_context().set_option("nk_fused", 1)
        if created:
            dist.destroy_process_group()


def test_onepass_step_hands_degenerate_rasters_to_the_plain_route(coreg):
    """Round 4: the one-pass step stages the candidates of the median of dh in small per-wave segments (they are ~0.7 % of the
    pixels on real pairs).  A pair whose dh is ONE value -- tba = ref + constant at shift 0: every pixel lies inside the bracket
    of the median -- overruns them by design: the step must notice (overflow flag), fall through to the plain route and return
    exactly what plain digit passes return (steps 1 and 3; step 2 shifts tba by whole pixels: dh varies, any route may answer)."""
    from xdem_amd.synth import fbm_numpy

    ctx = coreg._lib.default_context()
    H, W, res = 2304, 2200, 10.0
    ref = fbm_numpy((H, W), seed=31, std=180.0)
    tba = (ref + np.float32(1.25)).astype(np.float32)
    tba[100:140, 300:900] = np.nan
    got = {}
    try:
        for name, mode, fused in (("onepass", 3, 1), ("plain", 1, 0)):
            ctx.set_option("selection", mode)
            ctx.set_option("nk_fused", fused)
            plan = coreg.NKPlan(ref, tba, None)
            got[name] = [plan.step(sx, sy, (res, res), 8) for sx, sy in ((0.0, 0.0), (10.0, -20.0), (0.0, 0.0))]
            rc = plan.route_counts()
            plan.close()
            if name == "onepass":
                assert rc["onepass"] <= 1 and rc["plain"] >= 2, rc
    finally:
        ctx.set_option("selection", 0)
        ctx.set_option("nk_fused", 1)
    for a, b in zip(got["onepass"], got["plain"]):
        assert a["vshift"] == b["vshift"] and a["n_valid"] == b["n_valid"]
        assert np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["medians"], b["medians"], equal_nan=True)
        assert np.array_equal(a["edges"], b["edges"])


def test_onepass_step_on_a_hooked_plan():
    """Round 5, second half: plans with a reduction hook (the partitioned layout's) take the ONE-PASS step as well -- one data pass over
    the rank's rows and TEN all-reduces per step, all of them enqueued through the device-side hook -- instead of the plain route
    (stored dh, a reduction per key digit).  On a 1-rank RCCL group: every integer output identical to the hook-less plan's for fractional steps, the
    aligned pair (ties en masse in dh) included; route and reduction counts asserted; the whole fit stays on the route."""
    import os
    import sys
    import time

    import scipy.optimize
    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from xdem_amd import _lib, coreg

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29619")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    try:
        ref, tba = bench._c3_pair(dev, 12000)
        steps = ((0.0, 0.0), (1.7, 0.6), (1.2, 0.9), (-17.0, -6.0), (-16.9998, -5.9996))
        res, times = {}, {}
        for mode in ("plain", "hooked", "hooked_plain"):   # ("plain" here: the hook-less plan; "hooked_plain": the plain ROUTE under the hook)
            ctx.set_option("nk_fused_dist", 0 if mode == "hooked_plain" else 1)
            plan = coreg.NKPlan(ref, tba, None, ctx, group=None if mode == "plain" else "world")
            plan.step(0.3, 0.1, (10.0, 10.0), 72)   # (route agreement, buffers, bin cache)
            h0, d0 = ctx.reduction_calls()
            r0 = plan.route_counts()
            t0 = time.perf_counter()
            res[mode] = [plan.step(sx, sy, (10.0, 10.0), 72) for (sx, sy) in steps]
            times[mode] = (time.perf_counter() - t0) / len(steps)
            h1, d1 = ctx.reduction_calls()
            r1 = plan.route_counts()
            if mode == "hooked":
                assert r1["onepass"] - r0["onepass"] == len(steps) and r1["plain"] == r0["plain"], (r0, r1)
                # ten all-reduces per step with sampled brackets, FIVE with predicted ones (round 6: the last step moves the aligned
                # pair by 4e-5 px -- its brackets come from the step before it: no sample selections, no histogram exchanges)
                # (... and EIGHT where only the bracket of the median of dh is predicted: its three histogram exchanges go, the EXT slots
                #  get an exchange of their own)
                n_pred, n_pd = r1["predicted"] - r0["predicted"], r1["predicted_dh_only"] - r0["predicted_dh_only"]
                assert n_pred >= 1 and r1["predict_missed"] == r0["predict_missed"], (r0, r1)
                assert h1 == h0 and d1 - d0 == 10 * (len(steps) - n_pred - n_pd) + 8 * n_pd + 5 * n_pred, (h0, h1, d0, d1, n_pred, n_pd)
                off = coreg._iterate(plan, (10.0, 10.0), 0.0, 8, 72, scipy.optimize.curve_fit, True)
                r2 = plan.route_counts()
                assert r2["plain"] == r1["plain"] and r2["onepass"] == r1["onepass"] + 8, (r1, r2)
                assert abs(off[0] + 17.0) < 0.05 and abs(off[1] + 6.0) < 0.05 and abs(off[2] + 2.0) < 0.01, off
            elif mode == "hooked_plain":
                assert r1["plain"] - r0["plain"] == len(steps) and r1["onepass"] == r0["onepass"], (r0, r1)
                red_plain = (d1 - d0) / len(steps)
            else:
                assert r1["onepass"] - r0["onepass"] == len(steps), (r0, r1)
            plan.close()
        for a, b, c in zip(res["hooked"], res["plain"], res["hooked_plain"]):
            for o in (b, c):
                assert a["n_valid"] == o["n_valid"] and a["vshift"] == o["vshift"]
                assert np.array_equal(a["counts"], o["counts"]) and np.array_equal(a["medians"], o["medians"], equal_nan=True)
                assert np.array_equal(a["edges"], o["edges"])
            assert _moments_close(a, b, onepass=True)
        print(f"12000^2 step: hook-less {times['plain'] * 1e3:.2f} ms | hooked one-pass (10 reductions) {times['hooked'] * 1e3:.2f} ms | "
              f"hooked plain route ({red_plain:.0f} reductions) {times['hooked_plain'] * 1e3:.2f} ms")
        assert times["hooked"] < times["hooked_plain"]
    finally:
        ctx.set_allreduce(None)
        ctx.close()
        if created:
            dist.destroy_process_group()


def test_onepass_step_on_a_hooked_plan_float64():
    """The same route for float64 rasters (64-bit keys in the exchanged key lists, 8-byte values in the per-bin exchange): 6000^2 pair,
    18 bins, 1-rank RCCL group -- identical to the hook-less plan, ten reductions per step."""
    import os
    import sys

    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from xdem_amd import _lib, coreg

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29621")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    ctx = _lib.Context(0)
    try:
        ref, tba = bench._c3_pair(torch.device("cuda", 0), 6000)
        ref, tba = ref.double().contiguous(), tba.double().contiguous()
        steps = ((0.0, 0.0), (1.7, 0.6), (-17.0, -6.0), (-16.9998, -5.9996))   # (the last one: a settled step, predicted brackets)
        res = {}
        for mode in ("plain", "hooked"):
            plan = coreg.NKPlan(ref, tba, None, ctx, group=None if mode == "plain" else "world")
            plan.step(0.3, 0.1, (10.0, 10.0), 18)
            h0, d0 = ctx.reduction_calls()
            r0 = plan.route_counts()
            res[mode] = [plan.step(sx, sy, (10.0, 10.0), 18) for (sx, sy) in steps]
            h1, d1 = ctx.reduction_calls()
            r1 = plan.route_counts()
            assert r1["onepass"] - r0["onepass"] == len(steps) and r1["plain"] == r0["plain"], (mode, r0, r1)
            if mode == "hooked":
                # ten all-reduces per step with sampled brackets, FIVE with predicted ones (round 6: the last step moves the aligned
                # pair by 4e-5 px -- its brackets come from the step before it: no sample selections, no histogram exchanges)
                # (... and EIGHT where only the bracket of the median of dh is predicted: its three histogram exchanges go, the EXT slots
                #  get an exchange of their own)
                n_pred, n_pd = r1["predicted"] - r0["predicted"], r1["predicted_dh_only"] - r0["predicted_dh_only"]
                assert n_pred >= 1 and r1["predict_missed"] == r0["predict_missed"], (r0, r1)
                assert h1 == h0 and d1 - d0 == 10 * (len(steps) - n_pred - n_pd) + 8 * n_pd + 5 * n_pred, (h0, h1, d0, d1, n_pred, n_pd)
            plan.close()
        for a, b in zip(res["hooked"], res["plain"]):
            assert a["n_valid"] == b["n_valid"] and a["vshift"] == b["vshift"]
            assert np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["medians"], b["medians"], equal_nan=True)
            assert np.array_equal(a["edges"], b["edges"])
    finally:
        ctx.set_allreduce(None)
        ctx.close()
        if created:
            dist.destroy_process_group()
