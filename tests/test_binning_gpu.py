"""GPU parity of the N-D binned statistics: xdem_amd.spatialstats.nd_binning (csrc/binstats.hip through the
C-ABI) vs the reference's own DataFrames (tests/golden/binning_golden.npz) and vs the CPU oracle on larger inputs.
Counts, exact medians, NMADs and interval edges: BIT-EXACT."""
import os
import warnings

import numpy as np
import pytest

import binning_oracle as bo
from conftest import GOLDEN
from test_oracle_binning_golden import CASES, RANGE_CASES, flatten_like_reference, load_case, load_range_case

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


@pytest.mark.parametrize("name", RANGE_CASES)
def test_nd_binning_with_list_ranges_equals_reference_dataframe(name):
    """``list_ranges`` (SciPy's ``range=``): the reference's DataFrame for one variable and a (start, stop) pair / a one-element
    list of pairs -- edges from the range, samples outside in no bin, a sample on the last edge in the last bin."""
    from xdem_amd import spatialstats as ss

    z = np.load(os.path.join(GOLDEN, "binning_ranges_golden.npz"))
    values, list_var, bins, ranges = load_range_case(z, name)
    df = ss.nd_binning(values, list_var, ["v0"], list_var_bins=bins, statistics=["count", np.nanmedian, ss.nmad], list_ranges=ranges)
    got = df_to_cols(df, 1)
    for key, arr in got.items():
        ref = np.asarray(z[f"{name}|{key}"], np.float64)
        assert arr.shape == ref.shape and np.array_equal(arr, ref, equal_nan=True), (name, key)


def test_nd_binning_list_ranges_errors_are_scipys():
    """What upstream's ``range=list_ranges`` makes SciPy refuse (several variables, start after stop, wrong length) is refused
    with the same exception and message (tests/golden/binning_ranges_errors.json, recorded from the reference)."""
    import json

    from xdem_amd import spatialstats as ss

    errs = json.load(open(os.path.join(GOLDEN, "binning_ranges_errors.json")))
    x = np.linspace(0, 10, 200).astype(np.float32)
    v = np.sin(x).astype(np.float32)
    calls = {"two_vars_two_pairs": ([x, x[::-1].copy()], (4, 3), [(0.0, 5.0), (0.0, 3.0)]),
             "start_after_stop": ([x], 4, (5.0, 1.0)),
             "three_pairs_one_var": ([x], 4, [(0.0, 1.0), (0.0, 1.0), (0.0, 1.0)])}
    assert set(calls) == set(errs)
    for name, (lv, bins, ranges) in calls.items():
        want = errs[name]
        assert want is not None
        with pytest.raises(Exception) as ei:
            ss.nd_binning(v, lv, [f"v{i}" for i in range(len(lv))], list_var_bins=bins, statistics=["count", np.nanmedian], list_ranges=ranges)
        assert type(ei.value).__name__ == want["type"] and str(ei.value) == want["message"], (name, repr(ei.value), want)


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
    df = ss.nd_binning(v, [v], ["a"], list_var_bins=4, statistics=[np.nanmean])   # (a statistic of the host half: refused until round 6)
    assert list(df.columns) == ["nd", "count", "nanmean", "a"] and df["nanmean"].tolist() == [12.0, 37.0, 62.0, 87.0]
    df = ss.nd_binning(v, [v], ["a"], list_var_bins=4, statistics=[np.nanmedian])  # count is added in front
    assert list(df.columns) == ["nd", "count", "nanmedian", "a"] and df["count"].tolist() == [25.0] * 4
    assert df["nanmedian"].tolist() == [12.0, 37.0, 62.0, 87.0]


def test_heteroscedasticity_equals_reference_error_map():
    """infer_heteroscedasticity_from_stable through the product path vs the error map the reference's own pipeline produced
    (tests/golden/binning_golden.npz, het|*).  Binned table bit-exact; float64 error map within 1e-12 relative (the
    multilinear evaluation order of the GPU kernel follows SciPy's generic path; its 2-D Cython fast path differs in the
    last bits)."""
    from xdem_amd import spatialstats as ss

    z = np.load(os.path.join(GOLDEN, "binning_golden.npz"))
    dh, slope, maxc, stable = z["het|dh"], z["het|slope"], z["het|maxc"], z["het|stable"]
    err, df, fun = ss.infer_heteroscedasticity_from_stable(dh, [slope, maxc], stable_mask=stable, list_var_names=["slope", "maxc"],
                                                           list_var_bins=(8, 6), min_count=30)
    assert np.array_equal(df["nd"].values, z["het|df_nd"])
    assert np.array_equal(df["count"].values, z["het|df_count"])
    assert np.array_equal(df["nmad"].values, z["het|df_nmad"], equal_nan=True)
    assert err.shape == dh.shape and err.dtype == np.float64
    assert np.array_equal(np.isnan(err), np.isnan(z["het|error"]))
    assert np.allclose(err, z["het|error"], rtol=1e-12, atol=0, equal_nan=True)
    probe = (z["het|probe_x"], z["het|probe_y"])
    assert np.allclose(fun(probe), z["het|probe_out"], rtol=1e-12, atol=0, equal_nan=True)
    # the unscaled interpolants of interp_nd_binning, 2-D and 1-D
    f2 = ss.interp_nd_binning(df, ["slope", "maxc"], statistic="nmad", min_count=30)
    assert np.allclose(f2(probe), z["het|interp2_out"], rtol=1e-12, atol=0, equal_nan=True)
    f1 = ss.interp_nd_binning(df, ["slope"], statistic="nmad", min_count=30)
    assert np.allclose(f1((probe[0],)), z["het|interp1_out"], rtol=1e-12, atol=0, equal_nan=True)
    # documented toy example of interp_nd_binning (spatialstats.py:266-290)
    import pandas as pd

    toy = pd.DataFrame({"var1": [1, 2, 3, 1, 2, 3, 1, 2, 3], "var2": [1, 1, 1, 2, 2, 2, 3, 3, 3], "statistic": [1, 2, 3, 4, 5, 6, 7, 8, 9]})
    ft = ss.interp_nd_binning(toy, list_var_names=["var1", "var2"], statistic="statistic", min_count=None)
    assert ft((2, 2)) == 5.0 and ft((1.5, 1.5)) == 3.0 and ft((-1, 1)) == 1.0


def test_heteroscedasticity_large_vs_oracle():
    """4e6-pixel grid, three explanatory variables (3-D binning + trilinear error function)."""
    from xdem_amd import spatialstats as ss

    rng = np.random.default_rng(8)
    shape = (2000, 2000)
    slope = rng.gamma(2.0, 8.0, shape).astype(np.float32)
    maxc = np.abs(rng.normal(0, 1.5, shape)).astype(np.float32)
    qual = rng.uniform(0, 100, shape).astype(np.float32)
    dh = (rng.normal(0, 1, shape) * (0.5 + 0.05 * slope + 0.3 * maxc + 0.01 * qual)).astype(np.float32)
    dh[::17, ::19] = np.nan
    err, df, fun = ss.infer_heteroscedasticity_from_stable(dh, [slope, maxc, qual], list_var_bins=(6, 5, 4))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res, ofun, scale = bo.estimate_model_heteroscedasticity(dh.ravel(), [slope.ravel(), maxc.ravel(), qual.ravel()], (6, 5, 4))
        ref = (scale * ofun((slope, maxc, qual)))
    got = df_to_cols_any(df, ["var1", "var2", "var3"])
    flat = flatten_like_reference(res, 3)
    assert np.array_equal(got["count"], flat["count"]) and np.array_equal(got["nmad"], flat["nmad"], equal_nan=True)
    assert np.allclose(err, ref, rtol=1e-12, atol=0, equal_nan=True)
    # heteroscedastic by construction: the error grows with slope
    assert fun((40.0, 1.0, 50.0)) > fun((5.0, 1.0, 50.0))


def df_to_cols_any(df, names):
    return {c: df[c].values.astype(np.float64) for c in ("count", "nmad")}


def test_nmad_device_matches_numpy():
    from xdem_amd import spatialstats as ss

    rng = np.random.default_rng(0)
    for dt in (np.float32, np.float64):
        v = rng.standard_t(3, 1_000_001).astype(dt)
        v[::1000] = np.nan
        med, nm, cnt = ss.nmad_device(v)
        assert cnt == np.isfinite(v).sum() and med == np.nanmedian(v) and nm == float(bo.nmad(v))
        lim = 5.0
        w = v.copy()
        w[np.abs(w) > lim] = np.nan
        med, nm, cnt = ss.nmad_device(v, abs_limit=lim)
        assert cnt == np.isfinite(w).sum() and med == np.nanmedian(w) and nm == float(bo.nmad(w))


def test_nmad_of_a_large_array_takes_the_device_route_and_returns_numpys_scalar():
    """``spatialstats.nmad`` on raster-sized float arrays goes through ``nmad_device`` (exact selection): the value and the scalar type
    NumPy's expression returns, for float32 / float64, NaNs, a masked array and a non-default ``nfact``; small arrays stay on the host."""
    from xdem_amd import spatialstats as ss

    rng = np.random.default_rng(3)
    for dt in (np.float32, np.float64):
        v = rng.standard_t(3, (2100, 2000)).astype(dt)
        v[::7, ::13] = np.nan
        for nfact in (1.4826, 1.0):
            want = nfact * np.nanmedian(np.abs(v - np.nanmedian(v)))
            got = ss.nmad(v, nfact)
            assert type(got) is type(want) and got == want, (dt, nfact, got, want)
        m = np.ma.masked_array(v, mask=np.isnan(v))
        assert ss.nmad(m) == 1.4826 * np.nanmedian(np.abs(v - np.nanmedian(v)))
    small = rng.normal(size=1000).astype(np.float32)
    assert ss.nmad(small) == 1.4826 * np.nanmedian(np.abs(small - np.nanmedian(small)))


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_selection_modes_agree_on_large_inputs(mode):
    """Bracketed selection (sample -> brackets -> one counting/compaction pass -> candidates; mode 3 forces it for the
    multi-bin cases too, which mode 0 leaves to the plain passes at these sizes), plain radix passes (1) and the
    bracket-miss fall-back (2) give the same exact order statistics: 6e6-sample binning with ties, global NMAD, and a
    Nuth-Kaab step (y and bin ids computed inside the passes on the bracketed route), all against NumPy / the oracle."""
    from xdem_amd import _lib, coreg
    from xdem_amd import spatialstats as ss
    import nuthkaab_oracle as no

    ctx = _lib.default_context()
    ctx.set_option("selection", mode)
    try:
        rng = np.random.default_rng(17)
        n = 6_000_000
        x = rng.gamma(2.0, 8.0, n).astype(np.float32)
        v = np.round(rng.normal(0, 1, n) * (0.5 + 0.05 * x), 2).astype(np.float32)   # many ties
        v[::997] = np.nan
        df = ss.nd_binning(v, [x], ["v0"], list_var_bins=16)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = flatten_like_reference(bo.nd_binning_arrays(v, [x], 16), 1)
        for key in ("count", "nanmedian", "nmad"):
            assert np.array_equal(df[key].values.astype(np.float64), ref[key], equal_nan=True), (mode, key)
        for dt in (np.float32, np.float64):
            w = rng.standard_t(3, 5_000_001).astype(dt)
            med, nm, cnt = ss.nmad_device(w)
            assert med == np.median(w) and nm == float(bo.nmad(w)) and cnt == w.size
        # Nuth-Kaab step on a 2304^2 pair (5.3e6 pixels): global nanmedian + 72 bin medians vs the oracle
        from xdem_amd.synth import fbm_numpy

        m = 2304
        refd = fbm_numpy((m, m), seed=3)
        tba = (np.roll(refd, (1, -1), (0, 1)) + 1.5 + rng.normal(0, 0.3, (m, m))).astype(np.float32)
        tba[rng.uniform(size=(m, m)) < 0.1] = np.nan
        plan = coreg.NKPlan(refd, tba, None, ctx)
        got = plan.step(2.0, -3.0, (10.0, 10.0), 72)
        got6 = plan.step(2.0, -3.0, (10.0, 10.0), 6)  # few bins: per-bin samples large enough for real brackets at this size
        plan.close()
        st, asp = no.aux_vars(refd)
        valid = np.isfinite(refd) & np.isfinite(tba) & np.isfinite(st) & np.isfinite(asp)
        dh = no.shifted_dh(refd, tba, 2.0, -3.0, (10.0, 10.0))[valid]
        vshift = np.nanmedian(dh)
        assert got["vshift"] == float(vshift)
        dh = dh - vshift
        ok = np.isfinite(dh)
        assert got["n_valid"] == int(ok.sum())
        with np.errstate(all="ignore"):
            y = dh[ok] / st[valid][ok]
        edges, counts, med = no.bin_medians(asp[valid][ok], y, 72)
        assert np.array_equal(got["counts"], counts) and np.array_equal(got["medians"], med, equal_nan=True)
        edges, counts, med = no.bin_medians(asp[valid][ok], y, 6)
        assert np.array_equal(got6["counts"], counts) and np.array_equal(got6["medians"], med, equal_nan=True)
        assert np.array_equal(got6["edges"], edges.astype(np.float64))
    finally:
        ctx.set_option("selection", 0)


def test_dem_estimate_uncertainty_end_to_end():
    """All three hot paths in the caller the reference builds on them (xdem/dem.py:667-780): terrain attributes ->
    N-D binning + error function -> standardized Dowd variogram -> model fit.  Synthetic pair whose error grows with slope
    and is spatially correlated over ~12 pixels: the error map must track the construction, the correlation must decay."""
    from scipy.ndimage import gaussian_filter

    from xdem_amd.dem import DEM
    from xdem_amd.synth import fbm_numpy

    n, res = 1200, 10.0
    ref = fbm_numpy((n, n), seed=1, std=400.0)
    d = DEM(ref, (res, 0, 0, 0, -res, 0))
    slope = d.slope().data
    rng = np.random.default_rng(2)
    noise = gaussian_filter(rng.normal(size=(n, n)), 4.0)
    noise = noise / noise.std()
    sigma_true = 0.5 + 0.08 * np.nan_to_num(slope, nan=0.0)
    other = DEM((ref + sigma_true * noise).astype(np.float32), d.transform)
    stable = rng.uniform(size=(n, n)) < 0.8
    sig, corr = d.estimate_uncertainty(other, stable_terrain=stable, random_state=42)
    assert sig.data.shape == (n, n) and sig.data.dtype == np.float64
    ok = np.isfinite(sig.data) & np.isfinite(slope)
    # the error map follows the constructed heteroscedasticity (within binning / interpolation accuracy)
    lo, hi = ok & (slope < 5), ok & (slope > 25)
    assert sig.data[hi].mean() > 1.8 * sig.data[lo].mean()
    rel = np.abs(sig.data[ok] - sigma_true[ok]) / sigma_true[ok]
    assert np.median(rel) < 0.15
    # correlation: 1 at lag 0, substantial inside the correlation length, gone far beyond it
    c = corr(np.array([0.0, 2 * res, 400 * res]))
    assert c[0] == pytest.approx(1.0) and 0.3 < c[1] <= 1.0 and c[2] < 0.2
    # constant-error approaches run too
    sig_b, corr_b = d.estimate_uncertainty(other, stable_terrain=stable, approach="Basic", list_vario_models="spherical", random_state=1)
    assert np.ptp(sig_b.data) == 0 and 0 <= corr_b(np.array([50.0]))[0] <= 1


def test_randomised_binnings_vs_oracle():
    """Seeded sweep over nd_binning: 1-4 variables of mixed dtype, integer or explicit-edge bins, constant variables, NaN / Inf
    rows, heavy ties, tiny and empty bins -- every column of every 1-D / 2-D / N-D block bit-exact against the oracle."""
    from xdem_amd import spatialstats as ss

    rng = np.random.default_rng(4242)
    for trial in range(25):
        n = int(rng.choice([50, 999, 20000, 150000]))
        nv = int(rng.integers(1, 5))
        vdt = rng.choice([np.float32, np.float64])
        values = np.round(rng.normal(0, 2, n), int(rng.integers(0, 4))).astype(vdt)
        values[rng.uniform(size=n) < 0.02] = np.nan
        list_var, bins = [], []
        for k in range(nv):
            dt = rng.choice([np.float32, np.float64])
            kind = rng.integers(0, 4)
            if kind == 0:
                var = rng.gamma(2.0, 5.0, n)
            elif kind == 1:
                var = rng.integers(0, 7, n).astype(np.float64)      # few distinct values, many on bin edges
            elif kind == 2:
                var = np.full(n, 3.25)                               # constant: SciPy widens the range by +-0.5
            else:
                var = rng.normal(100, 30, n)
            var = var.astype(dt)
            if rng.uniform() < 0.5:
                var[rng.integers(0, n)] = rng.choice([np.nan, np.inf])
            list_var.append(var)
            if rng.uniform() < 0.7 or kind == 2:
                bins.append(int(rng.integers(1, 9)))
            else:
                lo, hi = np.nanmin(var[np.isfinite(var)]), np.nanmax(var[np.isfinite(var)])
                bins.append(np.unique(np.round(np.linspace(lo - 1, hi + 0.5, int(rng.integers(2, 8))), 2)))
        names = [f"v{i}" for i in range(nv)]
        df = ss.nd_binning(values, list_var, names, list_var_bins=tuple(bins))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = flatten_like_reference(bo.nd_binning_arrays(values, list_var, tuple(bins)), nv)
        got = df_to_cols(df, nv)
        for key, arr in got.items():
            assert np.array_equal(arr, np.asarray(ref[key], np.float64), equal_nan=True), (trial, key, n, nv, bins)


@pytest.mark.parametrize("name", ["stats_f32_1var", "stats_f64_2var", "stats_f32_3var"])
def test_nd_binning_with_any_statistic_equals_reference_dataframe(name):
    """`statistics` beyond count / nanmedian / nmad (tests/golden/binning_stats_golden.npz: the reference's DataFrames for SciPy's
    names, NumPy function objects and plain callables): the device produces the bin numbers, the host applies the statistic per
    bin as scipy.stats.binned_statistic_dd does -- every column bit for bit, columns in upstream's order."""
    def p90(a):
        return np.percentile(a, 90)

    def spread(a):
        return float(np.max(a) - np.min(a)) if len(a) else -1.0

    from xdem_amd import spatialstats as ss

    z = np.load(os.path.join(GOLDEN, "binning_stats_golden.npz"))
    values = z[f"{name}|values"]
    list_var = [z[f"{name}|var{i}"] for i in range(int(name[-4]))]
    bins = z[f"{name}|bins"]
    bins = int(bins) if bins.ndim == 0 else tuple(int(b) for b in bins)
    STAT_LIST = ["count", np.nanmedian, ss.nmad, np.nanmean, np.nanstd, "mean", "std", np.sum, "min", np.max, "median", p90, spread]   # (oracle/gen_golden_binning.py)
    stats = STAT_LIST
    names = [f"v{i}" for i in range(len(list_var))]
    df = ss.nd_binning(values, list_var, names, list_var_bins=bins, statistics=stats)
    assert list(df.columns) == [str(c) for c in z[f"{name}|columns"]]
    assert np.array_equal(df["nd"].values.astype(np.int64), z[f"{name}|nd"])
    for f in STAT_LIST:
        col = f if isinstance(f, str) else f.__name__
        got, ref = df[col].values.astype(np.float64), z[f"{name}|{col}"]
        assert got.shape == ref.shape and np.array_equal(got, ref, equal_nan=True), (name, col, np.flatnonzero(~np.isclose(got, ref, rtol=0, atol=0, equal_nan=True))[:5])
    with pytest.raises(ValueError, match="invalid statistic 'mode'"):
        ss.nd_binning(values, list_var, names, list_var_bins=bins, statistics=["count", "mode"])


def test_heteroscedasticity_under_another_spread_statistic():
    """`spread_statistic=np.nanstd` through the whole pipeline (xdem/spatialstats.py:576-631): binned table (host half on the device's
    bin numbers) bit for bit, error function within 1e-12 relative of the reference's -- refused until the end of round 6."""
    from xdem_amd import spatialstats as ss

    z = np.load(os.path.join(GOLDEN, "binning_stats_golden.npz"))
    dh, slope, maxc = z["hetstd|dh"], z["hetstd|slope"], z["hetstd|maxc"]
    df, fun = ss._estimate_model_heteroscedasticity(dh.ravel(), [slope.ravel(), maxc.ravel()], ["slope", "maxc"], spread_statistic=np.nanstd,
                                                    list_var_bins=(6, 5), min_count=20)
    assert np.array_equal(df["count"].values, z["hetstd|df_count"])
    assert np.array_equal(df["nanstd"].values, z["hetstd|df_nanstd"], equal_nan=True)
    probe = (z["hetstd|probe_x"], z["hetstd|probe_y"])
    assert np.allclose(fun(probe), z["hetstd|probe_out"], rtol=1e-12, atol=0, equal_nan=True)
    err = fun((slope, maxc))
    assert np.array_equal(np.isnan(err), np.isnan(z["hetstd|error"])) and np.allclose(err, z["hetstd|error"], rtol=1e-12, atol=0, equal_nan=True)
