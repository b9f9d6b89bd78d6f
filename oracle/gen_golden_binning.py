"""Golden vectors of the N-D binning: runs the reference's own ``xdem.spatialstats.nd_binning`` (imported
from /root/reference through oracle/_refimport.py) on seeded inputs and records inputs + the statistic columns of its
DataFrame under tests/golden/binning_golden.npz.  Container-only; re-run with  python oracle/gen_golden.py binning
(geoutils is absent: ``nmad`` is passed in as a local function of the published definition)."""
from __future__ import annotations

import os

import numpy as np


def nmad(data, nfact: float = 1.4826):
    arr = np.asarray(data)
    return nfact * np.nanmedian(np.abs(arr - np.nanmedian(arr)))


def cases():
    rng = np.random.default_rng(77)
    n = 6000
    slope = rng.gamma(2.0, 8.0, n).astype(np.float32)
    curv = np.abs(rng.normal(0, 1.5, n)).astype(np.float32)
    third = rng.uniform(-3, 3, n)  # float64: mixed-dtype sample matrix in the 2-D / N-D binnings
    dh = (rng.normal(0, 1, n) * (0.5 + 0.05 * slope + 0.3 * curv)).astype(np.float32)
    dh[::97] = np.nan
    slope[5::131] = np.nan
    third[7::211] = np.inf
    yield "f32_3var_default", dh, [slope, curv, third], None
    yield "f32_2var_bins", dh, [slope, curv], (7, 4)
    yield "f64_1var", dh.astype(np.float64), [slope.astype(np.float64)], 12
    # custom edges, values exactly on edges and beyond the last one; duplicates -> even / odd counts with ties
    x = np.repeat(np.arange(0.0, 10.5, 0.5), 6).astype(np.float32)
    v = np.tile(np.array([1.0, 2.0, 2.0, 5.0, 7.0, 7.0], np.float32), x.size // 6)
    yield "custom_edges", v, [x], (np.array([0.0, 2.0, 2.5, 6.0, 10.0]),)
    yield "constant_var", v, [np.full(x.size, 3.0, np.float32), x], (3, 5)


def range_cases():
    """``list_ranges`` (round 4): what upstream hands to SciPy's ``range=`` -- usable with ONE variable (a (start, stop) pair, or a
    one-element list of pairs); samples outside the range fall into no bin, a sample ON the last edge into the last bin."""
    rng = np.random.default_rng(91)
    n = 5000
    x = rng.uniform(0, 10, n).astype(np.float32)
    x[:40] = np.float32(8.25)      # on the last edge of the first case
    x[40:60] = np.float32(2.0)     # on its first edge
    v = (rng.normal(0, 1, n) * (1 + 0.2 * x)).astype(np.float32)
    v[::53] = np.nan
    yield "range_pair_1var", v, [x], 6, (2.0, 8.25)
    yield "range_list_1var", v.astype(np.float64), [x.astype(np.float64)], 5, [(1.5, 9.0)]
    yield "range_equal_1var", v, [np.round(x)], 3, (3.0, 3.0)
    yield "range_wider_than_data", v, [x], 4, (-5.0, 25.0)


def range_errors():
    x = np.linspace(0, 10, 200).astype(np.float32)
    v = np.sin(x).astype(np.float32)
    yield "two_vars_two_pairs", v, [x, x[::-1].copy()], (4, 3), [(0.0, 5.0), (0.0, 3.0)]
    yield "start_after_stop", v, [x], 4, (5.0, 1.0)
    yield "three_pairs_one_var", v, [x], 4, [(0.0, 1.0), (0.0, 1.0), (0.0, 1.0)]


def p90(a):
    """A statistic SciPy has never heard of (and that fails on an empty bin: SciPy then fills NaN)."""
    return np.percentile(a, 90)


def spread(a):
    return float(np.max(a) - np.min(a)) if len(a) else -1.0   # (answers the empty bin itself: SciPy fills -1)


STAT_LIST = ["count", np.nanmedian, nmad, np.nanmean, np.nanstd, "mean", "std", np.sum, "min", np.max, "median", p90, spread]


def stat_cases():
    """`statistics` beyond the three the device evaluates: SciPy's names, NumPy function objects, plain callables."""
    rng = np.random.default_rng(314)
    n = 3000
    a = rng.gamma(2.0, 8.0, n).astype(np.float32)
    b = np.abs(rng.normal(0, 1.5, n))
    c = rng.uniform(-3, 3, n).astype(np.float32)
    v = (rng.normal(0, 1, n) * (0.5 + 0.05 * a)).astype(np.float32)
    v[::89] = np.nan
    a[3::101] = np.nan
    yield "stats_f32_1var", v, [a], 7
    yield "stats_f64_2var", v.astype(np.float64), [a, b], (5, 3)
    yield "stats_f32_3var", v, [a, b, c], (4, 3, 2)


def main(ref, out_dir: str) -> None:
    import json

    srec = {}
    for name, values, list_var, bins in stat_cases():
        names = [f"v{i}" for i in range(len(list_var))]
        df = ref.spatialstats.nd_binning(values, list_var, names, list_var_bins=bins, statistics=STAT_LIST)
        srec[f"{name}|values"] = values
        for i, v in enumerate(list_var):
            srec[f"{name}|var{i}"] = v
        srec[f"{name}|bins"] = np.array(bins)
        srec[f"{name}|nd"] = df["nd"].values.astype(np.int64)
        srec[f"{name}|columns"] = np.array(list(df.columns))
        for f in STAT_LIST:
            col = f if isinstance(f, str) else f.__name__
            srec[f"{name}|{col}"] = df[col].values.astype(np.float64)
    # the heteroscedasticity pipeline under a spread statistic other than the NMAD (spatialstats.py:576-631 with spread_statistic=np.nanstd):
    # binned table, interpolant, two-step standardization -- the error function on probe points and on the rasters
    rng = np.random.default_rng(321)
    shape = (90, 100)
    slope = rng.gamma(2.0, 8.0, shape).astype(np.float32)
    maxc = np.abs(rng.normal(0, 1.5, shape)).astype(np.float32)
    dh = (rng.normal(0, 1, shape) * (0.5 + 0.05 * slope + 0.3 * maxc)).astype(np.float32)
    dh[::11, ::7] = np.nan
    df, fun = ref.spatialstats._estimate_model_heteroscedasticity(dvalues=dh.ravel(), list_var=[slope.ravel(), maxc.ravel()], list_var_names=["slope", "maxc"],
                                                                  spread_statistic=np.nanstd, list_var_bins=(6, 5), min_count=20)
    probe = (np.array([-5.0, 0.0, 3.3, 20.0, 55.5, 500.0, np.nan, 10.0]), np.array([0.1, -2.0, 1.7, 9.0, 0.5, 3.0, 1.0, np.nan]))
    srec["hetstd|dh"], srec["hetstd|slope"], srec["hetstd|maxc"] = dh, slope, maxc
    srec["hetstd|df_nanstd"] = df["nanstd"].values.astype(np.float64)
    srec["hetstd|df_count"] = df["count"].values.astype(np.float64)
    srec["hetstd|probe_x"], srec["hetstd|probe_y"] = probe
    srec["hetstd|probe_out"] = np.asarray(fun(probe), np.float64)
    srec["hetstd|error"] = np.asarray(fun((slope, maxc)), np.float64)
    np.savez_compressed(os.path.join(out_dir, "binning_stats_golden.npz"), **srec)

    rrec, errs = {}, {}
    for name, values, list_var, bins, ranges in range_cases():
        names = [f"v{i}" for i in range(len(list_var))]
        df = ref.spatialstats.nd_binning(values, list_var, names, list_var_bins=bins, statistics=["count", np.nanmedian, nmad], list_ranges=ranges)
        rrec[f"{name}|values"] = values
        for i, v in enumerate(list_var):
            rrec[f"{name}|var{i}"] = v
        rrec[f"{name}|bins"] = np.array(bins)
        rrec[f"{name}|ranges"] = np.asarray(ranges, float)
        rrec[f"{name}|ranges_is_list"] = np.array(isinstance(ranges, list))
        rrec[f"{name}|nd"] = df["nd"].values.astype(np.int64)
        for col in ("count", "nanmedian", "nmad"):
            rrec[f"{name}|{col}"] = df[col].values.astype(np.float64)
        for nm in names:
            rrec[f"{name}|{nm}|left"] = np.array([iv.left for iv in df[nm].values], float)
            rrec[f"{name}|{nm}|right"] = np.array([iv.right for iv in df[nm].values], float)
    for name, values, list_var, bins, ranges in range_errors():
        try:
            ref.spatialstats.nd_binning(values, list_var, [f"v{i}" for i in range(len(list_var))], list_var_bins=bins,
                                        statistics=["count", np.nanmedian], list_ranges=ranges)
            errs[name] = None
        except Exception as e:  # noqa: BLE001 -- whatever SciPy raises is the behaviour to reproduce
            errs[name] = {"type": type(e).__name__, "message": str(e)}
    np.savez_compressed(os.path.join(out_dir, "binning_ranges_golden.npz"), **rrec)
    with open(os.path.join(out_dir, "binning_ranges_errors.json"), "w") as fh:
        json.dump(errs, fh, indent=1, sort_keys=True)
    rec = {}
    for name, values, list_var, bins in cases():
        names = [f"v{i}" for i in range(len(list_var))]
        df = ref.spatialstats.nd_binning(values, list_var, names, list_var_bins=bins, statistics=["count", np.nanmedian, nmad])
        rec[f"{name}|values"] = values
        for i, v in enumerate(list_var):
            rec[f"{name}|var{i}"] = v
        rec[f"{name}|bins"] = np.array(-1) if bins is None else (np.array(bins) if np.isscalar(bins) or all(np.isscalar(b) for b in bins)
                                                                 else np.concatenate([np.asarray(b, float) for b in bins]))
        rec[f"{name}|nd"] = df["nd"].values.astype(np.int64)
        for col in ("count", "nanmedian", "nmad"):
            rec[f"{name}|{col}"] = df[col].values.astype(np.float64)
        for nm in names:
            left = np.array([iv.left if hasattr(iv, "left") else np.nan for iv in df[nm].values], float)
            right = np.array([iv.right if hasattr(iv, "right") else np.nan for iv in df[nm].values], float)
            rec[f"{name}|{nm}|left"], rec[f"{name}|{nm}|right"] = left, right
    # heteroscedasticity inference (spatialstats.py:576-631, 808-878): error map from the reference's own pipeline
    rng = np.random.default_rng(123)
    shape = (150, 160)
    slope = rng.gamma(2.0, 8.0, shape).astype(np.float32)
    maxc = np.abs(rng.normal(0, 1.5, shape)).astype(np.float32)
    dh = (rng.normal(0, 1, shape) * (0.5 + 0.05 * slope + 0.3 * maxc)).astype(np.float32)
    dh[::13, ::7] = np.nan
    slope[5, 5] = np.nan
    stable = rng.uniform(size=shape) < 0.7
    # (infer_heteroscedasticity_from_stable itself needs a geoutils Raster to learn a ground sampling distance it never
    # uses, spatialstats.py:707-719; its body after the masking is reproduced call by call: 856-868)
    df, fun = ref.spatialstats._estimate_model_heteroscedasticity(
        dvalues=dh[stable], list_var=[slope[stable], maxc[stable]], list_var_names=["slope", "maxc"], spread_statistic=nmad,
        list_var_bins=(8, 6), min_count=30)
    err = fun((slope, maxc))
    rec["het|dh"], rec["het|slope"], rec["het|maxc"], rec["het|stable"] = dh, slope, maxc, stable
    rec["het|error"] = np.asarray(err, np.float64)
    rec["het|df_nd"] = df["nd"].values.astype(np.int64)
    rec["het|df_count"] = df["count"].values.astype(np.float64)
    rec["het|df_nmad"] = df["nmad"].values.astype(np.float64)
    probe = (np.array([-5.0, 0.0, 3.3, 20.0, 55.5, 500.0, np.nan, 10.0]), np.array([0.1, -2.0, 1.7, 9.0, 0.5, 3.0, 1.0, np.nan]))
    rec["het|probe_x"], rec["het|probe_y"] = probe
    rec["het|probe_out"] = np.asarray(fun(probe), np.float64)
    # the unscaled interpolant alone, 1-D and 2-D (interp_nd_binning, spatialstats.py:237-421)
    f2 = ref.spatialstats.interp_nd_binning(df, ["slope", "maxc"], statistic="nmad", min_count=30)
    rec["het|interp2_out"] = np.asarray(f2(probe), np.float64)
    f1 = ref.spatialstats.interp_nd_binning(df, ["slope"], statistic="nmad", min_count=30)
    rec["het|interp1_out"] = np.asarray(f1((probe[0],)), np.float64)
    np.savez_compressed(os.path.join(out_dir, "binning_golden.npz"), **rec)
    print("binning fixtures written")
