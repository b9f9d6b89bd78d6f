"""Golden vectors of the generic convolution and of the per-bin lookup (8f row f3): runs the reference's
own ``xdem.spatialstats.convolution`` -- SciPy engine, and its Numba engine through the identity-njit shim of _refimport.py --
and ``xdem.spatialstats.get_perbin_nd_binning`` (imported from /root/reference) on seeded inputs and records inputs + outputs
under tests/golden/conv_golden.npz (+ conv_errors.json: what the reference raises).  Container-only; re-run with
python oracle/gen_golden.py conv"""
from __future__ import annotations

import json
import os

import numpy as np


def nmad(data, nfact: float = 1.4826):
    arr = np.asarray(data)
    return nfact * np.nanmedian(np.abs(arr - np.nanmedian(arr)))


def conv_cases():
    rng = np.random.default_rng(2024)
    base = 1000.0 + np.cumsum(np.cumsum(rng.normal(scale=0.3, size=(3, 37, 45)), axis=1), axis=2)
    for dt in (np.float32, np.float64):
        imgs = base.astype(dt)
        imgs[0, 5:7, 9] = np.nan
        imgs[1, 20, 30] = np.inf
        imgs[2, 0, 0] = -np.inf
        imgs[2, 36, 44] = np.nan
        name = np.dtype(dt).name
        # the reference's own use: stencil tables divided by a resolution term (surfit.py:1107), several filters per call
        k5 = rng.integers(-3, 4, size=(4, 5, 5)).astype(np.float64) / 35.0
        k5[1, 2, :] = 0.0                 # zero weights: a NaN under them must not spread (SciPy engine)
        k5[3] = 0.0
        k5[3, 0, 4] = 1.0                 # a single off-centre tap: pins flip and origin
        yield f"{name}|5x5", imgs, k5
        k3 = rng.normal(size=(2, 3, 3))
        k3[0, 1, 1] = 1e-17               # below DBL_EPSILON: skipped by SciPy's footprint, multiplied by the Numba loop
        yield f"{name}|3x3", imgs, k3
        yield f"{name}|rect_3x7", imgs[:2], rng.normal(size=(2, 3, 7))
        yield f"{name}|even_4x4", imgs[:1], rng.normal(size=(2, 4, 4))
        yield f"{name}|even_2x5", imgs[1:2], rng.normal(size=(1, 2, 5))
        yield f"{name}|1x1", imgs[:1], np.array([[[2.5]]])
        yield f"{name}|wide_1x9", imgs[:1], rng.normal(size=(1, 1, 9))
    tiny = rng.normal(size=(1, 3, 4)).astype(np.float32)
    yield "tiny|5x5_larger_than_image", tiny, rng.normal(size=(1, 5, 5))


def perbin_inputs():
    rng = np.random.default_rng(404)
    n = 4000
    slope = rng.gamma(2.0, 8.0, n).astype(np.float32)
    curv = np.abs(rng.normal(0, 1.5, n)).astype(np.float32)
    third = rng.uniform(-3, 3, n)
    dh = (rng.normal(0, 1, n) * (0.5 + 0.05 * slope + 0.3 * curv)).astype(np.float32)
    dh[::97] = np.nan
    slope[5::131] = np.nan
    return dh, slope, curv, third


def main(ref, out_dir: str) -> None:
    ss = ref.spatialstats
    rec = {}
    names = []
    for name, imgs, filters in conv_cases():
        names.append(name)
        rec[f"conv|{name}|imgs"], rec[f"conv|{name}|filters"] = imgs, filters
        rec[f"conv|{name}|scipy"] = ss.convolution(imgs, filters, method="scipy")
        with np.errstate(invalid="ignore", over="ignore"):
            rec[f"conv|{name}|numba"] = ss.convolution(imgs, filters, method="numba")
    rec["conv|names"] = np.array(names)
    # ---- per-bin lookup: on nd_binning's own DataFrames (1, 2, 3 variables), evaluated on fresh points incl. values on edges
    dh, slope, curv, third = perbin_inputs()
    stats = ["count", np.nanmedian, nmad]
    # (a lookup over FEWER variables than the DataFrame bins fails upstream -- np.unique meets the NaN cells of the other
    # binnings' rows: recorded with the errors below -- so every DataFrame here is looked up in its full dimension)
    dfs = {"1": ss.nd_binning(dh, [slope], ["slope"], list_var_bins=9, statistics=stats),
           "2": ss.nd_binning(dh, [slope, curv], ["slope", "curv"], list_var_bins=(6, 4), statistics=stats),
           "3": ss.nd_binning(dh, [slope, curv, third], ["slope", "curv", "third"], list_var_bins=(6, 4, 3), statistics=stats)}
    df = dfs["3"]
    rng = np.random.default_rng(7)
    q_slope = rng.gamma(2.0, 9.0, (50, 60)).astype(np.float32)
    q_curv = np.abs(rng.normal(0, 1.7, (50, 60))).astype(np.float32)
    q_third = rng.uniform(-3.5, 3.5, (50, 60))
    q_slope[3, 3] = np.nan
    q_curv[4, 4] = np.inf
    edges_s = np.unique([iv.left for iv in dfs["1"]["slope"].values if hasattr(iv, "left")])
    q_slope[0, : len(edges_s)] = edges_s.astype(np.float32)      # float32 values next to float64 interval ends
    q_slope64 = q_slope.astype(np.float64)
    q_slope64[1, : len(edges_s)] = edges_s                       # exactly on the ends
    for d, frame in dfs.items():
        for c in ("nd", "count", "nanmedian", "nmad"):
            rec[f"perbin|df{d}|{c}"] = frame[c].values.astype(np.float64)
        for v in ("slope", "curv", "third")[: int(d)]:
            rec[f"perbin|df{d}|{v}|left"] = np.array([iv.left if hasattr(iv, "left") else np.nan for iv in frame[v].values], float)
            rec[f"perbin|df{d}|{v}|right"] = np.array([iv.right if hasattr(iv, "right") else np.nan for iv in frame[v].values], float)
    rec["perbin|q_slope"], rec["perbin|q_slope64"], rec["perbin|q_curv"], rec["perbin|q_third"] = q_slope, q_slope64, q_curv, q_third
    runs = {
        "1var_f32": ("1", [q_slope], ["slope"], "nanmedian", 0),
        "1var_f64_min30": ("1", [q_slope64], ["slope"], "nmad", 30),
        "1var_name_as_str": ("1", [q_slope.ravel()], "slope", "nmad", 0),
        "2var": ("2", [q_slope, q_curv], ["slope", "curv"], "nmad", 10),
        "2var_other_order": ("2", [q_curv, q_slope64], ["curv", "slope"], "nanmedian", 0),
        "3var": ("3", [q_slope, q_curv, q_third], ["slope", "curv", "third"], "nmad", 5),
        "3var_min_huge": ("3", [q_slope, q_curv, q_third], ["slope", "curv", "third"], "nmad", 10**6),
    }
    for key, (d, lv, ln, stat, mc) in runs.items():
        rec[f"perbin|{key}|out"] = ss.get_perbin_nd_binning(dfs[d], lv, ln, statistic=stat, min_count=mc)
    # callable statistic (its __name__ names the column), the DataFrame round-tripped through CSV (intervals as text)
    rec["perbin|callable|out"] = ss.get_perbin_nd_binning(dfs["2"], [q_slope, q_curv], ["slope", "curv"], statistic=np.nanmedian, min_count=0)
    import io

    import pandas as pd

    buf = io.StringIO()
    dfs["2"].to_csv(buf, index=False)
    rec["perbin|csv_text"] = np.array(buf.getvalue())
    df_csv = pd.read_csv(io.StringIO(buf.getvalue()))
    rec["perbin|csv|out"] = ss.get_perbin_nd_binning(df_csv, [q_slope, q_curv], ["slope", "curv"], statistic="nmad", min_count=10)
    buf1 = io.StringIO()
    dfs["1"].to_csv(buf1, index=False)
    rec["perbin|csv1_text"] = np.array(buf1.getvalue())
    rec["perbin|csv1_f32|out"] = ss.get_perbin_nd_binning(pd.read_csv(io.StringIO(buf1.getvalue())), [q_slope], ["slope"], statistic="nmad", min_count=0)
    # a hand-made DataFrame with OVERLAPPING intervals: later bins of the walk overwrite earlier ones
    hand = pd.DataFrame({"x": [pd.Interval(0.0, 5.0, closed="left"), pd.Interval(3.0, 8.0, closed="left"), pd.Interval(2.0, 4.0, closed="left")],
                         "count": [10.0, 1.0, 7.0], "val": [1.5, 2.5, 3.5]})
    xq = np.linspace(-1, 9, 41)
    rec["perbin|overlap|x"] = xq
    rec["perbin|overlap|out_min0"] = ss.get_perbin_nd_binning(hand, [xq], ["x"], statistic="val", min_count=0)
    rec["perbin|overlap|out_min5"] = ss.get_perbin_nd_binning(hand, [xq], ["x"], statistic="val", min_count=5)
    # what the reference raises
    errs = {}

    def attempt(label, fn):
        try:
            fn()
            errs[label] = None
        except Exception as e:  # noqa: BLE001 -- whatever upstream raises is the behaviour recorded
            errs[label] = {"type": type(e).__name__, "message": str(e)}

    attempt("fewer_variables_than_the_binning", lambda: ss.get_perbin_nd_binning(df, [q_slope], ["slope"], statistic="nmad"))
    df = dfs["1"]
    attempt("lengths_differ", lambda: ss.get_perbin_nd_binning(df, [q_slope, q_curv], ["slope"], statistic="nmad"))
    attempt("unknown_variable", lambda: ss.get_perbin_nd_binning(df, [q_slope], ["aspect"], statistic="nmad"))
    attempt("unknown_statistic", lambda: ss.get_perbin_nd_binning(df, [q_slope], ["slope"], statistic="mean"))
    attempt("no_count_column", lambda: ss.get_perbin_nd_binning(df.drop(columns="count"), [q_slope], ["slope"], statistic="nmad"))
    attempt("empty_dataframe", lambda: ss.get_perbin_nd_binning(df.iloc[:0], [q_slope], ["slope"], statistic="nmad"))
    attempt("min_count_none", lambda: ss.get_perbin_nd_binning(df, [q_slope], ["slope"], statistic="nmad", min_count=None))
    d2 = dfs["2"][dfs["2"].nd == 2]
    sparse = d2.drop(index=d2["count"].idxmax())       # the 2-D binning over (slope, curv) without the row of its fullest bin
    attempt("bin_without_row", lambda: ss.get_perbin_nd_binning(sparse, [q_slope, q_curv], ["slope", "curv"], statistic="nmad"))
    attempt("method_name", lambda: ss.convolution(np.zeros((1, 4, 4)), np.ones((1, 3, 3)), method="fft"))
    with open(os.path.join(out_dir, "conv_errors.json"), "w") as fh:
        json.dump(errs, fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(out_dir, "conv_golden.npz"), **rec)
    print("convolution / per-bin fixtures written")
