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


def main(ref, out_dir: str) -> None:
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
    np.savez_compressed(os.path.join(out_dir, "binning_golden.npz"), **rec)
    print("binning fixtures written")
