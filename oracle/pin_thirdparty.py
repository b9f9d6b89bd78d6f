"""Pins for the two third-party boundaries that nothing readable offline fixes -- run wherever geoutils and scikit-gstat
are importable (they are un-vendored dependencies of the reference, absent from the build image):

    python oracle/pin_thirdparty.py            # writes tests/golden/thirdparty_interp.npz / thirdparty_skgstat.npz

TEST INFRASTRUCTURE (like everything under oracle/).  The script records INPUTS and the packages' OUTPUTS only:

* geoutils ``_interp_points`` exactly as the reference calls it from the Nuth-Kaab loop (xdem/coreg/affine.py:172-184 via
  xdem/coreg/base.py:1642-1649): a small DEM with NaN holes, sampled at the shifted pixel centres for integer, half-pixel
  and generic shifts -> which of the product's ``nk_nan_rule`` conventions (0 "4tap", 1 "weighted", 2 "dilate3x3") is
  geoutils' is then decided by tests/test_thirdparty_pins.py, which compares every rule of the ORACLE with the recording;
* scikit-gstat ``Variogram`` exactly as xdem/spatialstats.py:1091, 1247-1255 construct it: lattice points whose pair
  distances fall exactly on the bin edges (3-4-5 triangles) -> edge inclusivity ("vario_edge"), float32 values with large
  offsets -> where |dv| is formed ("vario_diff"), estimator constants (matheron / cressie / dowd), and
  ``RasterEquidistantMetricSpace`` on a small grid -> which pixel pairs a run contains (centre disk x rings, or also the
  disk with itself: ADVICE.md round 1).

* geoutils ``subsample_array`` -- the draw behind NuthKaab's default 5e5-point subsample (xdem/coreg/base.py:600-605) and
  behind the variogram samplers (xdem/spatialstats.py:978): which indexes does it return for a given ``random_state``,
  array (with NaNs / a mask) and ``subsample`` (fraction and count)?  The product restates its published rule
  (``coreg.subsample_valid_mask``: ``default_rng(random_state).choice(valid flat indexes, n, replace=False)``); the recording
  settles the RNG protocol (generator type, order of the draws, sorted or not).

Nothing here is imported by the product; the fixtures are data.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def pin_interp() -> str | None:
    try:
        import geoutils as gu
        from geoutils.raster.georeferencing import _coords
        from geoutils.raster.interpolate import _interp_points
        import affine
    except Exception as e:  # pragma: no cover - absent in the build image
        print("geoutils not importable:", e)
        return None
    rng = np.random.default_rng(7)
    dem = (100 + np.cumsum(np.cumsum(rng.normal(size=(24, 31)), 0), 1)).astype(np.float32)
    dem[5, 6] = np.nan
    dem[11:13, 20:23] = np.nan
    dem[23, 30] = np.nan
    res = 10.0
    transform = affine.Affine(res, 0, 500000.0, 0, -res, 7000000.0)
    # the coordinates and the interpolator the reference builds (affine.py:171-184)
    xx, yy = _coords(transform=transform, shape=dem.shape, area_or_point=None, grid=True)
    out = {"dem": dem, "res": np.float64(res), "geoutils_version": np.array(gu.__version__)}
    for k, (sx, sy) in enumerate([(0.0, 0.0), (10.0, -20.0), (5.0, 5.0), (3.3, -7.1), (-17.0, 0.0), (0.0, 2.5)]):
        vals = _interp_points(dem, transform=transform, points=(xx + sx, yy + sy), method="linear", area_or_point=None,
                              shift_area_or_point=True)
        out[f"shift{k}"] = np.array([sx, sy])
        out[f"vals{k}"] = np.asarray(vals).reshape(dem.shape)
    path = os.path.join(GOLDEN, "thirdparty_interp.npz")
    np.savez_compressed(path, **out)
    return path


def pin_skgstat() -> str | None:
    try:
        import skgstat as skg
    except Exception as e:  # pragma: no cover
        print("scikit-gstat not importable:", e)
        return None
    out = {"skgstat_version": np.array(skg.__version__)}
    # (1) edge inclusivity + estimator constants: lattice points, edges on exact pair distances
    ix, iy = np.meshgrid(np.arange(12), np.arange(9))
    coords = np.column_stack([ix.ravel(), iy.ravel()]).astype(np.float64) * 2.0
    rng = np.random.default_rng(3)
    values = (np.sin(coords[:, 0] / 5.0) * 1000.0 + rng.normal(size=coords.shape[0])).astype(np.float32)
    edges = [2.0, 4.0, 10.0, 20.0, 26.0]
    out["coords"], out["values"], out["edges"] = coords, values, np.array(edges)
    for est in ("matheron", "cressie", "dowd"):
        for vdt in (np.float32, np.float64):
            V = skg.Variogram(coords, values.astype(vdt), normalize=False, fit_method=None, bin_func=edges, maxlag=edges[-1], estimator=est)
            bins, exp = V.get_empirical(bin_center=False)
            out[f"exp_{est}_{np.dtype(vdt).name}"] = np.asarray(exp, dtype=np.float64)
            out[f"count_{est}_{np.dtype(vdt).name}"] = np.asarray(V.bin_count, dtype=np.int64)
            out[f"bins_{est}_{np.dtype(vdt).name}"] = np.asarray(bins, dtype=np.float64)
    # (2) RasterEquidistantMetricSpace: which pairs does one run hold?
    shape, gsd = (40, 50), 1.0
    x, y = np.meshgrid(np.arange(0, shape[0] * gsd, gsd), np.arange(0, shape[1] * gsd, gsd))
    gc = np.dstack((x.flatten(), y.flatten())).squeeze()
    extent = (gc[:, 0].min(), gc[:, 0].max(), gc[:, 1].min(), gc[:, 1].max())
    M = skg.RasterEquidistantMetricSpace(gc, shape=shape, extent=extent, samples=12, ratio_subsample=0.05, runs=3, rnd=np.random.default_rng(11))
    D = M.dists.tocoo()
    out["rems_rows"], out["rems_cols"], out["rems_dists"] = D.row.astype(np.int64), D.col.astype(np.int64), D.data.astype(np.float64)
    out["rems_shape"], out["rems_coords"] = np.array(shape), gc
    out["rems_samples"], out["rems_ratio"], out["rems_runs"] = np.int64(12), np.float64(0.05), np.int64(3)
    for name, attr in (("rems_centers", "_centers"), ("rems_center_radius", "_center_radius"), ("rems_radii", "equidistant_radii"),
                       ("rems_max_dist", "_max_dist")):
        if getattr(M, attr, None) is not None:
            out[name] = np.asarray(getattr(M, attr), dtype=np.float64)
    path = os.path.join(GOLDEN, "thirdparty_skgstat.npz")
    np.savez_compressed(path, **out)
    return path


def pin_subsample() -> str | None:
    try:
        import geoutils as gu
        from geoutils.raster import subsample_array
    except Exception as e:  # pragma: no cover - absent in the build image
        print("geoutils.raster.subsample_array not importable:", e)
        return None
    rng = np.random.default_rng(21)
    arr = rng.normal(size=(37, 53)).astype(np.float32)
    arr[3:6, 10:20] = np.nan
    arr[30, 5] = np.nan
    marr = np.ma.masked_array(arr.copy(), mask=rng.uniform(size=arr.shape) < 0.1)
    out = {"arr": arr, "mask": np.ma.getmaskarray(marr), "geoutils_version": np.array(gu.__version__)}
    k = 0
    for name, a in (("nan", arr), ("masked", marr)):
        for subsample in (0.25, 1, 200, 10**6):
            for seed in (42, 7):
                idx = subsample_array(a, subsample=subsample, return_indices=True, random_state=seed)
                out[f"case{k}"] = np.array([name, str(subsample), str(seed)])
                out[f"rows{k}"], out[f"cols{k}"] = (np.asarray(i, dtype=np.int64) for i in idx)
                vals = subsample_array(a, subsample=subsample, return_indices=False, random_state=seed)
                out[f"vals{k}"] = np.asarray(vals, dtype=np.float32)
                k += 1
    # a Generator passed in (the variogram samplers hand one over): state consumed?
    g = np.random.default_rng(5)
    i1 = subsample_array(arr, subsample=50, return_indices=True, random_state=g)
    i2 = subsample_array(arr, subsample=50, return_indices=True, random_state=g)
    out["gen_rows1"], out["gen_cols1"] = (np.asarray(i, dtype=np.int64) for i in i1)
    out["gen_rows2"], out["gen_cols2"] = (np.asarray(i, dtype=np.int64) for i in i2)
    out["n_cases"] = np.int64(k)
    path = os.path.join(GOLDEN, "thirdparty_subsample.npz")
    np.savez_compressed(path, **out)
    return path


def decide() -> str | None:
    """From the recorded fixtures: WHICH of the switchable conventions reproduce the packages -- written to
    xdem_amd/thirdparty_decision.json, which the product reads its defaults of "nk_nan_rule", "vario_edge", "vario_diff" from
    when the file is present (xdem_amd/_lib.py: thirdparty_decision).  The lowest-numbered matching value wins; a convention
    none of the values reproduces is left out (the built-in default stays, and tests/test_thirdparty_pins.py fails)."""
    import json

    sys.path.insert(0, HERE)
    golden = os.path.join(os.path.dirname(HERE), "tests", "golden")
    out = {}
    p_i = os.path.join(golden, "thirdparty_interp.npz")
    if os.path.exists(p_i):
        import nuthkaab_oracle as nko

        z = np.load(p_i)
        dem, res = z["dem"], float(z["res"])
        for rule in (0, 1, 2, 3):
            ok = True
            for k in range(6):
                sx, sy = z[f"shift{k}"]
                got = nko.bilinear_shifted(dem, -sy / res, sx / res, nan_rule=rule)
                ok &= bool(np.array_equal(np.isnan(got), np.isnan(z[f"vals{k}"])) and np.allclose(got, z[f"vals{k}"], rtol=1e-6, atol=0, equal_nan=True))
            if ok:
                out["nk_nan_rule"] = rule
                break
    p_s = os.path.join(golden, "thirdparty_skgstat.npz")
    if os.path.exists(p_s):
        import variogram_oracle as vo

        z = np.load(p_s)
        coords, values, edges = z["coords"], z["values"], [float(e) for e in z["edges"]]
        for right_closed in (False, True):
            for diff_f64 in (False, True):
                ok = True
                for est in ("matheron", "cressie", "dowd"):
                    e, c = vo.empirical_variogram_blocks([(coords[:, 0], coords[:, 1], values)], edges, est, right_closed=right_closed, diff_f64=diff_f64)
                    ok &= bool(np.array_equal(c, z[f"count_{est}_float32"]) and np.allclose(e, z[f"exp_{est}_float32"], rtol=1e-9, equal_nan=True))
                if ok and "vario_edge" not in out:
                    out["vario_edge"], out["vario_diff"] = int(right_closed), int(diff_f64)
    if not out:
        return None
    out["_source"] = "oracle/pin_thirdparty.py: decided from tests/golden/thirdparty_*.npz"
    path = os.path.join(os.path.dirname(HERE), "xdem_amd", "thirdparty_decision.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    return path


if __name__ == "__main__":
    made = [p for p in (pin_interp(), pin_skgstat(), pin_subsample()) if p]
    print("written:", made if made else "nothing (packages absent)")
    print("decision file:", decide() or "none (no fixture to decide from: the built-in defaults stay)")
    sys.exit(0)
