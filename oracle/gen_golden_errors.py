"""Records how the reference's terrain entry point reacts to invalid arguments (exception type and message) so that the
mirror's validation can be pinned without the reference.  Data only.  Container-only:  python oracle/gen_golden_errors.py"""
from __future__ import annotations

import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "terrain_errors.json")

# (function name, kwargs) -- evaluated on a 12x14 float32 ramp; "DEM" marks where the array goes
CASES = [
    ("get_terrain_attribute", {"attribute": "slope"}),                                   # no resolution
    ("get_terrain_attribute", {"attribute": "slope", "resolution": (1.0, 2.0)}),       # non-square resolution
    ("get_terrain_attribute", {"attribute": "not_an_attribute", "resolution": 1.0}),
    ("get_terrain_attribute", {"attribute": ["slope", "nope"], "resolution": 1.0}),
    ("get_terrain_attribute", {"attribute": "slope", "resolution": 1.0, "surface_fit": "Hornn"}),
    ("get_terrain_attribute", {"attribute": "slope", "resolution": 1.0, "curv_method": "x"}),
    ("get_terrain_attribute", {"attribute": "terrain_ruggedness_index", "tri_method": "x"}),
    ("get_terrain_attribute", {"attribute": "profile_curvature", "resolution": 1.0, "surface_fit": "Horn"}),
    ("get_terrain_attribute", {"attribute": ["slope", "max_curvature"], "resolution": 1.0, "surface_fit": "Horn"}),
    ("get_terrain_attribute", {"attribute": "hillshade", "resolution": 1.0, "hillshade_azimuth": 361}),
    ("get_terrain_attribute", {"attribute": "hillshade", "resolution": 1.0, "hillshade_azimuth": -1}),
    ("get_terrain_attribute", {"attribute": "hillshade", "resolution": 1.0, "hillshade_altitude": 91}),
    ("get_terrain_attribute", {"attribute": "hillshade", "resolution": 1.0, "hillshade_altitude": -0.5}),
    ("get_terrain_attribute", {"attribute": "hillshade", "resolution": 1.0, "hillshade_z_factor": -1.0}),
    ("get_terrain_attribute", {"attribute": "hillshade", "resolution": 1.0, "hillshade_z_factor": float("inf")}),
    ("get_terrain_attribute", {"attribute": "rugosity"}),
    ("get_terrain_attribute", {"attribute": ["rugosity", "slope"]}),
    ("get_terrain_attribute", {"attribute": "fractal_roughness", "window_size_fractal": 12}),
    ("get_terrain_attribute", {"attribute": "texture_shading", "texture_alpha": 2.5}),
    ("hillshade", {"resolution": 1.0, "azimuth": 400.0}),
    ("hillshade", {"resolution": 1.0, "altitude": 100.0}),
    ("hillshade", {"resolution": 1.0, "z_factor": float("nan")}),
    ("slope", {}),
    ("terrain_ruggedness_index", {"method": "Rileyy"}),
    ("texture_shading", {"alpha": -0.1}),
]


OUT_SS = os.path.join(os.path.dirname(HERE), "tests", "golden", "spatialstats_errors.json")

# Argument refusals of the two spatialstats entry points whose checks the mirror restates (they fire before any third-party call).
# Inputs are named recipes so that the fixture stays plain JSON: see build_input().
SS_CASES = [
    ("sample_empirical_variogram", {"values": "v1d", "gsd": 1.0}),
    ("sample_empirical_variogram", {"values": "v1d", "coords": "c50x2", "subsample_method": "cdist_equidistant"}),
    ("sample_empirical_variogram", {"values": "v1d", "coords": "c50x2", "subsample_method": "pdist_ring"}),
    ("sample_empirical_variogram", {"values": "v2d", "coords": "c120x2", "subsample_method": "cdist_point"}),
    ("sample_empirical_variogram", {"values": "v1d", "coords": "c50x3", "subsample_method": "pdist_point"}),
    ("sample_empirical_variogram", {"values": "v2d", "subsample_method": "cdist_point"}),
    ("sample_empirical_variogram", {"values": "v2d", "gsd": 2.0, "subsample_method": "foo"}),
    ("sample_empirical_variogram", {"values": "text", "gsd": 2.0}),
    ("interp_nd_binning", {"df": "toy", "list_var_names": ["var1", "zz"], "statistic": "statistic", "min_count": None}),
    ("interp_nd_binning", {"df": "toy", "list_var_names": "var1", "statistic": "foo", "min_count": None}),
    ("interp_nd_binning", {"df": "toy", "list_var_names": ["var1", "var2"], "statistic": "statistic", "min_count": 5}),
    ("interp_nd_binning", {"df": "toy_empty", "list_var_names": ["var1"], "statistic": "statistic", "min_count": None}),
    ("interp_nd_binning", {"df": "toy_nan_stat", "list_var_names": ["var1", "var2"], "statistic": "statistic", "min_count": None}),
    ("interp_nd_binning", {"df": "toy_count", "list_var_names": ["var1", "var2"], "statistic": "statistic", "min_count": 1000}),
    ("interp_nd_binning", {"df": "toy_nd", "list_var_names": ["var1", "var2"], "statistic": "statistic", "min_count": None}),
]


def build_input(name):
    """Named inputs of SS_CASES (the test rebuilds them with the same function, imported from here as data recipe)."""
    import pandas as pd

    rng = np.random.default_rng(0)
    if name == "v1d":
        return rng.normal(size=50)
    if name == "v2d":
        return rng.normal(size=(10, 12))
    if name == "text":
        return "not an array"
    if name.startswith("c") and "x" in name:
        n, d = name[1:].split("x")
        return rng.uniform(0, 10, size=(int(n), int(d)))
    toy = pd.DataFrame({"var1": [1, 2, 3, 1, 2, 3, 1, 2, 3], "var2": [1, 1, 1, 2, 2, 2, 3, 3, 3],
                        "statistic": [1, 2, 3, 4, 5, 6, 7, 8, 9]})
    if name == "toy":
        return toy
    if name == "toy_empty":
        return toy.iloc[:0]
    if name == "toy_nan_stat":
        return toy.assign(statistic=np.nan)
    if name == "toy_count":
        return toy.assign(count=10)
    if name == "toy_nd":
        return toy.assign(nd=1)
    raise KeyError(name)


def main_spatialstats(ref) -> None:
    rec = []
    for fn, kw in SS_CASES:
        args = {k: (build_input(v) if k in ("values", "coords", "df") else v) for k, v in kw.items()}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                getattr(ref.spatialstats, fn)(**args)
                rec.append({"function": fn, "kwargs": kw, "raises": None, "message": None})
            except Exception as e:  # noqa: BLE001
                rec.append({"function": fn, "kwargs": kw, "raises": type(e).__name__, "message": str(e)})
    with open(OUT_SS, "w") as f:
        json.dump(rec, f, indent=1)
    for r in rec:
        print(r["function"], r["kwargs"], "->", r["raises"], "|", (r["message"] or "")[:110])


def main() -> None:
    ref = _refimport.load()
    main_spatialstats(ref)
    dem = (np.arange(12 * 14, dtype=np.float32).reshape(12, 14) * 0.5)
    rec = []
    for fn, kw in CASES:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                getattr(ref.terrain, fn)(dem, **kw)
                rec.append({"function": fn, "kwargs": kw, "raises": None, "message": None})
            except Exception as e:  # noqa: BLE001
                rec.append({"function": fn, "kwargs": kw, "raises": type(e).__name__, "message": str(e)})
    with open(OUT, "w") as f:
        json.dump(rec, f, indent=1)
    for r in rec:
        print(r["function"], r["kwargs"], "->", r["raises"], "|", (r["message"] or "")[:110])


if __name__ == "__main__":
    main()
