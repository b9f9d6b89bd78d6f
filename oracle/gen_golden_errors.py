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


def main() -> None:
    ref = _refimport.load()
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
