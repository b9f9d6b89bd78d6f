"""Records the call signatures (parameter names, order, plain-literal defaults) of the reference's public entry points on the
three hot paths, so that the drop-in mirror in xdem_amd/ can be checked against them without the reference (which does
not exist on the GPU box).  Data only: names and literals, no code.  Container-only:  python oracle/gen_golden_signatures.py"""
from __future__ import annotations

import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "signatures.json")

TERRAIN = ["get_terrain_attribute", "slope", "aspect", "hillshade", "curvature", "profile_curvature", "tangential_curvature",
           "planform_curvature", "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index",
           "terrain_ruggedness_index", "roughness", "rugosity", "fractal_roughness", "texture_shading"]
SPATIALSTATS = ["nd_binning", "interp_nd_binning", "two_step_standardization", "infer_heteroscedasticity_from_stable",
                "sample_empirical_variogram", "get_variogram_model_func", "covariance_from_variogram", "correlation_from_variogram",
                "fit_sum_model_variogram", "infer_spatial_correlation_from_stable", "neff_circular_approx_theoretical",
                "neff_circular_approx_numerical", "neff_exact", "neff_hugonnet_approx", "number_effective_samples",
                "spatial_error_propagation", "mean_filter_nan", "patches_method", "_patches_convolution", "_patches_loop_quadrants",
                "convolution", "get_perbin_nd_binning", "_pandas_str_to_interval", "nmad"]


def _literal(v):
    if v is inspect.Parameter.empty:
        return "<required>"
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    if isinstance(v, (tuple, list)) and all(x is None or isinstance(x, (bool, int, float, str)) for x in v):
        return list(v)
    return "<object>"  # callables, arrays, generators ... : presence only


def record(fn) -> list:
    out = []
    for name, p in inspect.signature(fn).parameters.items():
        out.append({"name": name, "kind": p.kind.name, "default": _literal(p.default)})
    return out


DEM_METHODS = ["slope", "aspect", "hillshade", "curvature", "profile_curvature", "tangential_curvature", "planform_curvature",
               "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index",
               "roughness", "rugosity", "fractal_roughness", "texture_shading", "get_terrain_attribute", "coregister_3d",
               "estimate_uncertainty"]


def record_dem_methods() -> dict:
    """The DEM class cannot be imported here (it subclasses geoutils' raster class): its method signatures are read from the
    syntax tree of xdem/dem.py instead -- parameter names, order and plain-literal defaults only."""
    import ast

    tree = ast.parse(open(os.path.join(_refimport.REFERENCE_ROOT, "xdem", "dem.py")).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DEM")
    out = {}
    for node in cls.body:
        if not isinstance(node, ast.FunctionDef) or node.name not in DEM_METHODS:
            continue
        if any(isinstance(d, ast.Name) and d.id == "overload" for d in node.decorator_list):
            continue
        a = node.args
        pos = a.posonlyargs + a.args
        defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
        params = []
        for arg, d in zip(pos, defaults):
            params.append({"name": arg.arg, "kind": "POSITIONAL_OR_KEYWORD", "default": _ast_literal(d)})
        for arg, d in zip(a.kwonlyargs, a.kw_defaults):
            params.append({"name": arg.arg, "kind": "KEYWORD_ONLY", "default": _ast_literal(d)})
        if a.kwarg is not None:
            params.append({"name": a.kwarg.arg, "kind": "VAR_KEYWORD", "default": "<required>"})
        out[node.name] = params
    return out


def _ast_literal(node):
    import ast

    if node is None:
        return "<required>"
    try:
        return _literal(ast.literal_eval(node))
    except Exception:
        return "<object>"


def main() -> None:
    ref = _refimport.load()
    rec = {"terrain": {}, "spatialstats": {}, "coreg": {}, "dem": record_dem_methods()}
    for n in TERRAIN:
        rec["terrain"][n] = record(getattr(ref.terrain, n))
    for n in SPATIALSTATS:
        rec["spatialstats"][n] = record(getattr(ref.spatialstats, n))
    # the engine-boundary functions of SURVEY 8b rows 1-2 and the texture helper, under their own modules
    import importlib

    rec["surfit"] = {"_get_surface_attributes": record(ref.surfit._get_surface_attributes)}
    rec["window"] = {"_get_windowed_indexes": record(ref.window._get_windowed_indexes)}
    rec["freq"] = {"_nextprod_fft": record(importlib.import_module("xdem.terrain.freq")._nextprod_fft)}
    rec["coreg"]["NuthKaab.__init__"] = record(ref.affine.NuthKaab.__init__)
    rec["coreg"]["NuthKaab.fit"] = record(ref.affine.NuthKaab.fit)
    rec["coreg"]["NuthKaab.fit_and_apply"] = record(ref.affine.NuthKaab.fit_and_apply)
    with open(OUT, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("signatures written:", {k: len(v) for k, v in rec.items()})


if __name__ == "__main__":
    main()
