"""Minimal georeferenced DEM container whose terrain / co-registration methods run on the GPU.

In the reference ``xdem.DEM`` subclasses ``geoutils.Raster`` (I/O, CRS, reprojection -- all out of scope here) and
only *forwards* to the hot paths (``xdem/dem.py:429-665``).  This class mirrors exactly that forwarding layer so that
``DEM.slope()``, ``DEM.get_terrain_attribute()`` and ``DEM.coregister_3d()`` keep their signatures; for real I/O keep
using geoutils / rasterio and either pass their Raster objects straight to ``xdem_amd.terrain`` functions (any object
with ``.data``, ``.res``, ``.transform``, ``.crs`` and a ``from_array`` classmethod is accepted) or wrap arrays with
``DEM.from_array``.
"""
from __future__ import annotations

import warnings
from typing import Any

import numpy as np

from . import coreg as _coreg
from . import spatialstats as _ss
from . import terrain


class DEM:
    """Array + north-up affine geotransform ``(a, b, c, d, e, f)`` (x = a*col + c, y = e*row + f) + opaque CRS."""

    def __init__(self, data: np.ndarray, transform=(1.0, 0.0, 0.0, 0.0, -1.0, 0.0), crs: Any = None, nodata=None) -> None:
        arr = np.asarray(data.filled(np.nan) if isinstance(data, np.ma.MaskedArray) and np.issubdtype(data.dtype, np.floating) else data)
        if isinstance(data, np.ma.MaskedArray) and not np.issubdtype(data.dtype, np.floating):
            arr = data.astype(np.float32).filled(np.nan)
        if arr.ndim == 3 and arr.shape[0] == 1:
            arr = arr[0]
        if arr.ndim != 2:
            raise ValueError("DEM data must be 2D")
        if nodata is not None and not isinstance(data, np.ma.MaskedArray) and not np.issubdtype(arr.dtype, np.floating):
            # integer rasters with a nodata value: geoutils hands them over masked, i.e. as float32 with NaN once filled
            # (xdem/dem.py:429-619 all start from get_nanarray()); elevations equal to nodata must not reach the kernels
            if np.any(arr == nodata):
                arr = np.where(arr == nodata, np.float32(np.nan), arr.astype(np.float32))
        elif nodata is not None and np.issubdtype(arr.dtype, np.floating):
            arr = np.where(arr == nodata, np.nan, arr)
        self.data = arr
        t = transform
        self.transform = tuple(float(v) for v in ((t.a, t.b, t.c, t.d, t.e, t.f) if hasattr(t, "a") else t))
        self.crs = crs
        self.nodata = nodata

    @classmethod
    def from_array(cls, data, transform, crs=None, nodata=None) -> "DEM":
        return cls(data, transform=transform, crs=crs, nodata=nodata)

    @property
    def res(self) -> tuple[float, float]:
        return (abs(self.transform[0]), abs(self.transform[4]))

    @property
    def shape(self) -> tuple[int, int]:
        return self.data.shape

    @property
    def dtype(self):
        return self.data.dtype

    def copy(self) -> "DEM":
        return DEM(self.data.copy(), self.transform, self.crs, self.nodata)

    # ---- terrain forwarders (xdem/dem.py:429-619) -------------------------------------------------------------
    def _surface_fit(self, method, surface_fit):
        if method is not None:
            warnings.warn("'method' is deprecated, use 'surface_fit' instead.", DeprecationWarning, stacklevel=3)
            return method
        return surface_fit

    def slope(self, method=None, surface_fit="Florinsky", degrees=True, mp_config=None) -> "DEM":
        return terrain.slope(self, surface_fit=self._surface_fit(method, surface_fit), degrees=degrees, mp_config=mp_config)

    def aspect(self, method=None, surface_fit="Florinsky", degrees=True, mp_config=None) -> "DEM":
        return terrain.aspect(self, surface_fit=self._surface_fit(method, surface_fit), degrees=degrees, mp_config=mp_config)

    def hillshade(self, method=None, surface_fit="Florinsky", azimuth=315.0, altitude=45.0, z_factor=1.0, mp_config=None) -> "DEM":
        return terrain.hillshade(self, surface_fit=self._surface_fit(method, surface_fit), azimuth=azimuth, altitude=altitude,
                                 z_factor=z_factor, mp_config=mp_config)

    def curvature(self, surface_fit="Florinsky", mp_config=None) -> "DEM":
        return terrain.curvature(self, surface_fit=surface_fit, mp_config=mp_config)

    def profile_curvature(self, surface_fit="Florinsky", curv_method="geometric", mp_config=None) -> "DEM":
        return terrain.profile_curvature(self, surface_fit=surface_fit, curv_method=curv_method, mp_config=mp_config)

    def tangential_curvature(self, surface_fit="Florinsky", curv_method="geometric", mp_config=None) -> "DEM":
        return terrain.tangential_curvature(self, surface_fit=surface_fit, curv_method=curv_method, mp_config=mp_config)

    def planform_curvature(self, surface_fit="Florinsky", curv_method="geometric", mp_config=None) -> "DEM":
        return terrain.planform_curvature(self, surface_fit=surface_fit, curv_method=curv_method, mp_config=mp_config)

    def flowline_curvature(self, surface_fit="Florinsky", curv_method="geometric", mp_config=None) -> "DEM":
        return terrain.flowline_curvature(self, surface_fit=surface_fit, curv_method=curv_method, mp_config=mp_config)

    def max_curvature(self, surface_fit="Florinsky", curv_method="geometric", mp_config=None) -> "DEM":
        return terrain.max_curvature(self, surface_fit=surface_fit, curv_method=curv_method, mp_config=mp_config)

    def min_curvature(self, surface_fit="Florinsky", curv_method="geometric", mp_config=None) -> "DEM":
        return terrain.min_curvature(self, surface_fit=surface_fit, curv_method=curv_method, mp_config=mp_config)

    def topographic_position_index(self, window_size=3, mp_config=None) -> "DEM":
        return terrain.topographic_position_index(self, window_size=window_size, mp_config=mp_config)

    def terrain_ruggedness_index(self, method="Riley", window_size=3, mp_config=None) -> "DEM":
        return terrain.terrain_ruggedness_index(self, method=method, window_size=window_size, mp_config=mp_config)

    def roughness(self, window_size=3, mp_config=None) -> "DEM":
        return terrain.roughness(self, window_size=window_size, mp_config=mp_config)

    def rugosity(self, mp_config=None) -> "DEM":
        return terrain.rugosity(self, mp_config=mp_config)

    def fractal_roughness(self, window_size_fractal=13, mp_config=None) -> "DEM":
        return terrain.fractal_roughness(self, window_size_fractal=window_size_fractal, mp_config=mp_config)

    def texture_shading(self, alpha: float = 0.8, mp_config=None) -> "DEM":
        return terrain.texture_shading(self, alpha=alpha, mp_config=mp_config)

    def get_terrain_attribute(self, attribute, **kwargs: Any):
        return terrain.get_terrain_attribute(self, attribute=attribute, **kwargs)

    # ---- co-registration forwarder (xdem/dem.py:621-665) -----------------------------------------------------------
    def coregister_3d(self, reference_elev: "DEM", coreg_method=None, inlier_mask=None, bias_vars=None, random_state=None,
                      **kwargs) -> "DEM":
        """Align this DEM to ``reference_elev`` (same grid) with ``coreg_method`` (upstream requires one and names Nuth and
        Kaab as the default in its docstring: ``None`` means ``NuthKaab(subsample=1)`` here).  ``random_state`` seeds the
        subsampling; ``resample`` (keyword, default True) as upstream; ``bias_vars`` belongs to bias-correction methods, which
        are not part of this package."""
        resample = kwargs.pop("resample", True)
        if bias_vars is not None:
            raise NotImplementedError("bias_vars is only used by bias-correction methods (not part of xdem_amd).")
        if random_state is not None:
            kwargs["random_state"] = random_state
        method = coreg_method if coreg_method is not None else _coreg.NuthKaab(subsample=1)
        if not isinstance(method, _coreg.NuthKaab):
            raise ValueError("Argument `coreg_method` must be an xdem_amd.coreg instance (e.g. xdem_amd.coreg.NuthKaab()).")
        if reference_elev.shape != self.shape or reference_elev.transform != self.transform:
            raise NotImplementedError("reference and to-be-aligned DEM must share one grid (reprojection is geoutils' job).")
        mask = None if inlier_mask is None else np.asarray(getattr(inlier_mask, "data", inlier_mask), dtype=bool)
        method.fit(reference_elev.data, self.data, mask, resolution=self.res, **kwargs)
        # array interface with transform=: for resample=False the horizontal shift moves the geotransform and only the
        # vertical shift touches the data (xdem/coreg/base.py:1567-1570, _apply_matrix_rst case 2)
        out, out_transform = method.apply(self.data, resample=resample, transform=self.transform)
        return DEM(out, out_transform, self.crs, self.nodata)

    # ---- uncertainty forwarder (xdem/dem.py:667-780) ---------------------------------------------------------------
    def estimate_uncertainty(self, other_elev: "DEM", stable_terrain=None, approach: str = "H2022", precision_of_other: str = "finer",
                             spread_estimator=_ss.nmad, variogram_estimator: str = "dowd", list_vars=("slope", "max_curvature"),
                             list_vario_models=("gaussian", "spherical"), z_name: str = "z", random_state=None):
        """Per-pixel error map and spatial correlation function of the errors of this DEM, from its difference to
        ``other_elev`` on stable terrain (Hugonnet et al., 2022) -- same steps as upstream: terrain attributes, N-D binning
        of the spread and its interpolant (``infer_heteroscedasticity_from_stable``), standardized Dowd variogram and model
        fit (``infer_spatial_correlation_from_stable``); all array work on the GPU.  ``other_elev`` must share this grid."""
        approach_dict = {"H2022": {"heterosc": True, "multi_range": True}, "R2009": {"heterosc": False, "multi_range": True},
                         "Basic": {"heterosc": False, "multi_range": False}}
        if approach not in approach_dict:
            raise ValueError(f"approach must be one of {list(approach_dict)}")
        if not isinstance(other_elev, DEM):
            raise TypeError("Other elevation should be a DEM or elevation point cloud object.")
        if other_elev.shape != self.shape or other_elev.transform != self.transform:
            raise NotImplementedError("both DEMs must share one grid (reprojection is geoutils' job).")
        dh = other_elev.data - self.data
        if precision_of_other == "same":
            dh = dh / np.sqrt(2)
        stable = None if stable_terrain is None else np.asarray(getattr(stable_terrain, "data", stable_terrain), dtype=bool)
        if approach_dict[approach]["heterosc"]:
            list_var_rast = [getattr(terrain, v)(self).data if isinstance(v, str) else np.asarray(getattr(v, "data", v)) for v in list_vars]
            sig = _ss.infer_heteroscedasticity_from_stable(dvalues=dh, list_var=list_var_rast, spread_statistic=spread_estimator,
                                                           stable_mask=stable)[0]
        else:
            sel = dh if stable is None else dh[stable]
            # xdem/dem.py:742-744: spread_estimator(dh[stable_terrain]); the default NMAD runs on the GPU (exact-median
            # selection), any other callable is the caller's host function
            sig = (_ss.nmad_device(sel)[1] if spread_estimator is _ss.nmad else float(spread_estimator(sel))) * np.ones(self.shape)
        if not approach_dict[approach]["multi_range"] and not isinstance(list_vario_models, str) and len(list_vario_models) > 1:
            warnings.warn("Several variogram models passed but this approach uses a single range,keeping only the first model.",
                          category=UserWarning)
            list_vario_models = list_vario_models[0]
        models = [list_vario_models] if isinstance(list_vario_models, str) else list(list_vario_models)
        corr_sig = _ss.infer_spatial_correlation_from_stable(dvalues=dh, list_models=models, stable_mask=stable, errors=sig,
                                                             estimator=variogram_estimator, gsd=self.res[0],
                                                             random_state=random_state)[2]
        return DEM(sig, self.transform, self.crs, None), corr_sig
