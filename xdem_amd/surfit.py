"""The surface-fit ENGINE BOUNDARY of the reference under its own name and signature:

    xdem.terrain.surfit._get_surface_attributes(dem, resolution, surface_attributes, out_dtype, surface_fit, curv_method,
                                                engine, **kwargs)                      xdem/terrain/surfit.py:1197-1305

-- the function `get_terrain_attribute` hands the DEM array to (terrain.py:571-580) and the reference's tests call directly
(tests/test_terrain/test_surfit.py:445-452).  Here it is one launch of the fused HIP kernel (``xdemhip_terrain``): the stack
(n_attributes, H, W) in `out_dtype`, attributes in the order asked for, slope and aspect in RADIANS as the engine returns them
(the conversion to degrees is the caller's post-step, terrain.py:586-591).

Hillshade comes out as the engine returns it, NOT clipped to [0, 255]: the clip is the caller's post-step (terrain.py:594-596),
which every other entry of the library fuses into the kernel; here the launch asks for the unclipped plane (bit 1 of
``xdemhip_terrain``'s `degrees` argument; float64 attribute tail).  Pinned by fixtures recorded from the reference's engine
functions called directly (tests/golden/terrain_T12_engine_boundary.npz, tests/test_engine_boundary_gpu.py).
"""
from __future__ import annotations

from typing import Any

import numpy as np

from . import _lib, terrain


def _engine_stack(dem, names: list[str], out_dtype, engine: str, resolution: float, surface_fit: str = "Florinsky",
                  curv_method: str = "geometric", tri_method: str = "Riley", window_size: int = 3, hillshade_altitude: float = 45.0,
                  hillshade_azimuth: float = 315.0, hillshade_z_factor: float = 1.0, ctx: _lib.Context | None = None) -> np.ndarray:
    """(len(names), H, W) stack of the fused kernel's planes for a 2-D float array, radians, hillshade unclipped, no Raster
    handling: what both engine-boundary mirrors (this module and ``xdem_amd.window``) return."""
    if engine not in ("scipy", "numba", "hip"):
        raise ValueError(f"engine must be 'scipy', 'numba' or 'hip' (got '{engine}'); all of them run on the GPU.")
    out_dtype = np.dtype(out_dtype)
    if out_dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError(f"out_dtype must be float32 or float64 on the HIP engine (got {out_dtype}).")
    arr = np.asarray(dem)
    if arr.ndim != 2:
        raise ValueError("The DEM must be a 2D array.")
    if arr.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        arr = arr.astype(np.float32 if (np.issubdtype(arr.dtype, np.integer) or arr.dtype.itemsize <= 4) else np.float64)
    arr = np.ascontiguousarray(arr)
    H, W = arr.shape
    stack = np.empty((len(names), H, W), dtype=out_dtype)
    planes = {}
    for i, a in enumerate(names):
        planes.setdefault(a, i)
    ctx = ctx or _lib.default_context()
    groups = [(arr, list(planes), 0)]
    if engine == "numba" and any(a in terrain.list_requiring_surface_fit for a in planes):
        # upstream's Numba recipe (surfit.py:1270-1303): unrounded float64 derivatives, no dilated non-finite mask
        wide = arr if arr.dtype == np.float64 else arr.astype(np.float64)
        groups = [(wide, [a for a in planes if a in terrain.list_requiring_surface_fit], 1),
                  (arr, [a for a in planes if a not in terrain.list_requiring_surface_fit], 0)]
    for src, group, nonfinite in groups:
        if not group:
            continue
        with ctx.option_scope("terrain_nonfinite", nonfinite):
            terrain.launch_terrain(ctx, src.ctypes.data, src.dtype, H, W, W, 0, 0, resolution, surface_fit, curv_method, group, tri_method,
                                   window_size, hillshade_altitude, hillshade_azimuth, hillshade_z_factor, 2, out_dtype,
                                   {a: stack[planes[a]].ctypes.data for a in group}, _lib.HOST, window_size)
    for i, a in enumerate(names):   # (a name asked for twice: the same plane twice, as upstream's index lists give)
        if planes[a] != i:
            stack[i] = stack[planes[a]]
    return stack


def _get_surface_attributes(dem, resolution: float, surface_attributes: list[str], out_dtype=np.float32, surface_fit: str = "Florinsky",
                            curv_method: str = "geometric", engine: str = "scipy", **kwargs: Any) -> np.ndarray:
    """See the module docstring.  `kwargs`: ``hillshade_azimuth`` / ``hillshade_altitude`` / ``hillshade_z_factor`` as upstream
    forwards them (terrain.py:564-568; defaults 315 / 45 / 1).  `engine` names which of upstream's two precision recipes is
    reproduced ("scipy": derivatives rounded to the DEM dtype, dilated non-finite mask; "numba": unrounded float64 derivatives,
    +-Inf through the arithmetic) -- both run on the GPU."""
    unknown = set(kwargs) - {"hillshade_azimuth", "hillshade_altitude", "hillshade_z_factor"}
    if unknown:
        raise TypeError(f"_get_surface_attributes() got unexpected keyword arguments {sorted(unknown)}")
    bad = [a for a in surface_attributes if a not in terrain.list_requiring_surface_fit]
    if bad:
        raise ValueError(f"not surface-fit attributes: {bad}")
    if surface_fit.lower() not in ("horn", "zevenbergthorne", "florinsky"):
        raise ValueError(f"surface_fit must be 'Horn', 'ZevenbergThorne' or 'Florinsky' (got '{surface_fit}')")
    return _engine_stack(dem, list(surface_attributes), out_dtype, engine, float(resolution), surface_fit, curv_method,
                         hillshade_altitude=kwargs.get("hillshade_altitude", 45.0), hillshade_azimuth=kwargs.get("hillshade_azimuth", 315.0),
                         hillshade_z_factor=kwargs.get("hillshade_z_factor", 1.0))
