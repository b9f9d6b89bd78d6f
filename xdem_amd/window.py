"""The windowed-index ENGINE BOUNDARY of the reference under its own name and signature:

    xdem.terrain.window._get_windowed_indexes(dem, window_size, windowed_indexes, resolution, out_dtype, tri_method, engine,
                                              force_scipy_backend)                      xdem/terrain/window.py:926-1002

-- what `get_terrain_attribute` calls for TPI / TRI / roughness / rugosity / fractal roughness (terrain.py:600-630) and the
reference's tests call directly (tests/test_terrain/test_window.py:120-186).  Here: one launch of the HIP kernels, the stack
(n_indexes, H, W) in `out_dtype`, indexes in the order asked for.  `engine` and `force_scipy_backend` name upstream's CPU
back-ends, whose results upstream's own tests hold equal; every value runs the same kernels (float64 window sums, the SciPy
engine's recipe)."""
from __future__ import annotations

import numpy as np

from . import terrain
from .surfit import _engine_stack


def _get_windowed_indexes(dem, window_size: int, windowed_indexes: list[str], resolution: float, out_dtype=np.float32,
                          tri_method: str = "Riley", engine: str = "scipy", force_scipy_backend: str | None = None) -> np.ndarray:
    """See the module docstring.  ``window_size`` applies to every index asked for, fractal roughness included (the caller uses
    its own ``window_size_fractal`` call for that one, terrain.py:619-630)."""
    allowed = set(terrain.list_requiring_windowed_index) | set(terrain.list_requiring_windowed_fractal_index)
    bad = [a for a in windowed_indexes if a not in allowed]
    if bad:
        raise ValueError(f"not windowed indexes: {bad}")
    if force_scipy_backend not in (None, "generic", "vectorized"):
        raise ValueError("force_scipy_backend must be None, 'generic' or 'vectorized'")
    if tri_method.lower() not in ("riley", "wilson"):
        raise ValueError("tri_method must be 'Riley' or 'Wilson'")
    return _engine_stack(dem, list(windowed_indexes), out_dtype, engine, float(resolution) if resolution is not None else 1.0,
                         tri_method=tri_method, window_size=int(window_size))
