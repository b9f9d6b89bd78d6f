"""Theoretical variogram models used to fit the empirical variograms (host-side, NumPy).

The reference takes these from scikit-gstat (``skgstat.models``, reached at xdem/spatialstats.py:1561-1609 and 1707-1722);
that package is not vendored in the reference tree and absent here, so the six models xDEM supports are restated from
scikit-gstat's published definitions (effective-range convention: the model reaches ~95 % of its sill at ``r``).
**Parity unpinned** (no reference fixture can be generated without the package).

``model(h, r, c0[, s])``: lags ``h`` (array), effective range ``r``, partial sill ``c0``, smoothness ``s`` (stable, matern).
"""
from __future__ import annotations

import numpy as np
from scipy import special

SUPPORTED = ["spherical", "gaussian", "exponential", "cubic", "stable", "matern"]


def spherical(h, r, c0):
    h = np.asarray(h, dtype=float)
    x = h / r
    return np.where(h <= r, c0 * (1.5 * x - 0.5 * x**3), c0)


def exponential(h, r, c0):
    a = r / 3.0
    return c0 * (1.0 - np.exp(-np.asarray(h, dtype=float) / a))


def gaussian(h, r, c0):
    a = r / 2.0
    h = np.asarray(h, dtype=float)
    return c0 * (1.0 - np.exp(-(h**2) / a**2))


def cubic(h, r, c0):
    h = np.asarray(h, dtype=float)
    x = h / r
    return np.where(h < r, c0 * (7 * x**2 - 8.75 * x**3 + 3.5 * x**5 - 0.75 * x**7), c0)


def stable(h, r, c0, s):
    a = r / np.power(3.0, 1.0 / s)
    return c0 * (1.0 - np.exp(-np.power(np.asarray(h, dtype=float) / a, s)))


def matern(h, r, c0, s):
    h = np.asarray(h, dtype=float)
    a = r / 2.0
    x = h * np.sqrt(s) / a
    with np.errstate(all="ignore"):
        v = c0 * (1.0 - (2.0 / special.gamma(s)) * np.power(x, s) * special.kv(s, 2.0 * x))
    return np.where(h == 0, 0.0, v)


def model_name(model) -> str:
    """Canonical name from a 3-letter / full name or one of this module's functions (xdem/spatialstats.py:1549-1580)."""
    if callable(model):
        if getattr(model, "__module__", "") == __name__ and model.__name__ in SUPPORTED:
            return model.__name__
        raise ValueError("Variogram models can only be passed as functions of the xdem_amd.variogram_models module.")
    if isinstance(model, str):
        for supp in SUPPORTED:
            if model.lower() in (supp[:3], supp):
                return supp
        raise ValueError(f"Variogram model name {model} not recognized. Supported models are: " + ", ".join(SUPPORTED) + ".")
    raise ValueError("Variogram models can be passed as strings or functions. Supported models are: " + ", ".join(SUPPORTED) + ".")


def n_params(name: str) -> int:
    return 3 if name in ("stable", "matern") else 2
