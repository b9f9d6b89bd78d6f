"""Import harness for the upstream reference (TEST INFRASTRUCTURE ONLY, container-only).

Makes the pure-Python reference at /root/reference importable *in the build container* even though
its geo dependencies (geoutils, rasterio, pyproj, ...) are not installed, so that
``oracle/gen_golden.py`` can run the reference's own numeric functions and record golden vectors.
Nothing in the shipped product, the -m gpu tests, smoke() or bench.py imports this module; the
reference itself never travels to the GPU box (only the .npz fixtures under tests/golden/ do).

Mechanism: ``xdem`` is registered as a namespace-like package (its ``__init__`` is skipped because
it pulls every subpackage), the missing third-party roots are auto-stubbed through a meta-path
finder, and the three geoutils entry points the numeric code really calls get tiny real shims.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("XDEM_REFERENCE_ROOT", "/root/reference")

_STUB_ROOTS = ("geoutils", "geopandas", "rasterio", "affine", "pyproj", "shapely", "pyogrio", "skgstat")


class _Anything:
    """Attribute sink: any attribute / call / subscript returns another sink (only used for type hints)."""

    def __init__(self, name: str = "stub") -> None:
        self.__name__ = name

    def __getattr__(self, item: str):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Anything(item)

    def __call__(self, *a, **k):
        return _Anything("call")

    def __getitem__(self, item):
        return _Anything("item")

    def __or__(self, other):
        return self

    def __ror__(self, other):
        return self

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, item: str):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Anything(item)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []  # behaves as a package so that submodule imports resolve through this finder too
        return m

    def exec_module(self, module):
        pass


def _profile(*a, **k):
    """geoutils.profiler.profile stand-in: decorator factory that returns the function unchanged."""

    def deco(f):
        return f

    return deco


class _Raster:
    """Sentinel for isinstance(dem, gu.Raster) checks: no instance of it is ever created here."""


def _get_array_and_mask(array, check_shape=True, copy=True):
    """geoutils.raster.get_array_and_mask for plain / masked ndarrays: masked or non-finite -> NaN, plus the mask."""
    if isinstance(array, np.ma.MaskedArray):
        arr = np.array(array.data, copy=True)
        m = np.ma.getmaskarray(array)
        if np.issubdtype(arr.dtype, np.integer):
            if m.any():
                arr = arr.astype(np.float32)
                arr[m] = np.nan
        else:
            arr[m] = np.nan
    else:
        arr = np.array(array, copy=copy)
    invalid = ~np.isfinite(arr) if np.issubdtype(arr.dtype, np.floating) else np.zeros(arr.shape, bool)
    return arr, invalid


def _numba_shim() -> types.ModuleType:
    """A `numba` that compiles nothing: the reference's @njit functions (surfit.py:948-1088, window.py:767-870,
    spatialstats.py:2528-2555) are plain Python, so with `njit` = identity (both call forms: `@njit(...)` on a
    function and `njit(...)(f)`), `prange` = `range` and `typed.List` = `list` the reference's own numba-engine code
    runs in the interpreter -- same statements, same IEEE operations in the same order (Numba does not contract or
    reassociate without fastmath), only slower.  That is what pins row a8: `engine="numba"` fixtures are outputs of the
    reference's code, not of a restatement."""
    m = types.ModuleType("numba")

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]  # bare @njit

        def deco(f):
            return f

        return deco

    m.njit = njit
    m.jit = njit
    m.prange = range
    typed = types.ModuleType("numba.typed")
    typed.List = list
    m.typed = typed
    m.__xdem_oracle_shim__ = True
    sys.modules["numba.typed"] = typed
    return m


_installed = False


def install() -> None:
    """Install the stubs and make ``xdem.*`` submodules importable from the reference tree."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "xdem")):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT} (only exists in the build container)")
    sys.meta_path.insert(0, _StubFinder())
    if "numba" not in sys.modules:
        sys.modules["numba"] = _numba_shim()
    import geoutils  # noqa: F401  (stub)
    import geoutils.profiler
    import geoutils.raster

    geoutils.profiler.profile = _profile
    geoutils.Raster = _Raster
    geoutils.raster.Raster = _Raster
    geoutils.raster.RasterType = _Raster
    geoutils.raster.get_array_and_mask = _get_array_and_mask
    import geoutils.raster.array

    geoutils.raster.array.get_array_and_mask = _get_array_and_mask   # spatialstats.py:37 imports it from there
    import geopandas

    geopandas.GeoDataFrame = type("GeoDataFrame", (), {})  # isinstance() sentinel, never instantiated
    import geoutils.vector.vector

    geoutils.vector.vector.Vector = type("Vector", (), {})  # isinstance() sentinel (spatialstats.py:676-693)
    geoutils.vector.vector.VectorType = geoutils.vector.vector.Vector

    pkg = types.ModuleType("xdem")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "xdem")]
    pkg.__version__ = "reference"
    sys.modules["xdem"] = pkg
    for sub in ("terrain", "coreg"):
        m = types.ModuleType(f"xdem.{sub}")
        m.__path__ = [os.path.join(REFERENCE_ROOT, "xdem", sub)]
        sys.modules[f"xdem.{sub}"] = m
        setattr(pkg, sub, m)
    _installed = True


def load():
    """Return a namespace with the reference modules used by the golden-vector generator."""
    install()
    import importlib

    ns = types.SimpleNamespace()
    ns.terrain = importlib.import_module("xdem.terrain.terrain")
    ns.surfit = importlib.import_module("xdem.terrain.surfit")
    ns.window = importlib.import_module("xdem.terrain.window")
    ns.spatialstats = importlib.import_module("xdem.spatialstats")
    ns.affine = importlib.import_module("xdem.coreg.affine")
    ns.base = importlib.import_module("xdem.coreg.base")
    return ns
