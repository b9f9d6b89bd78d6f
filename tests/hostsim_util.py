"""ctypes driver of the host-compiled numerics harness (tests/hostsim/terrain_hostsim.cpp) -- test tool only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostsim", "terrain_hostsim.cpp")
HDR = os.path.join(os.path.dirname(HERE), "xdem_amd", "csrc", "terrain_math.h")
HDR2 = os.path.join(os.path.dirname(HERE), "xdem_amd", "csrc", "terrain_nonfinite.h")
SO = os.path.join(HERE, "hostsim", "_hostsim.so")

ATTR_BITS = {
    "slope": 0, "aspect": 1, "hillshade": 2, "curvature": 3, "profile_curvature": 4, "tangential_curvature": 5,
    "planform_curvature": 6, "flowline_curvature": 7, "max_curvature": 8, "min_curvature": 9,
    "topographic_position_index": 10, "terrain_ruggedness_index": 11, "roughness": 12,
}
FITS = {"horn": 0, "zevenbergthorne": 1, "florinsky": 2}


def build():
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR), os.path.getmtime(HDR2)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    return ctypes.CDLL(SO)


def hostsim_terrain(dem, attrs, resolution=1.0, surface_fit="Florinsky", curv_method="geometric", tri_method="Riley",
                    hillshade_altitude=45.0, hillshade_azimuth=315.0, hillshade_z_factor=1.0, degrees=True,
                    out_dtype=None, halo_top=0, halo_bottom=0, tile_rows=32, tail=2, engine="scipy", _nonfinite=False):
    """`tail`: attribute math of the specialised float32 kernels, 2 = lean (library default), 0 = mixed (round 2).
    `engine="numba"`: what xdem_amd.terrain does for that name -- surface-fit attributes from the float64-input kernels on the
    widened DEM plus the +-Inf rule of terrain_nonfinite.h; windowed indexes on the DEM as is."""
    if engine == "numba":
        surf = [a for a in attrs if ATTR_BITS[a] < 10]
        rest = [a for a in attrs if ATTR_BITS[a] >= 10]
        res = {}
        kw = dict(resolution=resolution, surface_fit=surface_fit, curv_method=curv_method, tri_method=tri_method,
                  hillshade_altitude=hillshade_altitude, hillshade_azimuth=hillshade_azimuth,
                  hillshade_z_factor=hillshade_z_factor, degrees=degrees, out_dtype=out_dtype or dem.dtype, halo_top=halo_top,
                  halo_bottom=halo_bottom, tile_rows=tile_rows, tail=tail)
        if surf:
            res.update(zip(surf, hostsim_terrain(np.asarray(dem, dtype=np.float64), surf, _nonfinite=True, **kw)))
        if rest:
            res.update(zip(rest, hostsim_terrain(dem, rest, **kw)))
        return [res[a] for a in attrs]
    lib = build()
    lib.hostsim_set_tail(int(tail))
    dem = np.ascontiguousarray(dem)
    assert dem.dtype in (np.float32, np.float64)
    out_dtype = np.dtype(out_dtype or dem.dtype)
    Hbuf, W = dem.shape
    H = Hbuf - halo_top - halo_bottom
    mask = 0
    planes = (ctypes.c_void_p * 13)()
    outs = {}
    for a in attrs:
        b = ATTR_BITS[a]
        mask |= 1 << b
        outs[a] = np.full((H, W), -12345.0, dtype=out_dtype)
        planes[b] = outs[a].ctypes.data
    lib.hostsim_terrain.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                    ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_uint32, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rc = lib.hostsim_terrain(dem.ctypes.data, 0 if dem.dtype == np.float32 else 1, H, W, halo_top, halo_bottom,
                             tile_rows, float(resolution), FITS[surface_fit.lower()],
                             int(curv_method.lower() == "directional"), mask, int(tri_method.lower() == "wilson"),
                             float(hillshade_altitude), float(hillshade_azimuth), float(hillshade_z_factor),
                             int(bool(degrees)), 0 if out_dtype == np.float32 else 1, planes)
    assert rc == 0
    if _nonfinite:
        lib.hostsim_nonfinite.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p]
        rc = lib.hostsim_nonfinite(dem.ctypes.data, 0 if dem.dtype == np.float32 else 1, H, W, halo_top, halo_bottom,
                                   float(resolution), FITS[surface_fit.lower()], int(curv_method.lower() == "directional"), mask,
                                   float(hillshade_altitude), float(hillshade_azimuth), float(hillshade_z_factor),
                                   int(bool(degrees)), 0 if out_dtype == np.float32 else 1, planes)
        assert rc == 0
    return [outs[a] for a in attrs]


def ulp_diff(a, b):
    """Distance in units-in-the-last-place between two float arrays of the same dtype (NaN==NaN -> 0)."""
    a = np.asarray(a)
    b = np.asarray(b)
    it = np.int32 if a.dtype == np.float32 else np.int64
    ia = a.view(it).astype(np.int64)
    ib = b.view(it).astype(np.int64)
    sign = np.int64(-(2**31)) if a.dtype == np.float32 else np.int64(-(2**63))
    ia = np.where(ia < 0, sign - ia, ia)
    ib = np.where(ib < 0, sign - ib, ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    d[both_nan] = 0
    one_nan = np.isnan(a) ^ np.isnan(b)
    d[one_nan] = np.iinfo(np.int64).max
    return d
