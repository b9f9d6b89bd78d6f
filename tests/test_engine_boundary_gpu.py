"""The engine boundary of SURVEY 8b rows 1-2 under the reference's own names: xdem_amd.terrain.surfit._get_surface_attributes and
xdem_amd.terrain.window._get_windowed_indexes against outputs of the reference's functions of the same name called directly
(tests/golden/terrain_T12_engine_boundary.npz: both upstream engines, float32 / float64 DEMs and out_dtypes, radians, hillshade
NOT clipped -- on either side).  Bar as for the terrain path: NaN masks identical, values within 1e-6 true relative of the reference
(the SciPy engine's planes; the Numba engine's float32 window sums within their rounding noise)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

Z = np.load(os.path.join(GOLDEN, "terrain_T12_engine_boundary.npz"))
SURF = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature", "flowline_curvature",
        "max_curvature", "min_curvature"]
SAH = ["slope", "aspect", "hillshade"]
WIN = ["topographic_position_index", "terrain_ruggedness_index", "roughness"]


def close(got, want, rel, what):
    assert got.shape == want.shape and got.dtype == want.dtype, what
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"NaN mask differs: {what}"
    ok = ~np.isnan(want)
    g, w = got[ok].astype(np.float64), want[ok].astype(np.float64)
    scale = np.maximum(np.abs(w), 1e-30)
    bad = np.abs(g - w) > rel * np.maximum(scale, np.percentile(np.abs(w), 50) * 1e-3 if w.size else 1.0)
    assert not bad.any(), f"{what}: {int(bad.sum())} values off by more than {rel} relative, worst {np.max(np.abs(g - w) / scale):.3e}"


@pytest.mark.parametrize("key", [k for k in Z.files if k.startswith("surf|")])
def test_surface_engine_equals_the_reference_engine(key):
    import xdem_amd

    _, dname, engine, fit, cm, od, az = key.split("|")
    hs = {"315.0": (315.0, 45.0, 1.0), "120.0": (120.0, 5.0, 4.0)}[az]
    attrs = SAH if fit == "Horn" else SURF
    got = xdem_amd.terrain.surfit._get_surface_attributes(Z[f"dem|{dname}"], 5.0, attrs, out_dtype=np.dtype(od), surface_fit=fit, curv_method=cm,
                                                          engine=engine, hillshade_azimuth=hs[0], hillshade_altitude=hs[1], hillshade_z_factor=hs[2])
    want = Z[key]
    i_hs = attrs.index("hillshade")
    assert az == "315.0" or np.nanmin(want[i_hs]) < 0          # the second setting leaves [0, 255]: the engine does not clip, nor does the mirror
    assert got.shape == want.shape and got.dtype == np.dtype(od)
    for i, a in enumerate(attrs):
        if a == "aspect":    # flat or near-flat pixels may sit on either side of the 0 / 2 pi seam
            d = np.abs(got[i].astype(np.float64) - want[i]) % (2 * np.pi)
            d = np.minimum(d, 2 * np.pi - d)
            assert np.array_equal(np.isnan(got[i]), np.isnan(want[i])) and np.nanmax(d) < 2e-6, (key, a, np.nanmax(d))
        else:
            close(got[i], want[i], 2e-6 if od == "float32" else 1e-6, (key, a))


@pytest.mark.parametrize("key", [k for k in Z.files if k.startswith(("win|", "rug|", "frac|"))])
def test_window_engine_equals_the_reference_engine(key):
    import xdem_amd

    parts = key.split("|")
    dem = Z[f"dem|{parts[1]}"]
    engine = parts[2]
    if parts[0] == "win":
        w, tri, names = int(parts[3]), parts[4], WIN
    elif parts[0] == "rug":
        w, tri, names = 3, "Riley", ["rugosity"]
    else:
        w, tri, names = 13, "Riley", ["fractal_roughness"]
    got = xdem_amd.terrain.window._get_windowed_indexes(dem, w, names, 5.0, out_dtype=dem.dtype, tri_method=tri, engine=engine,
                                                        force_scipy_backend=None if engine == "numba" else "generic")
    want = Z[key]
    if engine == "numba" and dem.dtype == np.float32:
        # upstream's Numba engine sums the window in the DEM dtype (window.py:851): the float32 rounding noise of w x w values of a 1000 m
        # DEM, absolute -- the rule of test_T11_numba_engine_reference_fixtures_on_the_hip_path (the kernels evaluate float64 windows)
        tol = w * w * 2.0**-24 * float(np.nanmax(np.abs(dem)))
        for i, a in enumerate(names):
            assert got[i].dtype == want[i].dtype and np.array_equal(np.isnan(got[i]), np.isnan(want[i])), (key, a)
            fin = np.isfinite(want[i])
            if a == "fractal_roughness":   # (a box-counting dimension of the surface: the noise of the sums does not enter it)
                close(got[i], want[i], 2e-6, (key, a))
            else:
                assert np.all(np.abs(got[i][fin].astype(np.float64) - want[i][fin]) <= tol), (key, a)
        return
    # the SciPy engine hands float64 windows to its callbacks: exact up to the final rounding
    rel = 2e-6 if parts[0] == "frac" else 1e-6   # (fractal roughness: NumPy's float32 log)
    for i, a in enumerate(names):
        close(got[i], want[i], rel, (key, a))


def test_engine_stack_against_the_public_call():
    """The boundary mirror against get_terrain_attribute(degrees=False): the same planes up to the clip of the hillshade, any order, duplicates."""
    import xdem_amd

    dem = Z["dem|float32"]
    names = ["min_curvature", "slope", "hillshade", "slope", "aspect"]
    planes = xdem_amd.terrain.get_terrain_attribute(dem, ["min_curvature", "slope", "hillshade", "aspect"], resolution=5.0, degrees=False,
                                                    surface_fit="ZevenbergThorne", hillshade_altitude=5.0, hillshade_z_factor=4.0)
    stack = xdem_amd.terrain.surfit._get_surface_attributes(dem, 5.0, names, out_dtype=np.float32, surface_fit="ZevenbergThorne",
                                                            hillshade_altitude=5.0, hillshade_z_factor=4.0)
    by_name = dict(zip(["min_curvature", "slope", "hillshade", "aspect"], planes))
    assert np.nanmin(stack[2]) < 0 and np.nanmin(by_name["hillshade"]) == 0          # the engine does not clip, the public call does
    for i, a in enumerate(names):
        want = np.clip(stack[i], 0, 255) if a == "hillshade" else stack[i]
        # (the public call runs the mixed float32 tail, the engine call the float64 one: same planes within the float32 rounding)
        assert np.array_equal(np.isnan(want), np.isnan(by_name[a])) and np.allclose(want, by_name[a], rtol=2e-6, atol=1e-6, equal_nan=True), a
    assert np.array_equal(stack[1], stack[3], equal_nan=True)
    with pytest.raises(ValueError, match="not surface-fit attributes"):
        xdem_amd.terrain.surfit._get_surface_attributes(dem, 5.0, ["roughness"])
    with pytest.raises(ValueError, match="not windowed indexes"):
        xdem_amd.terrain.window._get_windowed_indexes(dem, 3, ["slope"], 5.0)
    with pytest.raises(TypeError, match="unexpected keyword"):
        xdem_amd.terrain.surfit._get_surface_attributes(dem, 5.0, ["slope"], azimuth=3.0)
