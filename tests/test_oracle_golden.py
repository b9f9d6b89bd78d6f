"""Pins the CPU oracle (oracle/terrain_oracle.py) against golden vectors recorded from the reference itself
(oracle/gen_golden.py -> tests/golden/terrain_*.npz).  Bar: BIT-EXACT, NaN positions included."""
import os

import numpy as np
import pytest

import terrain_oracle as to

from conftest import GOLDEN


def _same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.mark.parametrize("fname", ["terrain_T1_float32_nan.npz", "terrain_T1_float32_inf.npz",
                                   "terrain_T1_float64_nan.npz", "terrain_T1_float64_inf.npz"])
def test_T1_random_with_holes(fname):
    z = _load(fname)
    dem = z["dem"]
    n = 0
    for key in z.files:
        if key == "dem":
            continue
        fit, cm, res, attr = key.split("|")
        got = to.terrain_attributes(dem, [attr], resolution=float(res), surface_fit=fit, curv_method=cm)[0]
        assert _same(got, z[key]), key
        n += 1
    assert n > 50


def test_T2_terrain_like_full_set():
    z = _load("terrain_T2_f32.npz")
    dem = z["dem"]
    groups = {}
    for key in z.files:
        if key == "dem":
            continue
        *cfg, attr = key.split("|")
        groups.setdefault(tuple(cfg), []).append(attr)
    assert len(groups) > 10
    for cfg, attrs in groups.items():
        fit, cm, deg, az, alt, zf, tri = cfg
        got = to.terrain_attributes(dem, attrs, resolution=10.0, degrees=bool(int(deg)), hillshade_azimuth=float(az),
                                    hillshade_altitude=float(alt), hillshade_z_factor=float(zf), surface_fit=fit,
                                    curv_method=cm, tri_method=tri)
        for a, g in zip(attrs, got):
            assert _same(g, z["|".join(cfg) + "|" + a]), (cfg, a)


def test_T3_known_answer_dems():
    z = _load("terrain_T3_known_answers.npz")
    n = 0
    for key in z.files:
        if key.startswith("dem|"):
            continue
        name, fit, res, attr = key.split("|")
        got = to.terrain_attributes(z["dem|" + name], [attr], resolution=float(res), surface_fit=fit)[0]
        assert _same(got, z[key]), key
        n += 1
    assert n > 100


def test_T4_int32_input():
    z = _load("terrain_T4_int32.npz")
    attrs = [k for k in z.files if k != "dem"]
    got = to.terrain_attributes(z["dem"], attrs, resolution=5.0)
    for a, g in zip(attrs, got):
        assert g.dtype == np.float32
        assert _same(g, z[a]), a


def test_T5_window_sizes():
    z = _load("terrain_T5_windows.npz")
    for key in z.files:
        if "|" not in key:
            continue
        src, w, tri, attr = key.split("|")
        got = to.terrain_attributes(z[src], [attr], window_size=int(w), tri_method=tri)[0]
        assert _same(got, z[key]), key


def test_T9_rugosity_fractal_roughness():
    """f2 of SURVEY 8f.  Rugosity: bit-exact.  Fractal roughness: bit-exact on this NumPy build; np.log on float32 is a
    SIMD routine that is not correctly rounded, so 4 ulp are allowed for other CPU dispatch targets."""
    z = _load("terrain_T9_rugosity_fractal.npz")
    n = 0
    for key in z.files:
        if key.startswith("dem|"):
            continue
        parts = key.split("|")
        if parts[0] == "pyramid":
            dem, (attr, par) = z[f"dem|pyramid|{parts[1]}"], parts[2:]
        else:
            dem, (attr, par) = z[f"dem|{parts[0]}"], parts[1:]
        if attr == "rugosity":
            got = to.terrain_attributes(dem, [attr], resolution=float(par))[0]
            assert _same(got, z[key]), key
        else:
            got = to.terrain_attributes(dem, [attr], window_size_fractal=int(par))[0]
            ref = z[key]
            assert got.dtype == ref.dtype and np.array_equal(np.isnan(got), np.isnan(ref)), key
            ok = np.isfinite(ref)
            assert np.array_equal(np.isinf(got), np.isinf(ref))
            assert np.all(np.abs(got[ok] - ref[ok]) <= 4 * np.spacing(np.abs(ref[ok]))), key
        n += 1
    assert n >= 25
    # the reference's own known answers (tests/test_terrain/test_window.py:21-89)
    assert z["jenness|rugosity|100.0"][1, 1] == pytest.approx(10280.48 / 10000.0, rel=1e-4)
    for name, d in (("line", 1.0), ("plane", 2.0), ("cube", 3.0)):
        assert np.round(z[f"{name}|fractal_roughness|13"][6, 6], 3) == d


def test_T11_numba_engine_surface_fit():
    """Row a8 of SURVEY section 8: the oracle's numba recipe against outputs of the reference's OWN numba-engine code
    (surfit.py:948-1088, 1270-1303, run in the interpreter through oracle/_refimport.py's identity-njit shim): float64
    derivatives from the explicit loop, no dilated non-finite mask.  float32 outputs bit-exact (NaN and +-Inf positions
    included: slope 90 deg next to an Inf pixel, `curvature` -Inf, ...); float64 outputs to 1e-14 (the per-pixel scalar code
    path of NumPy uses pow() where the array path uses sqrt)."""
    z = _load("terrain_T11_numba_engine.npz")
    n = n_inf_windows = 0
    for key in z.files:
        parts = key.split("|")
        if parts[0] in ("dem", "boundary") or len(parts) != 5 or parts[1] == "win":
            continue
        name, fit, cm, res, attr = parts
        dem = z["dem|" + name]
        got = to.terrain_attributes(dem, [attr], resolution=float(res), surface_fit=fit, curv_method=cm, engine="numba")[0]
        ref = z[key]
        assert got.dtype == ref.dtype and np.array_equal(np.isnan(got), np.isnan(ref)), key
        if ref.dtype == np.float32:
            assert _same(got, ref), key
        else:
            fin = np.isfinite(ref)
            assert np.array_equal(got[~fin], ref[~fin], equal_nan=True), key
            assert np.all(np.abs(got[fin] - ref[fin]) <= 1e-14 * np.abs(ref[fin])), key
        if name.endswith("_inf"):
            # the engines differ here: values (not NaN) inside the dilated non-finite mask of the SciPy engine
            scipy_mask = to._window_invalid(dem, 5 if fit == "Florinsky" else 3)
            n_inf_windows += int(np.sum(scipy_mask & ~np.isnan(ref)))
        n += 1
    assert n > 500 and n_inf_windows > 100
    # the engine boundary called directly: radians, explicit out_dtype
    d = z["dem|T1_float32_nan"]
    for od in (np.float32, np.float64):
        got = to.surface_attributes(d, 2.0, list(to.SURFACE_ATTRIBUTES), od, "Florinsky", "geometric", engine="numba")
        ref = z[f"boundary|Florinsky|{np.dtype(od).name}"]
        assert got.dtype == ref.dtype and got.shape == ref.shape
        fin = np.isfinite(ref)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.all(np.abs(got[fin] - ref[fin]) <= (0 if od == np.float32 else 1e-14) * np.abs(ref[fin]))


def test_T11_numba_engine_windowed_indexes():
    """The Numba engine hands the callbacks the window in the DEM dtype (window.py:851): for a float64 DEM that is the SciPy
    engine's arithmetic up to the order of a float64 sum; for a float32 DEM the sums (TPI's mean, TRI's squares, rugosity's
    areas) are formed in float32 -- a noisier evaluation of the same quantity.  The oracle (like the HIP path) keeps the
    float64 window under every engine name; the recorded numba outputs must agree with it within the rounding noise of a
    float32 sum over the window, w^2 * 2^-24 * max|z| (float64 DEM: 1e-12 relative), masks bit-exact."""
    z = _load("terrain_T11_numba_engine.npz")
    n = 0
    for key in z.files:
        parts = key.split("|")
        if len(parts) == 5 and parts[1] == "win":
            name, _, w, tri, attr = parts
            kw = dict(window_size=int(w), tri_method=tri)
        elif len(parts) == 3 and parts[1] in ("rugosity", "fractal_roughness"):
            name, attr, par = parts
            kw = dict(resolution=float(par)) if attr == "rugosity" else dict(window_size_fractal=int(par))
            w = 3 if attr == "rugosity" else par
        else:
            continue
        dem = z["dem|" + name]
        got = to.terrain_attributes(dem, [attr], **kw)[0]
        ref = z[key]
        assert got.dtype == ref.dtype and np.array_equal(np.isnan(got), np.isnan(ref)), key
        fin = np.isfinite(ref) & np.isfinite(got)
        assert np.array_equal(got[~fin], ref[~fin], equal_nan=True), key
        zmax = float(np.max(np.abs(dem[np.isfinite(dem)])))
        if dem.dtype == np.float32:
            tol = int(w) ** 2 * 2.0**-24 * zmax * (4.0 if attr in ("rugosity", "fractal_roughness") else 1.0)
            tol = np.maximum(tol, 1e-5 * np.abs(ref[fin])) if attr in ("rugosity", "fractal_roughness") else tol
        else:
            tol = 1e-12 * np.maximum(np.abs(ref[fin]), zmax)
        assert np.all(np.abs(got[fin].astype(np.float64) - ref[fin]) <= tol), (key, float(np.max(np.abs(got[fin].astype(np.float64) - ref[fin]))))
        n += 1
    assert n >= 40


def test_T10_texture_shading():
    """f4 of SURVEY 8f: the oracle runs the same scipy.fft transforms as the reference -> bit-exact on this SciPy build
    (a different pocketfft build may differ in the last bits: 1e-5 of the output scale allowed)."""
    z = _load("terrain_T10_texture.npz")
    n = 0
    for key in z.files:
        if key.startswith("dem|"):
            continue
        name, alpha = key.split("|")
        got = to.terrain_attributes(z[f"dem|{name}"], ["texture_shading"], texture_alpha=float(alpha))[0]
        ref = z[key]
        assert got.dtype == ref.dtype and np.array_equal(np.isnan(got), np.isnan(ref)), key
        ok = np.isfinite(ref)
        if ok.any():
            assert np.abs(got[ok] - ref[ok]).max() <= 1e-5 * max(np.abs(ref[ok]).max(), 1e-30), key
        n += 1
    assert n == 13
    assert np.all(z["flat|0.8"] == 0) and np.isnan(z["allnan|0.8"]).all()  # tests/test_terrain/test_freq.py:53-57


def test_oracle_convolution_equals_scipy():
    """The restated convolution must reproduce scipy.ndimage.convolve (the reference's engine call) bit for bit."""
    scipy_ndimage = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(0)
    dem = (1000 + rng.normal(size=(40, 37)).cumsum(axis=0)).astype(np.float32)
    dem[5, 5] = np.nan
    for fit in ("horn", "zevenbergthorne", "florinsky"):
        for name, (tab, (const, power)) in to.conv_kernels(fit).items():
            k = tab.astype(np.float64) / (const * 3.0**power)
            want = scipy_ndimage.convolve(dem, k, mode="constant", cval=np.nan)
            got = to._convolve_nan_const(dem, k)
            assert _same(got, want), (fit, name)


def test_T12_engine_boundary_called_directly():
    """The oracle's engine-level functions against the reference's `surfit._get_surface_attributes` / `window._get_windowed_indexes`
    called directly: radians, explicit out_dtype, hillshade BEFORE the caller's clip -- bit for bit for the SciPy
    engine's surface fit and windowed indexes and for the Numba recipe's float32 outputs, 2e-14 relative for its float64 outputs."""
    z = _load("terrain_T12_engine_boundary.npz")
    surf = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature", "flowline_curvature",
            "max_curvature", "min_curvature"]
    n = 0
    for key in z.files:
        parts = key.split("|")
        if parts[0] == "surf":
            _, dname, engine, fit, cm, od, az = parts
            alt, zf = (45.0, 1.0) if az == "315.0" else (5.0, 4.0)
            attrs = surf[:3] if fit == "Horn" else surf
            got = np.asarray(to.surface_attributes(z[f"dem|{dname}"], 5.0, attrs, np.dtype(od), fit, cm, alt, float(az), zf, engine))
            if engine == "scipy" or od == "float32":
                assert _same(got, z[key]), key
            else:   # the Numba loop evaluates the formulas pixel by pixel with scalar `**` / sqrt: last-digit differences from NumPy's array
                # routines in float64 outputs (as in test_T11_numba_engine_surface_fit)
                w = z[key]
                assert got.dtype == w.dtype and np.array_equal(np.isnan(got), np.isnan(w)), key
                ok = ~np.isnan(w)
                assert np.all(np.abs(got[ok] - w[ok]) <= 2e-14 * np.maximum(np.abs(w[ok]), 1e-300)), key
            n += 1
        elif parts[0] in ("win", "rug", "frac") and parts[2] == "scipy":
            dem = z[f"dem|{parts[1]}"]
            if parts[0] == "win":
                got = to.windowed_indexes(dem, int(parts[3]), ["topographic_position_index", "terrain_ruggedness_index", "roughness"], dem.dtype, parts[4], 5.0)
            elif parts[0] == "rug":
                got = to.windowed_indexes(dem, 3, ["rugosity"], dem.dtype, "Riley", 5.0)
            else:
                got = to.windowed_indexes(dem, 13, ["fractal_roughness"], dem.dtype, "Riley", 5.0)
            assert _same(np.asarray(got), z[key]), key
            n += 1
    assert n >= 40
