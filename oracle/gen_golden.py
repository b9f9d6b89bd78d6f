"""Golden-vector generator: runs the UPSTREAM REFERENCE (imported from /root/reference through
oracle/_refimport.py) on small seeded inputs and records inputs + reference outputs under
tests/golden/*.npz.  Container-only script; the fixtures (data only) are what is committed and what
travels to the GPU box.  Re-run with:  python oracle/gen_golden.py

Fixture families:
  terrain_T1_*   default_rng(42).normal((20,20)) f32/f64 with NaN/Inf holes x fits x curvature methods x res
  terrain_T2_*   cumulative-sum "terrain-like" 64x64 f32 near 1000 m (precision stress), full attribute
                 set incl. TPI/TRI (Riley + Wilson), degrees on/off, two hillshade settings
  terrain_T3     reference's data-free known-answer DEMs (test_surfit.py:228-411, terrain.py doctests)
  terrain_T4     int32 DEM -> float32 outputs
  terrain_T5     window sizes 3/5/7 for TPI/TRI
  terrain_T9     rugosity + fractal roughness (incl. the known answers of test_window.py:21-89)
  terrain_T11    the reference's own NUMBA-engine code (surfit.py:948-1088, 1270-1303; window.py:767-870, 980-1000) run
                 in the interpreter through the identity-njit shim of _refimport.py: T1 DEMs (f32 + f64, NaN and Inf
                 holes), a terrain-like DEM, three fits, both curvature methods, windowed indexes
  terrain_T12    the engine boundary called directly (surfit._get_surface_attributes, window._get_windowed_indexes), both engines
  nk_T5_*        Nuth-Kaab bin-fit cases (aspect binning, nanmedian per bin, curve_fit) + aux gradient (T8)
  nk_T6          _iterate_method stop-rule trace
  vario_T7       _choose_cdist_equidistant_sampling_parameters table + default bin edges
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
ref = _refimport.load()

SURF = ["slope", "aspect", "hillshade", "curvature", "profile_curvature", "tangential_curvature",
        "planform_curvature", "flowline_curvature", "max_curvature", "min_curvature"]
SAH = ["slope", "aspect", "hillshade"]
WIN = ["topographic_position_index", "terrain_ruggedness_index"]


def _check_tables() -> None:
    """The oracle's generated stencil tables must equal the reference's tabulated ones."""
    import terrain_oracle as to

    m = {"horn": {"zx": "h2", "zy": "h1"},
         "zevenbergthorne": {"zx": "zt_h", "zy": "zt_g", "zxx": "zt_e", "zyy": "zt_d", "zxy": "zt_f"},
         "florinsky": {"zx": "fl_p", "zy": "fl_q", "zxx": "fl_r", "zyy": "fl_t", "zxy": "fl_s"}}
    for fit, names in m.items():
        ks = to.conv_kernels(fit)
        for n, refname in names.items():
            tab, (const, power) = ks[n]
            assert np.array_equal(tab, ref.surfit.all_coefs[refname]), (fit, n)
            for r in (1.0, 2.0, 10.0, 0.3):
                assert const * r**power == ref.surfit._divider_method_coef(r, refname), (fit, n, r)
    print("stencil tables == reference tables")


def run_ref(dem, attrs, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = ref.terrain.get_terrain_attribute(dem, attrs, **kw)
    return out if isinstance(out, list) else [out]


def terrain_T1() -> None:
    rng = np.random.default_rng(42)
    base = rng.normal(size=(20, 20))
    for dt in (np.float32, np.float64):
        for hole in ("nan", "inf"):
            dem = base.astype(dt)
            dem[4, 4:6] = np.nan
            dem[17, 16] = np.nan if hole == "nan" else np.inf
            if hole == "inf":
                dem[10, 2] = -np.inf
            rec = {"dem": dem}
            for fit in ("Horn", "ZevenbergThorne", "Florinsky"):
                for cm in ("geometric", "directional"):
                    for res in (1.0, 2.0, 10.0):
                        attrs = SAH if fit == "Horn" else SURF
                        if fit == "Horn" and cm == "directional":
                            continue
                        outs = run_ref(dem, attrs, resolution=res, surface_fit=fit, curv_method=cm)
                        for a, o in zip(attrs, outs):
                            rec[f"{fit}|{cm}|{res}|{a}"] = o
            np.savez_compressed(os.path.join(OUT, f"terrain_T1_{np.dtype(dt).name}_{hole}.npz"), **rec)


def terrain_T2() -> None:
    rng = np.random.default_rng(7)
    dem = (1000.0 + np.cumsum(np.cumsum(rng.normal(scale=0.05, size=(64, 64)), axis=0), axis=1)).astype(np.float32)
    dem[30:33, 40] = np.nan
    rec = {"dem": dem}
    full = [a for a in SURF if a != "curvature"] + WIN
    for fit in ("ZevenbergThorne", "Florinsky"):
        for cm in ("geometric", "directional"):
            for deg in (True, False):
                for (az, alt, zf) in ((315.0, 45.0, 1.0), (90.0, 10.0, 10.0)):
                    for tri in ("Riley", "Wilson"):
                        if tri == "Wilson" and not (deg and zf == 1.0):
                            continue
                        outs = run_ref(dem, full, resolution=10.0, surface_fit=fit, curv_method=cm, degrees=deg,
                                       hillshade_azimuth=az, hillshade_altitude=alt, hillshade_z_factor=zf,
                                       tri_method=tri)
                        for a, o in zip(full, outs):
                            rec[f"{fit}|{cm}|{int(deg)}|{az}|{alt}|{zf}|{tri}|{a}"] = o
    outs = run_ref(dem, SAH, resolution=10.0, surface_fit="Horn")
    for a, o in zip(SAH, outs):
        rec[f"Horn|geometric|1|315.0|45.0|1.0|Riley|{a}"] = o
    np.savez_compressed(os.path.join(OUT, "terrain_T2_f32.npz"), **rec)


def terrain_T3() -> None:
    """Known-answer DEMs of the reference's own data-free tests, with the reference outputs on them."""
    rec = {}
    dems = {}
    dems["flat"] = np.ones((5, 5), dtype=np.float32)
    dems["ramp_x"] = np.stack([np.ones(5) * i for i in range(5)], axis=1)
    dems["ramp_y"] = np.stack([np.ones(5) * i for i in range(5)], axis=0)
    dems["ramp_xy"] = np.stack([np.arange(0, 5) + i for i in range(5)], axis=1)
    dems["ramp_yx"] = np.stack([np.flip(np.arange(0, 5)) + i for i in range(5)], axis=1)
    dems["v_y_convex_t"] = np.stack([np.array([2, 1, 0, 1, 2]) + i for i in range(5)], axis=0)
    dems["v_x_convex_t"] = np.stack([np.array([2, 1, 0, 1, 2]) + i for i in range(5)], axis=1)
    dems["v_y_concave_t"] = np.stack([np.array([0, 1, 2, 1, 0]) + i for i in range(5)], axis=0)
    dems["v_x_concave_t"] = np.stack([np.array([0, 1, 2, 1, 0]) + i for i in range(5)], axis=1)
    dems["v_y_convex_s"] = np.stack([np.array([2, 1, 0, 1, 2]) + np.linspace(0, 1, 5) for i in range(5)], axis=0)
    dems["v_x_convex_s"] = np.stack([np.array([2, 1, 0, 1, 2]) + np.linspace(0, 1, 5) for i in range(5)], axis=1)
    dems["v_y_concave_s"] = np.stack([np.array([0, 1, 2, 1, 0]) + np.arange(0, 5) for i in range(5)], axis=0)
    dems["v_x_concave_s"] = np.stack([np.array([0, 1, 2, 1, 0]) + np.arange(0, 5) for i in range(5)], axis=1)
    x = np.linspace(-1, 1, 5)
    X, Y = np.meshgrid(x, x)
    dems["saddle"] = X**2 - Y**2
    dems["ridge"] = 0.6 * X + 1.0 * np.exp(-(Y**2))
    dems["trough"] = 0.6 * X - 1.0 * np.exp(-(Y**2))
    dems["doc_south"] = np.repeat(np.arange(3), 3)[::-1].reshape(3, 3)  # terrain.py:269-278
    dems["doc_north"] = np.repeat(np.arange(3), 3).reshape(3, 3)  # terrain.py:717-724
    dems["doc_east"] = np.tile(np.arange(3), (3, 1))  # terrain.py:800-814 (values increasing eastward)
    curvs = [a for a in SURF if a != "curvature"]
    for name, dem in dems.items():
        rec[f"dem|{name}"] = dem
        for fit in ("ZevenbergThorne", "Florinsky"):
            if dem.shape[0] < 5 and fit == "Florinsky":
                continue
            for res in (1.0, 5.0, 10.0):
                outs = run_ref(dem, curvs, resolution=res, surface_fit=fit)
                for a, o in zip(curvs, outs):
                    rec[f"{name}|{fit}|{res}|{a}"] = o
    np.savez_compressed(os.path.join(OUT, "terrain_T3_known_answers.npz"), **rec)


def terrain_T4_T5() -> None:
    rng = np.random.default_rng(3)
    dem_i = rng.integers(0, 500, size=(24, 31)).astype(np.int32)
    rec = {"dem": dem_i}
    full = [a for a in SURF if a != "curvature"] + WIN
    outs = run_ref(dem_i, full, resolution=5.0)
    for a, o in zip(full, outs):
        rec[a] = o
    np.savez_compressed(os.path.join(OUT, "terrain_T4_int32.npz"), **rec)

    dem = rng.normal(loc=50.0, scale=3.0, size=(33, 29)).astype(np.float32)
    dem[12, 20] = np.nan
    rec = {"dem": dem, "dem64": dem.astype(np.float64) + rng.normal(scale=1e-9, size=dem.shape)}
    for w in (3, 5, 7):
        for tri in ("Riley", "Wilson"):
            for key in ("dem", "dem64"):
                outs = run_ref(rec[key], WIN + ["roughness"], window_size=w, tri_method=tri)
                for a, o in zip(WIN + ["roughness"], outs):
                    rec[f"{key}|{w}|{tri}|{a}"] = o
    np.savez_compressed(os.path.join(OUT, "terrain_T5_windows.npz"), **rec)


def terrain_T9() -> None:
    """Rugosity (3x3, needs resolution) and fractal roughness (box counting) -- f2 of SURVEY 8f.  Includes the
    reference's data-free known-answer DEMs (tests/test_terrain/test_window.py:21-89) with its outputs on them."""
    rng = np.random.default_rng(11)
    rec = {}
    base = 1000.0 + np.cumsum(np.cumsum(rng.normal(scale=0.5, size=(40, 44)), axis=0), axis=1)
    for dt in (np.float32, np.float64):
        dem = base.astype(dt)
        dem[5, 7] = np.nan
        dem[20, 30] = np.inf
        dem[30, 10] = -np.inf
        n = np.dtype(dt).name
        rec[f"dem|{n}"] = dem
        for res in (1.0, 10.0, 0.3):
            rec[f"{n}|rugosity|{res}"] = run_ref(dem, ["rugosity"], resolution=res)[0]
        for wf in (5, 7, 9, 13):
            rec[f"{n}|fractal_roughness|{wf}"] = run_ref(dem, ["fractal_roughness"], window_size_fractal=wf)[0]
    # rough, voxel-scale relief (V not saturated at 0 / w)
    dem = rng.uniform(0, 20, size=(30, 33)).astype(np.float32)
    rec["dem|rough"] = dem
    rec["rough|fractal_roughness|13"] = run_ref(dem, ["fractal_roughness"])[0]
    rec["rough|rugosity|2.0"] = run_ref(dem, ["rugosity"], resolution=2.0)[0]
    # known answers
    jen = np.array([[190, 170, 155], [183, 165, 145], [175, 160, 122]], dtype="float32")
    rec["dem|jenness"] = jen
    rec["jenness|rugosity|100.0"] = run_ref(jen, ["rugosity"], resolution=100.0)[0]
    for dh in np.linspace(0.01, 100, 3):
        for res in np.linspace(0.01, 100, 3):
            d = np.array([[1, 1, 1], [1, 1 + dh, 1], [1, 1, 1]], dtype="float64")
            rec[f"dem|pyramid|{dh}"] = d
            rec[f"pyramid|{dh}|rugosity|{res}"] = run_ref(d, ["rugosity"], resolution=float(res))[0]
    line = np.zeros((13, 13)); line[1, 1] = 6.5
    plane = np.zeros((13, 13)); plane[:, 1] = 13
    cube = np.zeros((13, 13)); cube[:, :6] = 13
    for name, d in (("line", line), ("plane", plane), ("cube", cube)):
        rec[f"dem|{name}"] = d
        rec[f"{name}|fractal_roughness|13"] = run_ref(d, ["fractal_roughness"])[0]
    np.savez_compressed(os.path.join(OUT, "terrain_T9_rugosity_fractal.npz"), **rec)


def terrain_T11_numba() -> None:
    """Row a8 of SURVEY section 8: outputs of the reference's `engine="numba"` code path.  numba itself is absent; the
    functions behind @njit are plain Python and run unchanged under _refimport's identity decorator (prange = range), i.e.
    the same IEEE operations in the same order as the compiled loops (no fastmath in the decorators).  Surface fit: the
    derivatives stay float64 (surfit.py:1044) -- and, unlike the SciPy engine, there is no dilation of the non-finite
    mask: NaN comes from the arithmetic alone (0 x Inf, Inf - Inf), so pixels next to a +-Inf hole keep values such as
    slope 90 deg.  Windowed indexes: the callbacks see the window in the DEM dtype (window.py:851), float32 sums
    included (NumPy's pairwise `np.sum` here, a sequential loop under real numba -- recorded for float32 as the
    reference's statements evaluated by NumPy, compared within the float32 rounding noise of that sum)."""
    assert getattr(sys.modules["numba"], "__xdem_oracle_shim__", False) and ref.surfit._HAS_NUMBA
    rng = np.random.default_rng(42)
    base = rng.normal(size=(20, 20))
    rec = {}
    dems = {}
    for dt in (np.float32, np.float64):
        for hole in ("nan", "inf"):
            dem = base.astype(dt)
            dem[4, 4:6] = np.nan
            dem[17, 16] = np.nan if hole == "nan" else np.inf
            if hole == "inf":
                dem[10, 2] = -np.inf
            dems[f"T1_{np.dtype(dt).name}_{hole}"] = dem
    rng2 = np.random.default_rng(7)
    tl = (1000.0 + np.cumsum(np.cumsum(rng2.normal(scale=0.05, size=(28, 30)), axis=0), axis=1)).astype(np.float32)
    tl[12:14, 20] = np.nan
    dems["terrainlike_float32"] = tl
    for name, dem in dems.items():
        rec[f"dem|{name}"] = dem
        for fit in ("Horn", "ZevenbergThorne", "Florinsky"):
            for cm in ("geometric", "directional"):
                if fit == "Horn" and cm == "directional":
                    continue
                for res in ((10.0,) if name.startswith("terrainlike") else (1.0, 2.0, 10.0)):
                    attrs = SAH if fit == "Horn" else SURF
                    outs = run_ref(dem, attrs, resolution=res, surface_fit=fit, curv_method=cm, engine="numba")
                    for a, o in zip(attrs, outs):
                        rec[f"{name}|{fit}|{cm}|{res}|{a}"] = o
    # direct call of the engine boundary, radians, explicit out_dtype
    d = dems["T1_float32_nan"]
    for od in (np.float32, np.float64):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = ref.surfit._get_surface_attributes(dem=d, resolution=2.0, surface_attributes=SURF, out_dtype=od,
                                                     surface_fit="Florinsky", curv_method="geometric", engine="numba")
        rec[f"boundary|Florinsky|{np.dtype(od).name}"] = out
    # windowed indexes through the numba path
    win_all = WIN + ["roughness"]
    for name in ("T1_float32_nan", "T1_float64_inf", "terrainlike_float32"):
        dem = dems[name]
        for w in (3, 5):
            for tri in ("Riley", "Wilson"):
                outs = run_ref(dem, win_all, window_size=w, tri_method=tri, engine="numba")
                for a, o in zip(win_all, outs):
                    rec[f"{name}|win|{w}|{tri}|{a}"] = o
        rec[f"{name}|rugosity|2.0"] = run_ref(dem, ["rugosity"], resolution=2.0, engine="numba")[0]
        rec[f"{name}|fractal_roughness|13"] = run_ref(dem, ["fractal_roughness"], engine="numba")[0]
    np.savez_compressed(os.path.join(OUT, "terrain_T11_numba_engine.npz"), **rec)


def terrain_T12_engine_boundary() -> None:
    """SURVEY 8b rows 1-2 called DIRECTLY: the reference's `surfit._get_surface_attributes` and `window._get_windowed_indexes`
    (the functions the HIP engine replaces), both of their engines -- the Numba one through the identity-njit shim --, explicit
    `out_dtype`, radians, UNCLIPPED hillshade (the caller's post-steps, terrain.py:586-596, are not part of the engine)."""
    rng = np.random.default_rng(12)
    base = 1000.0 + np.cumsum(np.cumsum(rng.normal(scale=0.08, size=(30, 34)), axis=0), axis=1)
    rec = {}
    for dt in (np.float32, np.float64):
        dem = base.astype(dt)
        dem[9:11, 30] = np.nan
        dem[25, 5] = np.nan
        name = np.dtype(dt).name
        rec[f"dem|{name}"] = dem
        for engine in ("scipy", "numba"):
            for fit in ("Horn", "ZevenbergThorne", "Florinsky"):
                for cm in ("geometric", "directional"):
                    if fit == "Horn" and cm == "directional":
                        continue
                    attrs = SAH if fit == "Horn" else [a for a in SURF if a != "curvature"]
                    main = fit == "Florinsky" and cm == "geometric"   # (the full product only for the default fit: the file stays small)
                    for od in ((np.float32, np.float64) if main else (dt,)):
                        for hs in (((315.0, 45.0, 1.0), (120.0, 5.0, 4.0)) if main else ((315.0, 45.0, 1.0),)):   # the second one drives hillshade below 0 (unclipped here)
                            with warnings.catch_warnings():
                                warnings.simplefilter("ignore")
                                out = ref.surfit._get_surface_attributes(dem=dem, resolution=5.0, surface_attributes=attrs, out_dtype=od, surface_fit=fit,
                                                                         curv_method=cm, engine=engine, hillshade_azimuth=hs[0],
                                                                         hillshade_altitude=hs[1], hillshade_z_factor=hs[2])
                            assert out.dtype == np.dtype(od)
                            rec[f"surf|{name}|{engine}|{fit}|{cm}|{np.dtype(od).name}|{hs[0]}"] = out
            for w in (3, 5):
                for tri in ("Riley", "Wilson"):
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        out = ref.window._get_windowed_indexes(dem=dem, window_size=w, windowed_indexes=WIN + ["roughness"], resolution=5.0,
                                                               out_dtype=dt, tri_method=tri, engine=engine)
                    rec[f"win|{name}|{engine}|{w}|{tri}"] = out
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                rec[f"rug|{name}|{engine}"] = ref.window._get_windowed_indexes(dem=dem, window_size=3, windowed_indexes=["rugosity"], resolution=5.0,
                                                                               out_dtype=dt, engine=engine)
                rec[f"frac|{name}|{engine}"] = ref.window._get_windowed_indexes(dem=dem, window_size=13, windowed_indexes=["fractal_roughness"],
                                                                                resolution=5.0, out_dtype=dt, engine=engine)
    np.savez_compressed(os.path.join(OUT, "terrain_T12_engine_boundary.npz"), **rec)


def terrain_T10() -> None:
    """Texture shading (freq.py:63-148), SURVEY 8f-4: reference outputs for float32 / float64 DEMs with holes, several alpha,
    FFT lengths below and above 1024 (power-of-two and 7-smooth padding), plus the data-free cases of test_freq.py."""
    rng = np.random.default_rng(19)
    rec = {}
    base = 800.0 + np.cumsum(np.cumsum(rng.normal(scale=0.3, size=(70, 90)), axis=0), axis=1)
    for dt in (np.float32, np.float64):
        dem = base.astype(dt)
        dem[10:13, 20] = np.nan
        dem[50, 60] = np.nan
        n = np.dtype(dt).name
        rec[f"dem|{n}"] = dem
        for alpha in (0.0, 0.5, 0.8, 1.5, 2.0):
            rec[f"{n}|{alpha}"] = run_ref(dem, ["texture_shading"], texture_alpha=alpha)[0]
    wide = (100.0 + np.cumsum(rng.normal(scale=0.5, size=(9, 1030)), axis=1)).astype(np.float32)  # FFT lengths 16 x 1050
    rec["dem|wide"] = wide
    rec["wide|0.8"] = run_ref(wide, ["texture_shading"])[0]
    flat = np.full((16, 16), 5.0, dtype=np.float32)
    rec["dem|flat"] = flat
    rec["flat|0.8"] = run_ref(flat, ["texture_shading"])[0]
    allnan = np.full((8, 8), np.nan, dtype=np.float32)
    rec["dem|allnan"] = allnan
    rec["allnan|0.8"] = run_ref(allnan, ["texture_shading"])[0]
    np.savez_compressed(os.path.join(OUT, "terrain_T10_texture.npz"), **rec)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["terrain", "nk", "vario", "binning", "patches", "conv"]
    if "terrain" in which:
        _check_tables()
        terrain_T1()
        terrain_T2()
        terrain_T3()
        terrain_T4_T5()
        terrain_T9()
        terrain_T10()
        terrain_T11_numba()
        terrain_T12_engine_boundary()
        print("terrain fixtures written")
    if "nk" in which:
        import gen_golden_nk

        gen_golden_nk.main(ref, OUT)
    if "vario" in which:
        import gen_golden_vario

        gen_golden_vario.main(ref, OUT)
    if "binning" in which:
        import gen_golden_binning

        gen_golden_binning.main(ref, OUT)
    if "patches" in which:
        import gen_golden_patches

        gen_golden_patches.main(ref, OUT)
    if "conv" in which:
        import gen_golden_conv

        gen_golden_conv.main(ref, OUT)
