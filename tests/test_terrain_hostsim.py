"""CPU check of the GPU kernel's own math: xdem_amd/csrc/terrain_math.h (the per-column marcher and the float64
attribute formulas the HIP kernel instantiates) compiled for the host by tests/hostsim and compared with the oracle.
Guards the stencil algebra, window rotation, NaN poisoning and polynomial accuracy on machines without a GPU."""
import numpy as np
import pytest

import terrain_oracle as to
from hostsim_util import hostsim_terrain
from parity import assert_parity, assert_parity_true, check_attribute, noise_floor

FULL = ["slope", "aspect", "hillshade", "curvature", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]
HOT11 = [a for a in FULL if a != "curvature"]


def _dem(shape=(140, 300), seed=3, dtype=np.float32):
    from xdem_amd.synth import fbm_numpy

    dem = fbm_numpy(shape, seed=seed, dtype=dtype)
    dem[5, 7] = np.nan
    dem[60, 250:262] = np.inf
    dem[-1, -1] = -np.inf
    return dem


@pytest.mark.parametrize("fit", ["Florinsky", "ZevenbergThorne", "Horn"])
@pytest.mark.parametrize("cm", ["geometric", "directional"])
def test_generic_kernel_math(fit, cm):
    if fit == "Horn" and cm == "directional":
        pytest.skip("Horn has no curvatures")
    dem = _dem()
    attrs = FULL if fit != "Horn" else ["slope", "aspect", "hillshade", "topographic_position_index", "terrain_ruggedness_index"]
    kw = dict(resolution=10.0, surface_fit=fit, curv_method=cm, hillshade_z_factor=2.0, tri_method="Wilson", degrees=False)
    got = hostsim_terrain(dem, attrs, **kw)
    ref = to.terrain_attributes(dem, attrs, **kw)
    for a, g, r in zip(attrs, got, ref):
        check_attribute(g, r, a, dem, 10.0, f"{fit}/{cm}/{a}")


@pytest.mark.parametrize("fit,attrs", [("Florinsky", HOT11), ("ZevenbergThorne", HOT11),
                                       ("Horn", ["slope", "aspect", "hillshade", "topographic_position_index",
                                                 "terrain_ruggedness_index"])])
def test_specialised_kernel_math(fit, attrs):
    """The compile-time specialised instantiations (what bench.py runs)."""
    dem = _dem(seed=8)
    got = hostsim_terrain(dem, attrs, resolution=10.0, surface_fit=fit)
    ref = to.terrain_attributes(dem, attrs, resolution=10.0, surface_fit=fit)
    for a, g, r in zip(attrs, got, ref):
        check_attribute(g, r, a, dem, 10.0, f"{fit}/{a}", exact_frac=0.9999)


@pytest.mark.parametrize("fit", ["Florinsky", "ZevenbergThorne"])
def test_specialised_directional_kernel_math(fit):
    """Round 5: the eleven planes with curv_method="directional" have their own compile-time specialisation (lean tail)."""
    dem = _dem(seed=9)
    got = hostsim_terrain(dem, HOT11, resolution=10.0, surface_fit=fit, curv_method="directional")
    ref = to.terrain_attributes(dem, HOT11, resolution=10.0, surface_fit=fit, curv_method="directional")
    for a, g, r in zip(HOT11, got, ref):
        check_attribute(g, r, a, dem, 10.0, f"{fit}/directional/{a}", exact_frac=0.9999)
    for fname in ("terrain_T1_float32_nan.npz",):   # and the reference's own T1 outputs for that method, float32
        import os

        from conftest import GOLDEN

        z = np.load(os.path.join(GOLDEN, fname))
        for res in ("1.0", "2.0", "10.0"):
            refs = [z[f"{fit}|directional|{res}|{a}"] for a in HOT11[:9]]
            # the specialised kernel needs the full mask: TPI / TRI ride along
            got = hostsim_terrain(z["dem"], HOT11, resolution=float(res), surface_fit=fit, curv_method="directional")
            for a, g, r in zip(HOT11[:9], got, refs):
                assert_parity_true(g, r, f"T1/{fit}/directional/{res}/{a}", floor=noise_floor(a, z["dem"], float(res)))


def test_float64_and_tile_boundaries():
    dem = _dem((70, 530), seed=5, dtype=np.float64)  # three column tiles, three row tiles of 32
    got = hostsim_terrain(dem, HOT11, resolution=2.0)
    ref = to.terrain_attributes(dem, HOT11, resolution=2.0)
    for a, g, r in zip(HOT11, got, ref):
        assert_parity(g, r, a)


def test_reference_known_answers_incl_flat_residue():
    """T3: the reference's data-free known-answer DEMs with the reference's own outputs (flat, ramps, V shapes, saddle...),
    TRUE relative error.  Exactly cancelling derivative sums are recomputed in the reference's accumulation order, so the
    flat Florinsky window carries the reference's residue: slope 2.5e-15, aspect 198.43494 deg (round 1 returned 0 / 0)."""
    import os

    from conftest import GOLDEN

    z = np.load(os.path.join(GOLDEN, "terrain_T3_known_answers.npz"))
    n = 0
    for key in z.files:
        if key.startswith("dem|"):
            continue
        name, fit, res, attr = key.split("|")
        dem = z["dem|" + name]
        dem = dem.astype(np.float32) if dem.dtype.kind in "iu" else dem
        ref = z[key]
        got = hostsim_terrain(dem, [attr], resolution=float(res), surface_fit=fit, out_dtype=ref.dtype)[0]
        assert_parity_true(got, ref, key, floor=noise_floor(attr, dem, float(res)))
        n += 1
    assert n > 900
    asp = hostsim_terrain(z["dem|flat"], ["aspect"], resolution=1.0, surface_fit="Florinsky")[0]
    assert asp[2, 2] == np.float32(198.43494)


@pytest.mark.parametrize("fname", ["terrain_T1_float32_nan.npz", "terrain_T1_float64_inf.npz"])
def test_reference_noise_fixtures(fname):
    """T1 (normal noise with NaN / Inf holes, every fit / curvature method / resolution): masks exact, TRUE relative error."""
    import os

    from conftest import GOLDEN

    z = np.load(os.path.join(GOLDEN, fname))
    dem = z["dem"]
    for key in z.files:
        if key == "dem":
            continue
        fit, cm, res, attr = key.split("|")
        ref = z[key]
        got = hostsim_terrain(dem, [attr], resolution=float(res), surface_fit=fit, curv_method=cm, out_dtype=ref.dtype)[0]
        assert_parity_true(got, ref, key, floor=noise_floor(attr, dem, float(res)))


def test_T11_numba_engine_reference_fixtures():
    """Row a8: outputs of the reference's own numba-engine code (tests/golden/terrain_T11_numba_engine.npz) against the kernels'
    math for engine="numba" -- float64-input marcher on the widened DEM + the +-Inf rule of terrain_nonfinite.h -- on the CPU.
    Masks bit-exact (the values next to +-Inf pixels included), TRUE relative error <= 1e-6."""
    import os

    from conftest import GOLDEN

    z = np.load(os.path.join(GOLDEN, "terrain_T11_numba_engine.npz"))
    n = n_inf = 0
    for key in z.files:
        parts = key.split("|")
        if parts[0] in ("dem", "boundary") or len(parts) != 5 or parts[1] == "win":
            continue
        name, fit, cm, res, attr = parts
        dem = z["dem|" + name]
        ref = z[key]
        got = hostsim_terrain(dem, [attr], resolution=float(res), surface_fit=fit, curv_method=cm, out_dtype=ref.dtype,
                              engine="numba")[0]
        assert_parity_true(got, ref, key, floor=noise_floor(attr, dem, float(res)))
        if name.endswith("_inf"):
            sel = to._window_invalid(dem, 5 if fit == "Florinsky" else 3) & ~np.isnan(ref)
            n_inf += int(sel.sum())
            if ref.dtype == np.float32:
                assert np.array_equal(got[sel], ref[sel]), (key, got[sel], ref[sel])
        n += 1
    assert n > 500 and n_inf > 100


@pytest.mark.parametrize("fit", ["Florinsky", "Horn", "ZevenbergThorne"])
@pytest.mark.parametrize("attrs", [["slope"], ["slope", "aspect"], ["hillshade"], ["slope", "aspect", "hillshade"]])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_first_derivative_sets_nonfinite_windows(fit, attrs, dtype):
    """The small attribute sets (no curvature, no windowed index) have no sum over the whole window to take their NaN rule from; for the
    Florinsky fit the two derivative sums and the centre row's partial stand in for it (terrain_math.h, `poison`).  NaN, +Inf, -Inf,
    runs of them, pixels next to the raster's edge, a hole of exactly one pixel in the window's centre / corner: the NaN mask must be the
    oracle's exactly -- an output is NaN iff its window holds a non-finite pixel or leaves the raster -- and the values agree."""
    from xdem_amd.synth import fbm_numpy

    dem = fbm_numpy((90, 200), seed=21, dtype=dtype)
    dem[5, 7] = np.nan
    dem[20, 100] = np.inf
    dem[21, 101] = -np.inf            # +Inf and -Inf in one window: sums that cancel to NaN
    dem[40, 50:62] = np.inf           # a run: every column of some windows
    dem[41, 50:62] = -np.inf
    dem[60:66, 150] = np.nan          # a column run: every row of some windows
    dem[70, 30] = -np.inf
    dem[2, 2] = np.inf                # within the halo of the raster's corner
    dem[-1, -1] = np.nan
    dem[80, 120] = 3.0e38             # huge but finite: must NOT poison anything (float32 max ~3.4e38)
    kw = dict(resolution=10.0, surface_fit=fit)
    got = hostsim_terrain(dem, attrs, **kw)
    ref = to.terrain_attributes(dem, attrs, **kw)
    for a, g, r in zip(attrs, got, ref):
        assert np.array_equal(np.isnan(g), np.isnan(r)), f"{fit}/{a}: NaN mask differs at {np.argwhere(np.isnan(g) != np.isnan(r))[:5]}"
        far = np.ones(dem.shape, dtype=bool)
        far[74:87, 114:127] = False    # (windows that hold the 3e38 pixel: slopes of 90 deg minus rounding, not a parity case)
        ok = np.isfinite(r) & far
        check_attribute(np.where(ok, g, np.nan), np.where(ok, r, np.nan), a, dem, 10.0, f"{fit}/{a}")
