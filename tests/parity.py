"""Parity metric shared by the GPU tests and the host-simulation tests.

Bar (BASELINE.json north_star): NaN/nodata masks bit-exact; float32 outputs within 1e-6 relative of the
reference recipe.  "Relative" is taken against max(|ref|, scale_attr) where scale_attr is the 99th
percentile of |ref| over the raster -- the same magnitude-scaled comparison the reference's own tests use
(tests/test_terrain/test_terrain.py:88-102) -- because several attributes pass through zero (aspect wraps,
curvatures change sign) where a pure relative error is meaningless.  The tests additionally report the
fraction of bit-identical pixels.
"""
import numpy as np

RTOL = 1e-6


def compare(got: np.ndarray, ref: np.ndarray, rtol: float = RTOL):
    assert got.shape == ref.shape and got.dtype == ref.dtype, (got.shape, ref.shape, got.dtype, ref.dtype)
    nan_equal = np.array_equal(np.isnan(got), np.isnan(ref))
    fin = np.isfinite(ref) & np.isfinite(got)
    inf_equal = np.array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)])
    if fin.any():
        r = ref[fin].astype(np.float64)
        g = got[fin].astype(np.float64)
        scale = np.percentile(np.abs(r), 99)
        scale = scale if scale > 0 else 1.0
        err = np.abs(g - r) / np.maximum(np.abs(r), scale)
        maxerr = float(err.max())
        exact = float(np.mean(g == r))
    else:
        maxerr, exact = 0.0, 1.0
    return {"nan_equal": nan_equal, "inf_equal": inf_equal, "max_scaled_err": maxerr, "bit_exact_frac": exact}


def assert_parity(got, ref, name="", rtol: float = RTOL, min_exact: float = 0.0):
    c = compare(got, ref, rtol)
    assert c["nan_equal"], f"{name}: NaN mask differs"
    assert c["inf_equal"], f"{name}: +-Inf positions/values differ"
    assert c["max_scaled_err"] <= rtol, f"{name}: scaled error {c['max_scaled_err']:.3e} > {rtol:g}"
    assert c["bit_exact_frac"] >= min_exact, f"{name}: only {c['bit_exact_frac']:.6f} bit-exact"
    return c


# ---- true relative error (round 2) ---------------------------------------------------------------------
# The scaled metric above forgives errors on small values in a raster of large ones.  `compare_true` does not: every
# finite pixel must satisfy |got - ref| <= rtol * |ref|, except where both numbers are float64 rounding noise of the
# reference's own accumulation -- results whose exact value is 0 (curvature of a planar ramp, TPI of flat ground) come
# out of the reference as ~1e-15 residues that depend on the order of its additions.  `noise_floor` bounds that noise
# from the DEM: 64 * eps(float64) * attribute scale, with the scale of a curvature = 100 * 4 max|z| / res^2, of a
# first-derivative attribute max|z| / res, of TPI / TRI / roughness max|z|.  It is ~1e-11 of real curvature values.
_EPS64 = float(np.finfo(np.float64).eps)


def noise_floor(attr: str, dem: np.ndarray, resolution: float = 1.0) -> float:
    fin = np.isfinite(dem)
    zmax = float(np.max(np.abs(dem[fin]))) if fin.any() else 0.0
    if "curvature" in attr:
        scale = 100.0 * 4.0 * zmax / (resolution * resolution)
    elif attr in ("slope", "aspect", "hillshade"):
        scale = 0.0  # first-derivative attributes are compared relatively everywhere (residues are reproduced exactly)
    else:
        scale = zmax
    return 64.0 * _EPS64 * scale


def ulp_distance(got: np.ndarray, ref: np.ndarray) -> np.ndarray:
    it = np.int32 if got.dtype == np.float32 else np.int64
    a = got.view(it).astype(np.int64)
    b = ref.view(it).astype(np.int64)
    sign = np.int64(-(2**31)) if got.dtype == np.float32 else np.int64(-(2**63))
    a = np.where(a < 0, sign - a, a)
    b = np.where(b < 0, sign - b, b)
    return np.abs(a - b)


def compare_true(got: np.ndarray, ref: np.ndarray, floor: float = 0.0):
    """NaN / Inf masks, worst TRUE relative error outside the noise floor, ulp histogram (0, 1, 2, 3-4, 5-8, >8)."""
    assert got.shape == ref.shape and got.dtype == ref.dtype, (got.shape, ref.shape, got.dtype, ref.dtype)
    nan_equal = np.array_equal(np.isnan(got), np.isnan(ref))
    fin = np.isfinite(ref) & np.isfinite(got)
    inf_equal = np.array_equal(got[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)])
    # "excused" = share of the finite pixels that DIFFER from the reference and were not judged because the difference lies
    # inside the float64 noise floor (0 on outputs of real relief; up to all of them on fixtures whose exact answer is 0)
    out = {"nan_equal": nan_equal, "inf_equal": inf_equal, "max_rel": 0.0, "exact": 1.0, "hist": [1.0, 0, 0, 0, 0, 0],
           "max_ulp": 0, "n": int(fin.sum()), "excused": 0.0, "n_signal": 0}
    if fin.any():
        r = ref[fin].astype(np.float64)
        g = got[fin].astype(np.float64)
        err = np.abs(g - r)
        judged = err > floor
        # pixels whose REFERENCE value stands above the floor: a result of exactly 0 there has err = |ref| > floor and is judged
        # (and fails) -- the floor can only excuse results on pixels whose reference value is itself noise
        out["n_signal"] = int(np.sum(np.abs(r) > floor))
        out["excused"] = float(np.mean((err > 0) & ~judged))
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.where(judged, err / np.abs(r), 0.0)
        out["max_rel"] = float(np.max(rel))
        d = ulp_distance(got[fin], ref[fin])
        d = np.where(judged, d, 0)
        out["max_ulp"] = int(d.max())
        out["exact"] = float(np.mean(d == 0))
        out["hist"] = [float(np.mean(d == 0)), float(np.mean(d == 1)), float(np.mean(d == 2)),
                       float(np.mean((d > 2) & (d <= 4))), float(np.mean((d > 4) & (d <= 8))), float(np.mean(d > 8))]
    return out


def assert_parity_true(got, ref, name="", floor: float = 0.0, rtol: float = RTOL, min_exact: float = 0.0, max_excused: float = 1.0):
    """``max_excused``: largest share of differing pixels the noise floor may excuse (1.0 = no limit: fixtures whose exact
    answer is 0 -- planar ramps, flat ground -- consist of nothing else; rasters with real relief pass 0.05)."""
    c = compare_true(got, ref, floor)
    assert c["nan_equal"], f"{name}: NaN mask differs"
    assert c["inf_equal"], f"{name}: +-Inf positions/values differ"
    assert c["max_rel"] <= rtol, f"{name}: true relative error {c['max_rel']:.3e} > {rtol:g} (max {c['max_ulp']} ulp)"
    assert c["exact"] >= min_exact, f"{name}: only {c['exact']:.6f} bit-exact (< {min_exact})"
    # (float64 planes: a difference in the last bits of a double IS float64 rounding noise -- of the order of the floor by
    # construction -- so the share is recorded but bounded only for float32 planes, whose ulp lies far above the floor)
    if got.dtype == np.float32:
        assert c["excused"] <= max_excused, f"{name}: the noise floor excuses {c['excused']:.4f} of the pixels (> {max_excused})"
    return c


# Attributes whose float32 result is, by construction of the kernel, the reference's float64 evaluation rounded once
# (>= 99.9 % of pixels bit-identical on terrain-like rasters).  Round 3: the lean tail of the specialised float32 kernels
# spends the 1e-6 budget on float32 scale factors (curvatures, hillshade: a few float32 roundings, <= 16 ulp, measured maxima
# 3e-7 .. 6e-7 true relative); what stays bit-exact are the planes that never leave float64 before their single rounding.
EXACT_ATTRS = {"curvature", "topographic_position_index", "roughness"}
# ... and with the mixed tail of round 2 (option "terrain_math" = 0, the runtime-mask kernels, tail=0 in the host simulator):
EXACT_ATTRS_MIXED = EXACT_ATTRS | {"hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
                                   "flowline_curvature", "max_curvature", "min_curvature"}


def check_attribute(got, ref, attr, dem, resolution, name="", exact_frac=0.999, max_ulp_f32_math=16, exact_attrs=None,
                    max_excused=0.05):
    """The round-2 parity bar for one attribute plane: masks bit-exact, TRUE relative error <= 1e-6 outside the float64
    noise floor (which may excuse at most ``max_excused`` of the pixels: these are rasters with relief), and for float32
    planes either the bit-exact share (EXACT_ATTRS) or an ulp bound."""
    floor = noise_floor(attr, dem, resolution)
    c = assert_parity_true(got, ref, name or attr, floor=floor, max_excused=max_excused)
    if got.dtype == np.float32 and c["n"] >= 1000:
        if attr in (EXACT_ATTRS if exact_attrs is None else exact_attrs):
            assert c["exact"] >= exact_frac, f"{name or attr}: only {c['exact']:.5f} bit-exact (< {exact_frac})"
        else:
            assert c["max_ulp"] <= max_ulp_f32_math, f"{name or attr}: {c['max_ulp']} ulp (> {max_ulp_f32_math})"
    return c
