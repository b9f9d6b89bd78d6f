"""CPU oracle for the terrain stencil path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the shipped package ``xdem_amd`` never does (its ops fail loudly when the HIP library is
missing instead of falling back to anything here).

This is a NumPy restatement (own code, written from the formulas) of the reference's *SciPy engine*
for the terrain attributes, following the reference's numeric recipe step by step so that results are
bit-comparable:

* coefficient stencils: ``xdem/terrain/surfit.py:61-304`` (Zevenbergen-Thorne, Horn, Florinsky tables
  and their resolution dividers), applied as a *true convolution* with NaN outside the array exactly
  like ``scipy.ndimage.convolve(..., mode="constant", cval=nan)`` called from
  ``xdem/spatialstats.py:2512-2525``: double accumulation in row-major order of the flipped kernel's
  non-zero weights, result rounded to the *input* dtype, then stored as float64
  (``spatialstats.py:2575``);
* attribute formulas in float64 from those coefficients: ``xdem/terrain/surfit.py:590-943``;
* NaN rule (any non-finite pixel in the full 3x3 / 5x5 window, or window leaving the array, gives
  NaN for every attribute): ``xdem/terrain/surfit.py:1185-1192``;
* windowed indexes TPI / TRI (Riley, Wilson) on a float64 window: ``xdem/terrain/window.py:67-252``
  through ``scipy.ndimage.generic_filter(mode="constant", cval=nan)``;
* post-processing (rad2deg in the output dtype, hillshade clip): ``xdem/terrain/terrain.py:586-596``.

Parity of this oracle is PINNED: ``tests/test_oracle_golden.py`` compares it bit-for-bit with golden
vectors produced by running the reference itself (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).
"""
from __future__ import annotations

import numpy as np

SURFACE_ATTRIBUTES = (
    "slope",
    "aspect",
    "hillshade",
    "curvature",
    "profile_curvature",
    "tangential_curvature",
    "planform_curvature",
    "flowline_curvature",
    "max_curvature",
    "min_curvature",
)
WINDOW_ATTRIBUTES = ("topographic_position_index", "terrain_ruggedness_index", "roughness", "rugosity")
FRACTAL_ATTRIBUTES = ("fractal_roughness",)
FITS = {"horn": 0, "zevenbergthorne": 1, "florinsky": 2}

# ---------------------------------------------------------------------------------------------
# Stencil tables, generated from their structure rather than tabulated (surfit.py:61-252).
# Names follow the derivative they estimate; "conv" tables are the reference's convolution kernels.
# ---------------------------------------------------------------------------------------------
_u5 = np.array([-2, -1, 0, 1, 2])
_c5 = np.array([2, -1, -2, -1, 2])
_alpha = np.array([44, 62, 68, 62, 44])
_beta = np.array([-31, 5, 17, 5, -31])
_a5 = np.array([0, -1, 0, 1, 0])
_b5 = np.array([-1, 0, 0, 0, 1])


def conv_kernels(surface_fit: str) -> dict[str, tuple[np.ndarray, str]]:
    """Integer convolution kernels per derivative and the divider rule ("res" power and constant)."""
    fit = surface_fit.lower()
    if fit == "horn":  # surfit.py:145-157, dividers 291-292
        zy = np.outer([1, 0, -1], [1, 2, 1])
        zx = np.outer([1, 2, 1], [-1, 0, 1])
        return {"zx": (zx, (8, 1)), "zy": (zy, (8, 1))}
    if fit == "zevenbergthorne":  # surfit.py:93-129, dividers 285-289
        e = np.zeros((3, 3), int)
        zyy = e.copy()
        zyy[:, 1] = [1, -2, 1]
        zxx = e.copy()
        zxx[1, :] = [1, -2, 1]
        zxy = np.outer([-1, 0, 1], [1, 0, -1])
        zy = e.copy()
        zy[:, 1] = [1, 0, -1]
        zx = e.copy()
        zx[1, :] = [-1, 0, 1]
        return {"zx": (zx, (2, 1)), "zy": (zy, (2, 1)), "zxx": (zxx, (1, 2)), "zyy": (zyy, (1, 2)), "zxy": (zxy, (4, 2))}
    if fit == "florinsky":  # surfit.py:204-252, dividers 297-301
        zxx = np.tile(_c5, (5, 1))
        zyy = zxx.T.copy()
        zxy = -np.outer(_u5, _u5)
        zx = np.outer(_alpha, _a5) + np.outer(_beta, _b5)
        zy = -zx.T
        return {
            "zx": (zx, (420, 1)),
            "zy": (zy, (420, 1)),
            "zxx": (zxx, (35, 2)),
            "zyy": (zyy, (35, 2)),
            "zxy": (zxy, (100, 2)),
        }
    raise ValueError(surface_fit)


def _convolve_nan_const(dem: np.ndarray, kernel_f64: np.ndarray) -> np.ndarray:
    """scipy.ndimage.convolve(dem, kernel, mode="constant", cval=nan) restated.

    Convolution = correlation with the kernel flipped on both axes; the accumulator is a double that
    starts at 0 and adds ``w * value`` for the non-zero weights in row-major order of the flipped
    kernel; the result is cast to the input dtype.
    """
    k = kernel_f64[::-1, ::-1]
    m = k.shape[0]
    h = m // 2
    H, W = dem.shape
    pad = np.full((H + 2 * h, W + 2 * h), np.nan, dtype=np.float64)
    pad[h : h + H, h : h + W] = dem
    acc = np.zeros((H, W), dtype=np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        for a in range(m):
            for b in range(m):
                w = k[a, b]
                if abs(w) > np.finfo(np.float64).eps:
                    acc += w * pad[a : a + H, b : b + W]
    return acc.astype(dem.dtype)


def _convolve_numba_loop(dem: np.ndarray, kernel_f64: np.ndarray) -> np.ndarray:
    """The per-pixel loop of the reference's Numba engine (surfit.py:948-971 on the NaN-padded DEM of 1275-1282),
    vectorised over pixels: a float64 accumulator adds ``value * weight`` for EVERY tap of the flipped kernel in row-major
    window order -- zero weights included, so 0 x Inf = NaN and Inf - Inf = NaN arise from the arithmetic itself -- and
    the sum is NOT rounded to the DEM dtype (out_dtype=np.float64 at surfit.py:1044)."""
    k = kernel_f64[::-1, ::-1]
    m = k.shape[0]
    h = m // 2
    H, W = dem.shape
    pad = np.full((H + 2 * h, W + 2 * h), np.nan, dtype=dem.dtype)
    pad[h : h + H, h : h + W] = dem
    acc = np.zeros((H, W), dtype=np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        for a in range(m):
            for b in range(m):
                acc += pad[a : a + H, b : b + W] * k[a, b]
    return acc


def surface_coefficients(dem: np.ndarray, resolution: float, surface_fit: str, names: list[str],
                         engine: str = "scipy") -> dict[str, np.ndarray]:
    """Per-pixel derivative estimates: float64 arrays holding input-dtype-rounded values (SciPy engine) or the unrounded
    float64 sums of the Numba engine's loop."""
    ks = conv_kernels(surface_fit)
    out = {}
    for n in names:
        tab, (const, power) = ks[n]
        kern = tab.astype(np.float64)
        kern /= const * resolution**power  # surfit.py:373-377: integer table / divider, in double
        if engine == "numba":
            out[n] = _convolve_numba_loop(dem, kern)
        else:
            out[n] = _convolve_nan_const(dem, kern).astype(np.float64)
    return out


def _window_invalid(dem: np.ndarray, w: int) -> np.ndarray:
    """True where the w x w window holds a non-finite value or leaves the array (surfit.py:1185-1192 + cval=nan)."""
    h = w // 2
    H, W = dem.shape
    bad = np.ones((H + 2 * h, W + 2 * h), dtype=bool)
    bad[h : h + H, h : h + W] = ~np.isfinite(dem)
    out = np.zeros((H, W), dtype=bool)
    for a in range(w):
        for b in range(w):
            out |= bad[a : a + H, b : b + W]
    return out


def surface_attributes(
    dem: np.ndarray,
    resolution: float,
    surface_attributes: list[str],
    out_dtype=np.float32,
    surface_fit: str = "Florinsky",
    curv_method: str = "geometric",
    hillshade_altitude: float = 45.0,
    hillshade_azimuth: float = 315.0,
    hillshade_z_factor: float = 1.0,
    engine: str = "scipy",
) -> np.ndarray:
    """Oracle of ``_get_surface_attributes`` (surfit.py:1197-1305). Output (n,H,W), radians.  ``engine="scipy"``: the
    default recipe; ``engine="numba"``: surfit.py:948-1088, 1270-1303 -- float64 derivatives from the explicit loop, the
    same formulas, and NO dilated non-finite mask (NaN only where the arithmetic produces it; pinned by
    tests/golden/terrain_T11_numba_engine.npz, outputs of the reference's own numba-engine code)."""
    fit_id = FITS[surface_fit.lower()]
    directional = curv_method.lower() == "directional"
    want = set(surface_attributes)
    need2 = bool(want & {"curvature", "profile_curvature", "tangential_curvature", "planform_curvature",
                         "flowline_curvature", "max_curvature", "min_curvature"}) and fit_id != 0
    need_grad = bool(want & {"slope", "aspect", "hillshade"}) or bool(
        want - {"slope", "aspect", "hillshade", "curvature"}
    )
    names = (["zx", "zy"] if need_grad else []) + (["zxx", "zyy"] if need2 else [])
    if need2 and (want - {"slope", "aspect", "hillshade", "curvature"}):
        names.append("zxy")
    C = surface_coefficients(dem, resolution, surface_fit, names, engine)

    H, W = dem.shape
    res = np.full((len(surface_attributes), H, W), np.nan, dtype=out_dtype)

    def put(name, val):
        if name in want:
            res[surface_attributes.index(name)] = val

    with np.errstate(all="ignore"):
        if need_grad:
            zx, zy = C["zx"], C["zy"]
            g2 = zx**2 + zy**2
            opg = 1 + zx**2 + zy**2  # the reference spells "1 + zx^2 + zy^2" left to right: (1 + zx^2) + zy^2
        if want & {"slope", "hillshade"}:
            slope = np.arctan(g2**0.5)
            put("slope", slope)
        if want & {"aspect", "hillshade"}:
            aspect = (-np.arctan2(-zx, zy)) % (2 * np.pi)
            put("aspect", aspect)
        if "hillshade" in want:
            smap = np.arctan(np.tan(slope) * hillshade_z_factor) if hillshade_z_factor != 1.0 else slope
            az = np.deg2rad(360 - hillshade_azimuth)
            alt = np.deg2rad(hillshade_altitude)
            put("hillshade", 1.5 + 254 * (np.sin(alt) * np.cos(smap) + np.cos(alt) * np.sin(smap) * np.sin(az - aspect)))
        if fit_id != 0:
            if "curvature" in want:
                put("curvature", -2.0 * (C["zxx"] + C["zyy"]) * 100)
            if want - {"slope", "aspect", "hillshade", "curvature"}:
                zxx, zyy, zxy = C["zxx"], C["zyy"], C["zxy"]
                flat = g2 == 0.0
                one = np.array([0.0])
                # slope-line (profile) and contour (tangential / planform) second directional derivatives
                num_prof = -(zxx * zx**2 + 2 * zxy * zx * zy + zyy * zy**2)
                num_tan = -(zxx * zy**2 - 2 * zxy * zx * zy + zyy * zx**2)
                num_flow = zx * zy * (zxx - zyy) - zxy * (zx**2 - zy**2)
                if "profile_curvature" in want:
                    den = g2 if directional else (g2 * np.sqrt(opg**3))
                    v = np.where(flat, one, num_prof / den)
                    put("profile_curvature", v * 100)
                if "tangential_curvature" in want:
                    den = g2 if directional else (g2 * np.sqrt(opg))
                    v = np.where(flat, one, num_tan / den)
                    put("tangential_curvature", v * 100)
                if "planform_curvature" in want:
                    v = np.where(g2 < 10e-15, one, num_tan / np.sqrt(g2**3))
                    put("planform_curvature", v * 100)
                if "flowline_curvature" in want:
                    if directional:
                        v = np.where(flat, one, num_flow / (g2**3) ** 0.5)
                    else:
                        v = np.where(g2 < 10e-15, one, num_flow / ((g2**3) ** 0.5 * opg**0.5))
                    put("flowline_curvature", v * 100)
                if want & {"max_curvature", "min_curvature"}:
                    if directional:
                        half_tr = (zxx + zyy) / 2
                        rad = (((zxx - zyy) / 2) ** 2 + zxy**2) ** 0.5
                        vmax = np.where(flat, one, -(half_tr - rad))
                        vmin = np.where(flat, one, -(half_tr + rad))
                    else:
                        q = (1 + zy**2) * zxx - 2 * zxy * zx * zy + (1 + zx**2) * zyy
                        # NB the reference spells the cross term as 2*zy*zx*zxy in unsphericity: same product order
                        q_u = (1 + zy**2) * zxx - 2 * zy * zx * zxy + (1 + zx**2) * zyy
                        d = 2 * (opg**3) ** 0.5
                        mean = np.where(flat, one, -q / d)
                        unsph = np.where(flat, one, ((q_u / d) ** 2 - (zxx * zyy - zxy**2) / (opg**2)) ** 0.5)
                        vmax = np.where(flat, one, mean + unsph)
                        vmin = np.where(flat, one, mean - unsph)
                    put("max_curvature", vmax * 100)
                    put("min_curvature", vmin * 100)

    if engine != "numba":
        res[:, _window_invalid(dem, 5 if fit_id == 2 else 3)] = np.nan
    return res


def _np_sum_axis0(stack: np.ndarray) -> np.ndarray:
    """np.sum over a contiguous 1-D float64 buffer of n = stack.shape[0] elements, vectorised over pixels.

    Mirrors NumPy's pairwise summation for n <= 128 (8 running accumulators over blocks of 8, combined as
    ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), remainder added sequentially); plain left-to-right for n < 8.
    """
    n = stack.shape[0]
    if n < 8:
        acc = stack[0].copy()
        for i in range(1, n):
            acc = acc + stack[i]
        return acc
    if n > 128:
        half = (n // 2) - ((n // 2) % 8)
        return _np_sum_axis0(stack[:half]) + _np_sum_axis0(stack[half:])
    r = [stack[j].copy() for j in range(8)]
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] = r[j] + stack[i + j]
        i += 8
    acc = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        acc = acc + stack[i]
        i += 1
    return acc


def _rugosity(stack: np.ndarray, resolution: float, T) -> np.ndarray:
    """Jenness rugosity of 3x3 windows (window.py:466-563, the per-window callback the SciPy generic_filter and
    Numba engines run): 16 half segment lengths and 8 Heron triangle areas, every intermediate array held in
    ``out_dtype`` (T) and Python ``sum`` (left to right) for the reductions."""
    T = np.dtype(T).type
    Z = stack
    L = float(resolution)
    dzs = [(Z[4] - Z[i]).astype(T) for i in range(9) if i != 4]
    dls = [T(np.sqrt(j * j + k * k) * L) for j in (-1, 0, 1) for k in (-1, 0, 1) if not (j == 0 and k == 0)]
    for a, b in ((0, 1), (1, 2), (6, 7), (7, 8), (0, 3), (3, 6), (2, 5), (5, 8)):
        dzs.append((Z[a] - Z[b]).astype(T))
        dls.append(T(L))
    hsl = [np.sqrt(dz * dz + dl * dl) / T(2) for dz, dl in zip(dzs, dls)]
    tri = ((3, 0, 12), (0, 1, 8), (1, 2, 9), (2, 4, 14), (4, 7, 15), (7, 6, 11), (6, 5, 10), (5, 3, 13))
    total = None
    for ia, ib, ic in tri:
        a, b, c = hsl[ia], hsl[ib], hsl[ic]
        hs = ((a + b) + c) / T(2)
        area = np.sqrt(hs * (hs - a) * (hs - b) * (hs - c))
        total = area if total is None else total + area
    return total / T(L**2)


def fractal_constants(window_size: int):
    """Regression abscissae of the box-counting fit (window.py:362-393): the divisors q of hw are a uint8 array, so
    ``np.log`` yields float16 and mean / SS_xx are float16 arithmetic.  Returns (qs, x[f16], m_x[f16], SS_xx[f16])."""
    hw = window_size // 2
    qs = np.array([q for q in range(1, hw + 1) if hw % q == 0], dtype=np.uint8)
    x = np.log(qs)
    n = len(x)
    m_x = np.mean(x)
    ss_xx = np.sum(x * x) - n * m_x * m_x
    return qs, x, m_x, ss_xx


def _fractal_roughness(stack: np.ndarray, w: int, T) -> np.ndarray:
    """Taud & Parrot box-counting dimension (window.py:316-401, generic/Numba callback): voxel heights
    V = clip(z - z_c, 0, w) in ``out_dtype``; for every divisor q of w//2 the (w-1)//q squared q x q block maxima
    are summed left to right in ``out_dtype`` and divided by q; slope of log Ns over log q."""
    T = np.dtype(T).type
    c = stack[(w * w) // 2]
    V = [[np.clip(stack[w * j + k] - c, 0, w).astype(T) for k in range(w)] for j in range(w)]
    qs, x, m_x, ss_xx = fractal_constants(w)
    n = len(qs)
    ys = []
    for q in qs:
        q = int(q)
        nq = int((w - 1) / q)
        acc = None
        for j in range(nq):
            for k in range(nq):
                blk = np.stack([V[a][b] for a in range(j * q, (j + 1) * q) for b in range(k * q, (k + 1) * q)], axis=0)
                m = np.max(blk, axis=0)  # NaN-propagating
                acc = m if acc is None else acc + m
        ys.append(np.log(acc / T(q)))
    sy = _np_sum_axis0(np.stack(ys, axis=0))  # np.mean / np.sum of an n-vector: left to right below 8 elements
    sxy = _np_sum_axis0(np.stack([ys[i] * x[i] for i in range(n)], axis=0))
    m_y = sy / T(n)
    ss_xy = sxy - T(n) * m_y * m_x
    return -(ss_xy / ss_xx)


def windowed_indexes(
    dem: np.ndarray, window_size: int, windowed_indexes: list[str], out_dtype=np.float32, tri_method: str = "Riley",
    resolution: float = 1.0,
) -> np.ndarray:
    """Oracle of ``_get_windowed_indexes(..., engine="scipy")`` for TPI / TRI (window.py:67-252, 873-923)."""
    w = int(window_size)
    h = w // 2
    H, W = dem.shape
    pad = np.full((H + 2 * h, W + 2 * h), np.nan, dtype=np.float64)
    pad[h : h + H, h : h + W] = dem
    # generic_filter hands the footprint as a row-major flattened float64 buffer
    stack = np.stack([pad[a : a + H, b : b + W] for a in range(w) for b in range(w)], axis=0)
    c = stack[(w * w) // 2]
    out = np.full((len(windowed_indexes), H, W), np.nan, dtype=out_dtype)
    with np.errstate(all="ignore"):
        for i, name in enumerate(windowed_indexes):
            if name == "topographic_position_index":
                out[i] = c - (_np_sum_axis0(stack) - c) / (w**2 - 1)
            elif name == "terrain_ruggedness_index":
                diff = np.abs(stack - c[None])
                if tri_method.lower() == "riley":
                    out[i] = np.sqrt(_np_sum_axis0(diff**2))
                else:
                    out[i] = _np_sum_axis0(diff) / (w**2 - 1)
            elif name == "roughness":  # window.py:261-289: max - min, NaN if any NaN in the window
                r = np.max(stack, axis=0) - np.min(stack, axis=0)
                r[np.isnan(stack).any(axis=0)] = np.nan
                out[i] = r
            elif name == "rugosity":  # always a 3x3 window (window.py:700-712)
                if w == 3:
                    s9 = stack
                else:
                    p3 = np.full((H + 2, W + 2), np.nan, dtype=np.float64)
                    p3[1 : 1 + H, 1 : 1 + W] = dem
                    s9 = np.stack([p3[a : a + H, b : b + W] for a in range(3) for b in range(3)], axis=0)
                out[i] = _rugosity(s9, resolution, out_dtype)
            elif name == "fractal_roughness":
                out[i] = _fractal_roughness(stack, w, out_dtype)
            else:
                raise ValueError(f"oracle does not cover windowed index '{name}'")
    return out


def _next_fft_len(n: int) -> int:
    """freq.py:32-60: power of two up to 1024, otherwise the next integer whose only prime factors are 2, 3, 5, 7."""
    if n <= 1:
        return 1
    if n <= 1024:
        return int(2 ** np.ceil(np.log2(n)))
    c = n
    while True:
        t = c
        for f in (2, 3, 5, 7):
            while t % f == 0:
                t //= f
        if t == 1:
            return c
        c += 1


def texture_shading(dem: np.ndarray, alpha: float = 0.8) -> np.ndarray:
    """Oracle of ``_texture_shading_fft`` (xdem/terrain/freq.py:63-148): mean-filled, symmetrically padded DEM times
    ``hypot(fx, fy)**alpha`` in the frequency domain (scipy.fft keeps the DEM's precision), cropped, NaN restored."""
    import scipy.fft as sfft

    valid = np.isfinite(dem)
    if not valid.any():
        return np.full_like(dem, np.nan)
    work = dem.copy()
    if not valid.all():
        work[~valid] = np.nanmean(dem)
    H, W = work.shape
    FH, FW = _next_fft_len(H), _next_fft_len(W)
    pr, pc = (FH - H) // 2, (FW - W) // 2
    work = np.pad(work, ((pr, FH - H - pr), (pc, FW - W - pc)), mode="symmetric")
    mag = np.hypot(sfft.rfftfreq(FW)[None, :], sfft.fftfreq(FH)[:, None])
    mag[0, 0] = 1.0
    filt = mag**alpha
    if alpha > 0:
        filt[0, 0] = 0.0
    spec = sfft.rfft2(work, s=(FH, FW))
    spec *= filt
    out = sfft.irfft2(spec, s=(FH, FW))[pr : pr + H, pc : pc + W]
    out[~valid] = np.nan
    return out


def terrain_attributes(
    dem: np.ndarray,
    attribute: list[str],
    resolution: float = 1.0,
    degrees: bool = True,
    hillshade_altitude: float = 45.0,
    hillshade_azimuth: float = 315.0,
    hillshade_z_factor: float = 1.0,
    surface_fit: str = "Florinsky",
    curv_method: str = "geometric",
    tri_method: str = "Riley",
    window_size: int = 3,
    out_dtype=None,
    window_size_fractal: int = 13,
    texture_alpha: float = 0.8,
    engine: str = "scipy",
) -> list[np.ndarray]:
    """Oracle of ``_get_terrain_attribute`` for ndarray input (terrain.py:528-666): engines + unit/clip post-steps.
    ``engine`` selects the surface-fit recipe only; windowed indexes are the SciPy engine's float64 callbacks either way
    (the Numba engine hands them the window in the DEM dtype, window.py:851 -- its float32 sums are a noisier evaluation
    of the same quantity, see tests/test_oracle_golden.py::test_T11_windowed)."""
    dem = np.asarray(dem)
    if out_dtype is None:
        out_dtype = np.float32 if np.issubdtype(dem.dtype, np.integer) else dem.dtype
    if np.issubdtype(dem.dtype, np.integer):
        dem = dem.astype(np.float32)
    surf = [a for a in attribute if a in SURFACE_ATTRIBUTES]
    win = [a for a in attribute if a in WINDOW_ATTRIBUTES]
    results: dict[str, np.ndarray] = {}
    if surf:
        s = surface_attributes(dem, resolution, surf, out_dtype, surface_fit, curv_method,
                               hillshade_altitude, hillshade_azimuth, hillshade_z_factor, engine)
        for i, name in enumerate(surf):
            v = s[i]
            if degrees and name in ("slope", "aspect"):
                v = np.rad2deg(v)
            if name == "hillshade":
                v = np.clip(v, 0, 255)
            results[name] = v
    if win:
        wi = windowed_indexes(dem, window_size, win, out_dtype, tri_method, resolution)
        for i, name in enumerate(win):
            results[name] = wi[i]
    frac = [a for a in attribute if a in FRACTAL_ATTRIBUTES]
    if frac:  # second windowed call with its own size (terrain.py:619-630)
        wi = windowed_indexes(dem, window_size_fractal, frac, out_dtype, tri_method, resolution)
        for i, name in enumerate(frac):
            results[name] = wi[i]
    if "texture_shading" in attribute:
        with np.errstate(all="ignore"):
            results["texture_shading"] = texture_shading(dem, texture_alpha).astype(out_dtype)
    return [results[a] for a in attribute]
