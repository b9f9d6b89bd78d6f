"""CPU-side tests: the C-ABI library loads and exports every symbol include/xdemhip.h declares; the host-side mirrors
validate arguments like the reference; no compute is attempted without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from xdem_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(lib):
    import glob

    total = 0
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        hdr = open(h).read()
        names = sorted(set(re.findall(r"\b(xdemhip_[a-z_0-9]+)\s*\(", hdr)))
        total += len(names)
        for n in names:
            assert hasattr(lib, n), f"{n} declared in include/{os.path.basename(h)} but not exported by libxdemhip.so"
    assert total >= 19, total
    assert lib.xdemhip_version() >= 100


def test_option_names_of_the_header_and_the_library_agree():
    """include/xdemhip.h documents the OPTIONS of a context (at most fourteen names since round 6), include/xdemhip_test.h the test
    switches between internal routes; the tables of csrc/capi.hip hold exactly those names, and the Python binding routes a name to
    the right entry point.  (That every option changes what it says it changes is the GPU suite's business.)"""
    from xdem_amd import _lib

    def doc_names(path):
        txt = open(os.path.join(ROOT, "include", path)).read()
        return set(re.findall(r'^ \*   "([a-z_0-9]+)"', txt, flags=re.M))   # (the list entries: three spaces of indent; continuation lines have more)

    src = open(os.path.join(ROOT, "xdem_amd", "csrc", "capi.hip")).read()
    tab = lambda name: set(re.findall(r'\{"([a-z_0-9]+)",', src[src.index(name):src.index("};", src.index(name))]))
    opts = tab("kOptions[]") | {"host_release", "host_copy_threads"}
    switches = tab("kTestSwitches[]")
    assert doc_names("xdemhip.h") == opts and len(opts) <= 14, (sorted(doc_names("xdemhip.h")), sorted(opts))
    assert doc_names("xdemhip_test.h") | {"terrain_store", "terrain_rows", "terrain_sync", "vario_deff"} == switches, sorted(switches)
    assert switches == set(_lib.Context.TEST_SWITCHES) and not (opts & switches)


def test_header_constants_match_the_binding():
    """The numeric constants the ctypes side hard-codes are the header's: memory spaces, dtypes, allocation flags, reduction kinds."""
    import re

    from xdem_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "xdemhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", " ", hdr)
    enums = {}
    for body in re.findall(r"enum\s*\{([^}]*)\}", hdr):
        nxt = 0
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            m = re.match(r"(\w+)\s*(?:=\s*(-?\w+))?", item)
            name, val = m.group(1), m.group(2)
            if val is not None:
                try:
                    nxt = int(val.rstrip("uUlL"), 0)
                except ValueError:   # an expression (attribute bits: 1u << k): not one of the constants checked here
                    nxt = None
            enums[name] = nxt
            nxt = None if nxt is None else nxt + 1
    assert (enums["XDEMHIP_HOST"], enums["XDEMHIP_DEVICE"]) == (_lib.HOST, _lib.DEVICE)
    assert (enums["XDEMHIP_F32"], enums["XDEMHIP_F64"]) == (_lib.F32, _lib.F64)
    assert (enums["XDEMHIP_ALLOC_CONTIGUOUS"], enums["XDEMHIP_ALLOC_RECYCLED"], enums["XDEMHIP_ALLOC_CHUNKED"],
            enums["XDEMHIP_ALLOC_SCATTERED"]) == (1, 2, 4, 8)
    src = open(os.path.join(ROOT, "xdem_amd", "_lib.py")).read()
    assert "flags = 8 if scattered else (4 if chunked else ((1 if contiguous else 0) | (2 if recycled else 0)))" in src


def test_fractal_constants_reproduce_numpy_float16(lib):
    """Host-only helper of the C-ABI against the NumPy arithmetic the reference runs (window.py:362-393): np.log of a
    uint8 divisor array is float16, and so are its mean and SS_xx."""
    import terrain_oracle as to

    for w in list(range(3, 131, 2)) + [241, 361, 481, 505, 511]:
        q = (ctypes.c_int * 24)()
        x = (ctypes.c_double * 24)()
        mx, ss = ctypes.c_double(), ctypes.c_double()
        with np.errstate(all="ignore"):
            qs, xo, mxo, sso = to.fractal_constants(w)
        n = lib.xdemhip_fractal_constants(w, 24, q, x, ctypes.byref(mx), ctypes.byref(ss))
        assert n == len(qs), w
        assert list(q[:n]) == [int(v) for v in qs]
        assert np.array_equal(np.array(x[:n]), xo.astype(np.float64)), w
        assert np.array_equal(np.float64(mx.value), np.float64(mxo), equal_nan=True), w
        assert np.array_equal(np.float64(ss.value), np.float64(sso), equal_nan=True), w
    assert lib.xdemhip_fractal_constants(4, 24, q, x, ctypes.byref(mx), ctypes.byref(ss)) < 0


def test_no_gpu_means_loud_failure(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    assert lib.xdemhip_create(0, ctypes.byref(h)) < 0 and not h
    from xdem_amd import _lib, terrain

    with pytest.raises(_lib.XdemHipError, match="no CPU fallback"):
        terrain.get_terrain_attribute(np.zeros((8, 8), np.float32), "slope", resolution=1.0)


def test_terrain_argument_validation_messages():
    """Same checks and messages as xdem/terrain/terrain.py:293-409 (asserted by the reference's tests
    tests/test_terrain/test_terrain.py:428-490, test_surfit.py:123-134, 169-176) -- all raised before any GPU work."""
    from xdem_amd import _lib
    from xdem_amd import terrain as t

    dem = np.ones((6, 6), np.float32)
    with pytest.raises(ValueError, match="'Horn' surface fit method cannot be used for to calculate curvatures"):
        t.get_terrain_attribute(dem, "profile_curvature", resolution=1.0, surface_fit="Horn")
    with pytest.raises(ValueError, match=re.escape("'resolution' must be provided as an argument for attributes: ['slope']")):
        t.get_terrain_attribute(dem, "slope")
    with pytest.raises(ValueError, match=re.escape(
            "Surface fit and rugosity require the same X and Y resolution ((1.0, 2.0) was given). "
            "This was required by: ['max_curvature'].")):
        t.get_terrain_attribute(dem, "max_curvature", resolution=(1.0, 2.0))
    with pytest.raises(ValueError, match="Attribute 'nope' is not supported. Choices:"):
        t.get_terrain_attribute(dem, "nope", resolution=1.0)
    with pytest.raises(ValueError, match="Surface fit 'x' is not supported"):
        t.get_terrain_attribute(dem, "slope", resolution=1.0, surface_fit="x")
    with pytest.raises(ValueError, match="Curvature method 'x' is not supported"):
        t.get_terrain_attribute(dem, "slope", resolution=1.0, curv_method="x")
    with pytest.raises(ValueError, match="TRI method 'x' is not supported"):
        t.get_terrain_attribute(dem, "terrain_ruggedness_index", tri_method="x")
    with pytest.raises(ValueError, match="Azimuth must be a value between 0 and 360"):
        t.hillshade(dem, resolution=1.0, azimuth=361)
    with pytest.raises(ValueError, match="Altitude must be a value between 0 and 90"):
        t.hillshade(dem, resolution=1.0, altitude=91)
    with pytest.raises(ValueError, match="z_factor must be a non-negative finite value"):
        t.hillshade(dem, resolution=1.0, z_factor=np.inf)
    with pytest.raises(ValueError, match="engine must be 'hip', 'scipy' or 'numba'"):
        t.slope(dem, resolution=1.0, engine="cpu")
    for bad in (-0.1, 2.1):  # tests/test_terrain/test_freq.py:47-51 (checked before any GPU work would matter)
        with pytest.raises(ValueError, match="Alpha must be between 0 and 2"):
            t.texture_shading(dem, alpha=bad)
    with pytest.raises(ValueError, match=re.escape("'resolution' must be provided as an argument for attributes: ['rugosity']")):
        t.rugosity(dem)
    with pytest.warns(UserWarning, match="window sizes larger or equal to 5"):
        try:  # (without a GPU the launch itself fails loudly, after the warning)
            t.fractal_roughness(dem, window_size_fractal=3)
        except _lib.XdemHipError:
            pass
    with pytest.warns(UserWarning, match="less than 13 can be inaccurate"):
        try:
            t.fractal_roughness(dem, window_size_fractal=9)
        except _lib.XdemHipError:
            pass
    with pytest.warns(DeprecationWarning, match="'slope_method' is deprecated"):
        with pytest.raises(ValueError):
            t.get_terrain_attribute(dem, "slope", slope_method="bad", resolution=1.0)


def test_halo_depth_and_row_blocks():
    from xdem_amd import dist as d

    assert d.halo_depth(["slope"], "Florinsky") == 2 and d.halo_depth(["slope"], "Horn") == 1
    assert d.halo_depth(["topographic_position_index"], window_size=7) == 3
    assert d.halo_depth(["slope", "terrain_ruggedness_index"], "ZevenbergThorne", 3) == 1
    for total, world in ((10, 3), (65536, 8), (7, 7), (40000, 8)):
        blocks = [d.row_block(total, world, r) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == total
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_default_context_does_not_deadlock():
    """default_context() creates the process-wide context under its own lock; Context() -> lib() takes the loader's lock: they
    must be two locks (one non-reentrant lock for both dead-locked every GPU run of round 3's first build).  Without a GPU the
    call has to come back -- with the library's "no usable device" error -- instead of hanging."""
    import threading

    import torch

    from xdem_amd import _lib

    box = []

    def run():
        try:
            box.append(_lib.default_context(0))
        except _lib.XdemHipError as e:
            box.append(e)

    _lib._default_ctx.pop(0, None) if not torch.cuda.is_available() else None
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(60)
    assert not t.is_alive(), "default_context() hung"
    assert box and (isinstance(box[0], _lib.Context) or "no usable" in str(box[0]))


def test_streaming_kernels_keep_plane_pointers_in_scalar_registers(tmp_path):
    """The streaming terrain kernels store through `global_store_dword v, v, s[base]` inline asm WITHOUT the scalar copy of the
    plane pointer the tile kernels carry (DirectSink<float, false>): that is only safe while no plane pointer is ever restored
    from a VGPR lane (v_readlane writes an SGPR on the vector unit; a VMEM instruction may not read it for 5 wait states and
    inline asm gets no hazard handling).  Checked on the compiled code: the scalar registers a streaming kernel restores with
    v_readlane (round 5 / 6: the saved exec mask of the cold path) are never the base pair of a plane store -- or every plane store
    takes its pointer through the `s_mov_b64` copy inside the asm text (a SALU read of a VALU-written SGPR and a VMEM read of a
    SALU-written SGPR are both interlocked by the hardware).  Also: no streaming kernel of the default (lean) tail uses scratch
    memory, and the Florinsky sets with curvatures reach four waves per SIMD (<= 128 VGPRs: round 6)."""
    import re
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "terrain_ff.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           os.path.join(root, "xdem_amd", "csrc", "terrain_ff.hip"), "-o", out], stderr=subprocess.DEVNULL)
    src = open(out).read()
    n = n_lane = n_four = 0
    for m in re.finditer(r"\n(_Z\w*terrain_strip_kernel\w+):[^\n]*\n(.*?)\n\t\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel", src, re.S):
        body, desc = m.group(2), m.group(3)
        assert "global_load_lds_dwordx4" in body and "global_store_dword" in body
        restored = set(re.findall(r"v_readlane_b32 (s\d+)", body))
        if restored:
            lines = [ln.strip() for ln in body.split("\n") if ln.startswith("\t") and not ln.strip().startswith((";", "."))]
            stores = [i for i, ln in enumerate(lines) if ln.startswith("global_store_dword")]
            copied = all(lines[i - 1].startswith("s_mov_b64") and lines[i - 1].split()[1].rstrip(",") == lines[i].split()[3] for i in stores)
            bases = set()
            for lo, hi in re.findall(r"global_store_dword v\d+, v\d+, s\[(\d+):(\d+)\]", body):
                bases.update(("s" + lo, "s" + hi))
            assert stores and (copied or not (restored & bases)), (m.group(1), sorted(restored & bases))
            n_lane += 1
        vgprs = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
        lean = re.search(r"SpecILj\d+ELi\d+ELi\d+ELi\d+ELi\d+ELi2EE", m.group(1)) is not None
        if lean:   # (the mixed tail of option terrain_math = 0 keeps three waves per SIMD)
            assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", desc), m.group(1)
            assert vgprs <= 128, (m.group(1), vgprs)
            n_four += 1
        n += 1
    assert n >= 6 and n_four >= 6   # 3 attribute sets x 2 tails (x band heights)


def test_largest_pair_distance_from_hulls():
    """`bin_func='even'` needs np.nanmax of the sampled pair distances (skgstat.binning.even_width_lags clips maxlag to it); the
    product takes it from the convex hulls of the point sets.  Against brute force: every-a-with-every-b and i < j blocks, lattice
    and real coordinates, several blocks, a collinear set, tiny sets."""
    from xdem_amd import spatialstats as ss

    rng = np.random.default_rng(5)

    def brute(blocks):
        best = 0.0
        for blk in blocks:
            xa, ya = np.asarray(blk[0], float), np.asarray(blk[1], float)
            xb, yb = (xa, ya) if len(blk) == 3 else (np.asarray(blk[3], float), np.asarray(blk[4], float))
            d = np.sqrt((xa[:, None] - xb[None, :]) ** 2 + (ya[:, None] - yb[None, :]) ** 2)
            best = max(best, float(d.max())) if d.size else best
        return best

    for trial in range(6):
        n, m = int(rng.integers(70, 900)), int(rng.integers(70, 900))
        lat = trial % 2 == 0
        mk = (lambda k: rng.integers(0, 500, k).astype(float)) if lat else (lambda k: rng.uniform(-3e5, 7e5, k))
        blocks = [(mk(n), mk(n), np.zeros(n)), (mk(n), mk(n), np.zeros(n), mk(m) + 100.0, mk(m), np.zeros(m))]
        assert ss._max_pair_distance(blocks) == brute(blocks)
        assert ss._max_pair_distance(blocks[1:]) == brute(blocks[1:])
    t = np.arange(200.0)
    line = [(3.0 * t, 7.0 - 2.0 * t, np.zeros(200))]
    assert ss._max_pair_distance(line) == brute(line)
    assert ss._max_pair_distance([(np.array([1.0]), np.array([2.0]), np.zeros(1))]) == 0.0
    assert ss._max_pair_distance([(np.array([0.0, 3.0]), np.array([0.0, 4.0]), np.zeros(2))]) == 5.0
    assert ss._max_pair_distance([]) == 0.0


def test_variogram_host_preparation():
    from xdem_amd import spatialstats as ss

    runs, samples, ratio = ss._choose_cdist_equidistant_sampling_parameters(extent=(0, 999, 0, 999), shape=(1000, 1000), subsample=1000)
    assert (runs, samples) == (100, 23) and abs(ratio - 0.0037559253144038175) < 1e-18   # SURVEY probe value
    with pytest.raises(ValueError, match="needs to be at least 10"):
        ss._choose_cdist_equidistant_sampling_parameters(extent=(0, 9, 0, 9), shape=(10, 10), subsample=5)
    # the whole T7 table recorded from the reference (integers exact, the ratio bit for bit), on the PRODUCT's function
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vario_golden.npz"))
    n_ok = 0
    for subsample, nx, ny, gsd, runs, samples, ratio in z["T7|params"]:
        shape = (int(nx), int(ny))
        kw = dict(extent=(0.0, (shape[0] - 1) * gsd, 0.0, (shape[1] - 1) * gsd), shape=shape, subsample=int(subsample))
        if runs < 0:
            with pytest.raises(ValueError, match="needs to be at least"):
                ss._choose_cdist_equidistant_sampling_parameters(**kw)
            continue
        got = ss._choose_cdist_equidistant_sampling_parameters(**kw)
        assert got[0] == int(runs) and got[1] == int(samples) and got[2] == ratio, (kw, got, runs, samples, ratio)
        n_ok += 1
    assert n_ok >= 30
    k = np.array([0x3F800000 << 1, 0x40000000 << 1], dtype=np.uint64)
    assert ss._key_to_value(k, 32).tolist() == [1.0, 2.0]


def test_raster_equidistant_sampler_matches_enumeration():
    """The raster form of the equidistant ring sampler (closed-form column spans per row, no per-pixel distances) against the
    oracle's enumeration of the same rings: identical pixel sets when a ring is taken in full, distinct valid ring members of
    the requested number otherwise, uniform over the ring; block structure of the centre-disk x rings scheme."""
    import variogram_oracle as vo
    from xdem_amd import spatialstats as ss

    rng = np.random.default_rng(1)
    ny, nx, gsd = 300, 420, 2.0
    v = rng.normal(size=(ny, nx)).astype(np.float32)
    v[50:80, 100:200] = np.nan
    valid = np.isfinite(v)
    iy, ix = np.mgrid[0:ny, 0:nx]
    for cx, cy, lo, hi, n in [(10, 10, 0.0, 30.0, 50), (200, 150, 100.0, 160.0, 20000), (419, 299, 37.5, 400.0, 300),
                              (5, 290, 250.0, 1000.0, 10**6), (100, 100, 0.0, 3.0, 100), (0, 0, 2000.0, 3000.0, 10)]:
        got = ss._draw_ring_pixels(valid, ny, nx, cx, cy, lo, hi, gsd, n, np.random.default_rng(3))
        d = np.sqrt(((ix - cx) * gsd) ** 2 + ((iy - cy) * gsd) ** 2)
        full = np.flatnonzero((valid & (d >= lo) & (d < hi)).ravel())
        assert np.unique(got).size == got.size and np.isin(got, full).all() and got.size == min(n, full.size)
        if n >= full.size:
            assert np.array_equal(np.sort(got), full)
    # uniformity: 400 draws of 200 from a ring of ~12000 pixels hit every octant of the ring about equally
    cx, cy, lo, hi = 210, 150, 100.0, 160.0
    hits = np.zeros(8)
    for seed in range(400):
        g = ss._draw_ring_pixels(None, ny, nx, cx, cy, lo, hi, gsd, 200, np.random.default_rng(seed))
        ang = np.arctan2((g // nx) - cy, (g % nx) - cx)
        hits += np.bincount(((ang + np.pi) / (2 * np.pi) * 8).astype(int) % 8, minlength=8)
    assert hits.sum() == 400 * 200 and np.all(np.abs(hits / hits.mean() - 1) < 0.05)
    # blocks: same ring bounds as the oracle's restatement (taken in full: samples >= every ring)
    coords, extent, maxlag = vo.grid_coords_extent_maxlag((nx, ny), gsd)  # (upstream's meshgrid convention: shape[0] along x)
    centres = []
    blocks = ss.equidistant_blocks_from_raster(v, gsd, 3, 30, 0.002, np.random.default_rng(9), centres_out=centres)
    assert len(blocks) == 3
    r0, radii = ss._equidistant_radii(30, 0.002, gsd, maxlag)
    flat, fvalid = v.reshape(-1), valid.reshape(-1)
    for (ax, ay, av, bx, by, bv), (cxi, cyi) in zip(blocks, centres):
        assert valid[cyi, cxi] and ax.size == 30 and np.isfinite(av).all() and np.isfinite(bv).all()
        assert np.array_equal(av, v[(ay / gsd).astype(int), (ax / gsd).astype(int)])
        assert np.all(np.hypot(ax - cxi * gsd, ay - cyi * gsd) < r0)
        d = np.hypot(bx - cxi * gsd, by - cyi * gsd)
        k = np.digitize(d, radii) - 1
        assert np.all(np.diff(k) >= 0)  # inner to outer
        # membership and sizes against the oracle's enumeration of the same rings around the same centre
        dist = np.hypot(coords[:, 0] - cxi * gsd, coords[:, 1] - cyi * gsd)
        for i, (lo, hi) in enumerate(zip(radii[:-1], radii[1:])):
            members = np.flatnonzero(fvalid & (dist >= lo) & (dist < hi))
            mine = ((by[k == i] / gsd).astype(np.int64) * nx + (bx[k == i] / gsd).astype(np.int64))
            assert mine.size == min(30, members.size) and np.isin(mine, members).all()


def test_nuthkaab_class_contract():
    from xdem_amd import coreg

    nk = coreg.NuthKaab(max_iterations=7, offset_threshold=0.01, subsample=1)
    assert nk.meta["inputs"]["iterative"] == {"max_iterations": 7, "tolerance": 0.01}
    assert nk.meta["inputs"]["fitorbin"]["bin_sizes"] == 72
    # the options round 1 refused: un-binned fit, explicit bin edges (both upstream forms), initial shift (upstream's checks)
    nf = coreg.NuthKaab(bin_before_fit=False, bin_sizes={"aspect": [0.0, 2.0, 4.0, 6.3]}, initial_shift=(3.0, -1.5))
    assert nf.meta["inputs"]["fitorbin"]["fit_or_bin"] == "fit" and nf.meta["inputs"]["affine"]["initial_shift"] == (3.0, -1.5, 0)
    assert np.array_equal(nf.meta["inputs"]["fitorbin"]["bin_sizes"], [0.0, 2.0, 4.0, 6.3])
    with pytest.raises(ValueError, match="exactly two or three numerical values"):
        coreg.NuthKaab(initial_shift=[1.0, 2.0])
    with pytest.warns(UserWarning, match="work in progress"):
        assert coreg.NuthKaab(initial_shift=(1.0, 2.0, 5.0)).meta["inputs"]["affine"]["initial_shift"] == (1.0, 2.0, 0)
    with pytest.raises(ValueError, match="increasing bin edges"):
        coreg.NuthKaab(bin_sizes=[3.0, 2.0, 1.0])
    with pytest.raises(ValueError, match="'transform' must be given if both DEMs are array-like."):
        nk.fit(np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32))
    x = np.linspace(0, 6, 50)
    assert np.allclose(coreg._nuth_kaab_fit_func(x, 2.0, 0.5, 1.0), 2.0 * np.cos(0.5 - x) + 1.0)


def test_subsample_ranks_are_the_host_draw():
    """``rng.choice(valids, n, replace=False)`` == ``valids[rng.choice(len(valids), n, replace=False)]`` -- what lets nuth_kaab draw
    RANKS from the count of valid pixels and leave the mask on the device (coreg.subsample_ranks, xdemhip_nk_subsample): both of
    NumPy's algorithms for a draw without replacement (tail shuffle, Floyd's set for n < size / 50 above 10 000 elements)."""
    from xdem_amd import coreg

    rng0 = np.random.default_rng(0)
    for size, frac_valid in ((977, 0.7), (200_000, 0.8)):
        valid = rng0.random(size) < frac_valid
        valids = np.flatnonzero(valid)
        for sub, seed in ((0.5, 1), (25, 2), (3000, 3), (1.0 - 1e-9, 4), (10**9, 5), (0.001, None)):
            if seed is None:
                assert coreg.subsample_ranks(len(valids), sub, None).size == int(sub * len(valids))
                continue
            ranks = coreg.subsample_ranks(len(valids), sub, seed)
            n = min(int(sub * len(valids)) if sub <= 1 else int(sub), len(valids))
            assert ranks.size == n and np.unique(ranks).size == n
            want = np.random.default_rng(seed).choice(valids, n, replace=False)
            assert np.array_equal(valids[ranks], want)
            m = coreg.subsample_valid_mask(valid.reshape(1, -1), sub, seed)
            assert m.sum() == n and np.array_equal(np.flatnonzero(m.ravel()), np.sort(want))
    with pytest.raises(ValueError, match="no valid points"):
        coreg.subsample_ranks(0, 0.5, 1)
    with pytest.raises(ValueError, match="must be > 0"):
        coreg.subsample_ranks(10, 0, 1)


def test_unbinned_fit_from_sums_equals_curve_fit():
    """bin_before_fit=False: the normal-equation solution from the ten sums xdemhip_nk_step_fit returns is the optimum that
    the reference's curve_fit call (xdem/coreg/base.py:975-989) converges to from its p0 (affine.py:384)."""
    import scipy.optimize

    from xdem_amd import coreg

    rng = np.random.default_rng(3)
    x = rng.uniform(0, 2 * np.pi, 5000)
    y = 1.7 * np.cos(0.9 - x) + 0.4 + rng.normal(0, 0.3, x.size)
    c, sn = np.cos(x), np.sin(x)
    sums = np.array([x.size, c.sum(), sn.sum(), (c * c).sum(), (sn * sn).sum(), (c * sn).sum(), y.sum(), (y * c).sum(), (y * sn).sum(),
                     (y * y).sum()])
    east, north, vert = coreg._fit_from_sums({"sums": sums})
    p0 = (3 * np.nanstd(y) / (2**0.5), 0.0, np.nanmean(y))
    (a, b, cc), _ = scipy.optimize.curve_fit(coreg._nuth_kaab_fit_func, x, y, p0=p0, absolute_sigma=True)
    assert np.allclose([east, north, vert], [a * np.sin(b), a * np.cos(b), cc], rtol=1e-6, atol=1e-9)


def test_variogram_models_fit_and_correlation():
    """Host-side callers of the variogram path (xdem/spatialstats.py:1583-1804): model table checks, sum-of-models fit with
    the reference's bounds / first guesses, covariance and correlation functions.  (Model formulas restate scikit-gstat's
    published definitions: parity unpinned, so these are self-consistency checks.)"""
    import pandas as pd

    from xdem_amd import spatialstats as ss
    from xdem_amd import variogram_models as vm

    h = np.linspace(0, 100, 51)
    for name in vm.SUPPORTED:
        extra = [1.5] if vm.n_params(name) == 3 else []
        g = getattr(vm, name)(h, 30.0, 2.0, *extra)
        assert g[0] == 0 and np.all(np.diff(g) >= -1e-12) and abs(g[-1] - 2.0) < 0.02 * 2.0   # monotone, reaches the sill
        at_range = getattr(vm, name)(np.array([30.0]), 30.0, 2.0, *extra)[0]
        assert at_range >= 0.94 * 2.0                                                             # effective range: ~95 % of the sill
        assert vm.model_name(name[:3].upper()) == name and vm.model_name(getattr(vm, name)) == name
    with pytest.raises(ValueError, match="not recognized"):
        vm.model_name("nope")
    true = vm.spherical(h, 30, 0.6) + vm.gaussian(h, 90, 0.4)
    df = pd.DataFrame({"exp": true[1:], "lags": h[1:], "count": 1000, "err_exp": np.nan})
    fun, par = ss.fit_sum_model_variogram(["Sph", "Gau"], df)
    assert list(par["model"]) == ["spherical", "gaussian"]
    assert np.allclose(par["range"].values, [30, 90], rtol=1e-5) and np.allclose(par["psill"].values, [0.6, 0.4], rtol=1e-5)
    assert np.allclose(fun(h), true, atol=1e-7)
    rho = ss.correlation_from_variogram(par)
    cov = ss.covariance_from_variogram(par)
    assert rho(np.array([0.0]))[0] == 1.0 and abs(rho(np.array([1e4]))[0]) < 1e-12 and np.isclose(cov(np.array([0.0]))[0], 1.0)
    # weighted fit path + a 3-parameter model
    df2 = pd.DataFrame({"exp": vm.stable(h[1:], 40, 1.0, 1.2), "lags": h[1:], "count": 10, "err_exp": 0.01 + 0.0 * h[1:]})
    _, par2 = ss.fit_sum_model_variogram(["stable"], df2, bounds=[(1, 100), (0.1, 2), (0.5, 2)], p0=[30, 0.8, 1.0])
    assert np.allclose([par2["range"].values[0], par2["psill"].values[0], par2["smooth"].values[0]], [40, 1.0, 1.2], rtol=1e-4)
    with pytest.raises(ValueError, match='must contain the columns "model", "range" and "psill"'):
        ss.get_variogram_model_func(pd.DataFrame({"model": ["spherical"], "range": [1.0]}))
    with pytest.raises(ValueError, match="ranges must have non-zero, positive values"):
        ss.get_variogram_model_func(pd.DataFrame({"model": ["spherical"], "range": [0.0], "psill": [1.0]}))
    with pytest.raises(ValueError, match='must contain the column "smooth"'):
        ss.get_variogram_model_func(pd.DataFrame({"model": ["matern"], "range": [1.0], "psill": [1.0]}))


def test_dem_coregister_3d_argument_plumbing(monkeypatch):
    """Host logic of DEM.coregister_3d (xdem/dem.py:621-665) without touching the GPU: random_state reaches fit(), `resample`
    reaches apply(), bias_vars / foreign methods / mismatched grids raise."""
    import xdem_amd
    from xdem_amd import coreg

    seen = {}

    def fake_fit(self, ref, tba, inlier_mask=None, resolution=None, **kw):
        seen["fit"] = dict(kw, resolution=resolution, mask=None if inlier_mask is None else inlier_mask.dtype)
        return self

    def fake_fit_with_shift(self, ref, tba, inlier_mask=None, resolution=None, **kw):
        fake_fit(self, ref, tba, inlier_mask, resolution, **kw)
        self.meta["outputs"]["affine"] = {"shift_x": 3.0, "shift_y": -5.0, "shift_z": 1.0}
        return self

    def fake_translation(elev, sx, sy, sz, res, resample=True):
        seen["apply"] = resample
        seen["apply_args"] = (sx, sy, sz, tuple(res))
        return elev + sz

    monkeypatch.setattr(coreg.NuthKaab, "fit", fake_fit_with_shift)
    monkeypatch.setattr(coreg, "apply_translation", fake_translation)
    a = xdem_amd.DEM(np.zeros((5, 6), dtype=np.float32), transform=(2.0, 0.0, 0.0, 0.0, -2.0, 10.0))
    b = xdem_amd.DEM(np.ones((5, 6), dtype=np.float32), transform=(2.0, 0.0, 0.0, 0.0, -2.0, 10.0))
    out = a.coregister_3d(b, coreg.NuthKaab(), inlier_mask=np.ones((5, 6), dtype=np.uint8), random_state=7, resample=False)
    assert isinstance(out, xdem_amd.DEM) and np.all(out.data == 1.0)
    assert seen["fit"] == {"random_state": 7, "resolution": (2.0, 2.0), "mask": np.dtype(bool)} and seen["apply"] is False
    assert seen["apply_args"] == (3.0, -5.0, 1.0, (2.0, 2.0))
    # resample=False: the horizontal shift moves the geotransform (xdem/coreg/base.py:1567-1570), the grid spacing stays
    assert out.transform == (2.0, 0.0, 3.0, 0.0, -2.0, 5.0)
    out = a.coregister_3d(b)  # default method, default resample: data resampled onto the unchanged grid
    assert seen["apply"] is True and "random_state" not in seen["fit"] and out.transform == a.transform
    with pytest.raises(NotImplementedError, match="bias_vars"):
        a.coregister_3d(b, bias_vars={"x": np.zeros((5, 6))})
    with pytest.raises(ValueError, match="must be an xdem_amd.coreg instance"):
        a.coregister_3d(b, coreg_method=object())
    with pytest.raises(NotImplementedError, match="share one grid"):
        a.coregister_3d(xdem_amd.DEM(np.ones((5, 7), dtype=np.float32)))


def test_number_effective_samples_host_logic(monkeypatch):
    """number_effective_samples / spatial_error_propagation (xdem/spatialstats.py:2311-2458) without the GPU: the numeric-area
    branch is host SciPy (checked against the closed form of a single model), the mask branch must hand the upstream pixel
    coordinates and unit errors to neff_hugonnet_approx, bad inputs raise upstream's message."""
    import pandas as pd

    from xdem_amd import spatialstats as ss

    one = pd.DataFrame({"model": ["spherical"], "range": [300.0], "psill": [1.0]})
    for area in (1e3, 1e5, 2e6):
        num, theo = ss.number_effective_samples(area, one), ss.neff_circular_approx_theoretical(area, one)
        assert np.isclose(num, theo, rtol=1e-6), (area, num, theo)
    two = pd.DataFrame({"model": ["spherical", "gaussian"], "range": [50.0, 500.0], "psill": [0.7, 0.3]})
    assert ss.number_effective_samples(25000, two) == ss.neff_circular_approx_numerical(25000, two)  # ints are numeric areas
    seen = {}

    def fake(coords, errors, params_variogram_model, **kw):
        seen.update(coords=coords, errors=errors, kw=kw)
        return 7.0

    monkeypatch.setattr(ss, "neff_hugonnet_approx", fake)
    mask = np.zeros((4, 6), dtype=bool)
    mask[1, 2] = mask[3, 5] = mask[0, 0] = True
    assert ss.number_effective_samples(mask, two, rasterize_resolution=2.5, subsample=10, random_state=3) == 7.0
    # upstream: x = res * arange(shape[0]), y = res * arange(shape[1]), coords = meshgrid(y, x)[:, mask].T
    assert np.array_equal(seen["coords"], np.array([[0.0, 0.0], [5.0, 2.5], [12.5, 7.5]]))
    assert np.array_equal(seen["errors"], np.ones(3)) and seen["kw"] == {"subsample": 10, "random_state": 3}
    with pytest.warns(UserWarning, match="20% of the shortest"):
        ss.number_effective_samples(mask, two)
    assert np.array_equal(seen["coords"][1], [20.0, 10.0])  # default resolution = min(range) / 5 = 10
    with pytest.raises(ValueError, match="Area must be a float, integer, Vector subclass or geopandas dataframe."):
        ss.number_effective_samples("area", two)
    with pytest.raises(ValueError, match="rasterize resolution must be"):
        ss.number_effective_samples(mask, two, rasterize_resolution="1")
    err = np.full((4, 6), 2.0)
    err[0, 0] = np.nan
    se = ss.spatial_error_propagation([25000.0, mask], err, two, rasterize_resolution=2.5)
    assert np.isclose(se[0], 2.0 / np.sqrt(ss.neff_circular_approx_numerical(25000.0, two)))
    assert np.isclose(se[1], 2.0 / np.sqrt(7.0))  # nanmean over the mask's pixels
    with pytest.raises(ValueError, match="needs the grid spacing"):
        ss.spatial_error_propagation([mask], err, two)


def test_nuthkaab_fit_apply_interface(monkeypatch):
    """Coreg.fit / apply / fit_and_apply call surface (xdem/coreg/base.py:2250-2590) without the GPU: grid spacing from an
    affine transform, (array, transform) returned when a transform comes in, shift moved into the transform for
    resample=False, copies are independent, foreign arguments raise."""
    from xdem_amd import coreg

    calls = {}

    def fake_nk(ref, tba, inlier, res, **kw):
        calls["fit"] = dict(res=res, **{k: kw[k] for k in ("subsample", "random_state", "bin_statistic")})
        return (3.0, -4.0, 1.5), 42

    def fake_apply(elev, sx, sy, sz, res, resample=True, ctx=None):
        calls["apply"] = (sx, sy, sz, res, resample)
        return elev + sz

    monkeypatch.setattr(coreg, "nuth_kaab", fake_nk)
    monkeypatch.setattr(coreg, "apply_translation", fake_apply)
    a, b = np.zeros((4, 5), np.float32), np.ones((4, 5), np.float32)
    tr = (2.0, 0.0, 100.0, 0.0, -2.0, 500.0)
    nk = coreg.NuthKaab().fit(a, b, transform=tr, crs="EPSG:32633", subsample=0.5, random_state=9)
    assert calls["fit"] == {"res": (2.0, 2.0), "subsample": 0.5, "random_state": 9, "bin_statistic": np.nanmedian}
    assert nk.meta["outputs"]["affine"] == {"shift_x": -3.0, "shift_y": 4.0, "shift_z": 1.5}
    assert nk.to_translations() == (-3.0, 4.0, 1.5) and nk.to_rotations() == (0.0, 0.0, 0.0) and nk.is_affine
    out, tr2 = nk.apply(b, transform=tr, crs="EPSG:32633")
    assert tr2 == tr and calls["apply"] == (-3.0, 4.0, 1.5, (2.0, 2.0), True) and np.all(out == 2.5)
    out, tr3 = nk.apply(b, transform=tr, resample=False)
    assert tr3 == (2.0, 0.0, 97.0, 0.0, -2.0, 504.0) and calls["apply"][4] is False
    assert isinstance(nk.apply(b, 2.0), np.ndarray)          # resolution form: the array alone
    aligned, _ = coreg.NuthKaab().fit_and_apply(a, b, transform=tr, random_state=1)
    assert np.all(aligned == 2.5) and calls["fit"]["random_state"] == 1
    assert isinstance(coreg.NuthKaab().fit_and_apply(a, b, fit_kwargs={"resolution": 2.0}), np.ndarray)
    c = nk.copy()
    c.meta["outputs"]["affine"]["shift_x"] = 0.0
    assert nk.meta["outputs"]["affine"]["shift_x"] == -3.0

    class Affine:  # rasterio-like transform object
        def __init__(self, a, b, c, d, e, f):
            self.a, self.b, self.c, self.d, self.e, self.f = a, b, c, d, e, f

    out, t4 = nk.apply(b, transform=Affine(*tr), resample=False)
    assert isinstance(t4, Affine) and (t4.c, t4.f) == (97.0, 504.0)
    with pytest.raises(NotImplementedError):
        nk.fit(a, b, transform=tr, weights=np.ones((4, 5)))
    with pytest.raises(NotImplementedError):
        nk.apply(b, transform=tr, resampling="cubic")


def test_dem_integer_nodata_becomes_nan():
    """An integer raster with a nodata value must not hand nodata to the kernels as an elevation: geoutils gives upstream a
    masked array whose get_nanarray() is float with NaN (xdem/terrain.py:1140-1160 takes it from there)."""
    import xdem_amd

    z = np.arange(30, dtype=np.int16).reshape(5, 6)
    z[2, 3] = -9999
    d = xdem_amd.DEM(z, nodata=-9999)
    assert d.dtype == np.float32 and np.isnan(d.data[2, 3]) and np.count_nonzero(np.isnan(d.data)) == 1
    assert xdem_amd.DEM(np.arange(30, dtype=np.int16).reshape(5, 6), nodata=-9999).dtype == np.int16  # nothing to mask: unchanged


def test_mp_config_refusals_before_any_gpu_work():
    """Upstream's tiled call needs a Raster (terrain.py:436-437: TypeError with this message); checked on CPU because it
    fires before a context exists.  (What mp_config DOES is a GPU test: tests/test_terrain_gpu.py::test_mp_config...)"""
    from types import SimpleNamespace

    from xdem_amd import terrain as t

    dem = np.arange(12 * 14, dtype=np.float32).reshape(12, 14)
    with pytest.raises(TypeError, match="The DEM must be a Raster to use multiprocessing."):
        t.get_terrain_attribute(dem, "slope", resolution=1.0, mp_config=SimpleNamespace(chunk_size=200, outfile=None, cluster=None))
    with pytest.raises(TypeError, match="The DEM must be a Raster to use multiprocessing."):
        t.slope(dem, resolution=1.0, mp_config=SimpleNamespace(chunk_size=200, outfile=None, cluster=None))


def test_bin_statistic_dispatch():
    """NuthKaab(bin_statistic=...): the median and the mean run on the GPU, any other callable over the GPU's y values on the host
    (xdem_amd/coreg.py: _bin_statistic_id); anything that is not callable is a TypeError at construction, before any GPU work."""
    from xdem_amd import coreg

    assert coreg._bin_statistic_id(np.nanmedian) == 0 and coreg._bin_statistic_id(np.median) == 0 and coreg._bin_statistic_id("median") == 0
    assert coreg._bin_statistic_id(np.nanmean) == 1 and coreg._bin_statistic_id(np.mean) == 1
    assert coreg._bin_statistic_id(np.nanmax) == 2 and coreg._bin_statistic_id(lambda v: 0.0) == 2 and coreg._bin_statistic_id(np.sum) == 2
    with pytest.raises(TypeError):
        coreg._bin_statistic_id(0.5)
    with pytest.raises(TypeError):
        coreg.NuthKaab(bin_statistic="percentile")
    assert coreg.NuthKaab(bin_statistic=np.nanstd).meta["inputs"]["fitorbin"]["bin_statistic"] is np.nanstd


def test_native_ring_sampler_and_gather():
    """csrc/hostprep.hip (host-only native code of the variogram path; no GPU): xdemhip_host_ring_sample draws what the NumPy form
    `_draw_ring_pixels` specifies -- distinct valid members of each ring, as many as asked or the whole ring (then in raster order,
    identical to the enumeration), uniformly over the ring, the same for any number of threads -- and xdemhip_host_gather_points
    returns coordinates and values in the order given and in Morton order (a permutation of the same points that sorts their Z-order
    keys).  `equidistant_blocks_from_raster` on the native path: the block structure of the NumPy path."""
    import ctypes

    from xdem_amd import _lib
    from xdem_amd import spatialstats as ss

    L = _lib.host_library()
    i64p, dp = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)
    rng = np.random.default_rng(1)
    ny, nx, gsd = 300, 420, 2.0
    v = rng.normal(size=(ny, nx)).astype(np.float32)
    v[50:80, 100:200] = np.nan
    valid = np.isfinite(v)
    iy, ix = np.mgrid[0:ny, 0:nx]
    cases = [(10, 10, 0.0, 30.0, 50), (200, 150, 100.0, 160.0, 20000), (419, 299, 37.5, 400.0, 300), (5, 290, 250.0, 1000.0, 10**5),
             (100, 100, 0.0, 3.0, 100), (0, 0, 2000.0, 3000.0, 10), (210, 150, 100.0, 160.0, 200)]

    def sample(cxs, cys, los, his, n, seed, threads, mask=valid):
        cx, cy = np.ascontiguousarray(cxs, dtype=np.int64), np.ascontiguousarray(cys, dtype=np.int64)
        lo, hi = np.ascontiguousarray(los, dtype=np.float64), np.ascontiguousarray(his, dtype=np.float64)
        out = np.empty((cx.size, lo.size, n), dtype=np.int64)
        cnt = np.empty((cx.size, lo.size), dtype=np.int64)
        m8 = None if mask is None else np.ascontiguousarray(mask).view(np.uint8)
        rc = L.xdemhip_host_ring_sample(None if m8 is None else m8.ctypes.data, ny, nx, gsd, cx.size, cx.ctypes.data_as(i64p), cy.ctypes.data_as(i64p),
                                        lo.size, lo.ctypes.data_as(dp), hi.ctypes.data_as(dp), n, ctypes.c_uint64(seed), threads,
                                        out.ctypes.data_as(i64p), cnt.ctypes.data_as(i64p))
        assert rc == 0
        return out, cnt

    for cx, cy, lo, hi, n in cases:
        out, cnt = sample([cx], [cy], [lo], [hi], n, 3, 1)
        got = out[0, 0, :cnt[0, 0]]
        assert np.all(out[0, 0, cnt[0, 0]:] == -1)
        d = np.sqrt(((ix - cx) * gsd) ** 2 + ((iy - cy) * gsd) ** 2)
        full = np.flatnonzero((valid & (d >= lo) & (d < hi)).ravel())
        assert np.unique(got).size == got.size and np.isin(got, full).all() and got.size == min(n, full.size), (cx, cy, lo, hi, n)
        if n >= full.size:
            assert np.array_equal(got, full)   # the whole ring, in raster order
    # many (run, ring) jobs: independent of the number of threads; different seeds differ
    cxs, cys = rng.integers(0, nx, 9), rng.integers(0, ny, 9)
    los, his = [0.0, 0.0, 40.0, 80.0, 160.0], [40.0, 40.0, 80.0, 160.0, 320.0]
    o1, c1 = sample(cxs, cys, los, his, 150, 77, 1)
    o4, c4 = sample(cxs, cys, los, his, 150, 77, 4)
    assert np.array_equal(o1, o4) and np.array_equal(c1, c4)
    assert not np.array_equal(o1, sample(cxs, cys, los, his, 150, 78, 4)[0])
    assert not np.array_equal(o1[:, 0], o1[:, 1])   # (the centre disk and the first ring: same bounds, streams of their own)
    # uniformity: 400 seeds x 200 of a ring of ~12000 pixels hit every octant of the ring about equally
    cx, cy, lo, hi = 210, 150, 100.0, 160.0
    hits = np.zeros(8)
    for seed in range(400):
        g = sample([cx], [cy], [lo], [hi], 200, seed, 1, mask=None)[0][0, 0]
        ang = np.arctan2((g // nx) - cy, (g % nx) - cx)
        hits += np.bincount(((ang + np.pi) / (2 * np.pi) * 8).astype(int) % 8, minlength=8)
    assert hits.sum() == 400 * 200 and np.all(np.abs(hits / hits.mean() - 1) < 0.05)
    # bad arguments
    assert L.xdemhip_host_ring_sample(None, ny, nx, gsd, 1, np.array([nx], dtype=np.int64).ctypes.data_as(i64p), np.array([0], dtype=np.int64).ctypes.data_as(i64p),
                                      1, np.array([0.0]).ctypes.data_as(dp), np.array([5.0]).ctypes.data_as(dp), 5, ctypes.c_uint64(1), 1,
                                      np.empty(5, dtype=np.int64).ctypes.data_as(i64p), np.empty(1, dtype=np.int64).ctypes.data_as(i64p)) == -1
    # gather: order given and Morton order
    for dtype in (np.float32, np.float64):
        vals = rng.normal(size=(ny, nx)).astype(dtype)
        off = np.array([0, 5, 5, 2005, 2007], dtype=np.int64)
        flat = rng.choice(ny * nx, 2007, replace=False).astype(np.int64)
        o = [np.empty(2007), np.empty(2007), np.empty(2007, dtype=dtype), np.empty(2007), np.empty(2007), np.empty(2007, dtype=dtype)]
        rc = L.xdemhip_host_gather_points(vals.ctypes.data, _lib.F32 if dtype == np.float32 else _lib.F64, nx, gsd, 4, off.ctypes.data_as(i64p),
                                          flat.ctypes.data_as(i64p), 3, o[0].ctypes.data_as(dp), o[1].ctypes.data_as(dp), o[2].ctypes.data,
                                          o[3].ctypes.data_as(dp), o[4].ctypes.data_as(dp), o[5].ctypes.data)
        assert rc == 0
        assert np.array_equal(o[0], (flat % nx) * gsd) and np.array_equal(o[1], (flat // nx) * gsd) and np.array_equal(o[2], vals.reshape(-1)[flat])
        for b in range(4):
            sl = slice(off[b], off[b + 1])
            key = lambda x, y, val: sorted(zip(x.tolist(), y.tolist(), val.tolist()))
            assert key(o[0][sl], o[1][sl], o[2][sl]) == key(o[3][sl], o[4][sl], o[5][sl])   # a permutation of the same points
            if sl.stop - sl.start >= 3:
                order = ss._morton_order(o[3][sl], o[4][sl])
                x, y = o[3][sl], o[4][sl]
                qx = ((x - x.min()) * (65535.0 / (x.max() - x.min()))).astype(np.uint32)
                qy = ((y - y.min()) * (65535.0 / (y.max() - y.min()))).astype(np.uint32)
                code = sum(((qx >> k) & 1).astype(np.uint64) << np.uint64(2 * k) | ((qy >> k) & 1).astype(np.uint64) << np.uint64(2 * k + 1) for k in range(16))
                assert np.all(np.diff(code.astype(np.int64)) >= 0) and order is not None
    # the product's sampler on the native path: block structure of the NumPy path (centre disk x rings, inner to outer, values of the pixels)
    centres, centres_np = [], []
    blocks = ss.equidistant_blocks_from_raster(v, gsd, 3, 30, 0.002, np.random.default_rng(9), centres_out=centres)
    blocks_np = ss.equidistant_blocks_from_raster(v, gsd, 3, 30, 0.002, np.random.default_rng(9), centres_out=centres_np, native=False)
    assert type(blocks).__name__ == "_Blocks" and blocks.packed is not None and blocks.packed_sorted is not None and centres == centres_np
    maxlag = float(np.hypot((nx - 1) * gsd, (ny - 1) * gsd))
    r0, radii = ss._equidistant_radii(30, 0.002, gsd, maxlag)
    assert len(blocks) == len(blocks_np) == 3
    for (ax, ay, av, bx, by, bv), (cxi, cyi), other in zip(blocks, centres, blocks_np):
        assert ax.size == other[0].size and bx.size == other[3].size
        assert np.array_equal(av, v[(ay / gsd).astype(int), (ax / gsd).astype(int)]) and np.isfinite(av).all() and np.isfinite(bv).all()
        assert np.all(np.hypot(ax - cxi * gsd, ay - cyi * gsd) < r0)
        k = np.digitize(np.hypot(bx - cxi * gsd, by - cyi * gsd), radii) - 1
        assert np.all(np.diff(k) >= 0)
        assert np.array_equal(np.bincount(k, minlength=len(radii) - 1), np.bincount(np.digitize(np.hypot(other[3] - cxi * gsd, other[4] - cyi * gsd), radii) - 1, minlength=len(radii) - 1))
    # packed arrays = the blocks, concatenated
    a_off, pax, pay, pav, b_off, pbx, pby, pbv = blocks.packed
    assert np.array_equal(pax, np.concatenate([b[0] for b in blocks])) and np.array_equal(pbv, np.concatenate([b[5] for b in blocks]))
    assert a_off[-1] == pax.size and b_off[-1] == pbx.size


def test_native_count_finite_equals_numpy():
    """xdemhip_host_count_finite (the NaN filter of sample_empirical_variogram on the library's host threads) against np.isfinite:
    NaN payloads, both infinities, zeros of both signs, subnormals, the largest finite values; float32 / float64; more elements than
    one piece of the thread loop; the NumPy fall-back for arrays the native code does not take."""
    from xdem_amd import spatialstats

    rng = np.random.default_rng(0)
    for dt, bits in ((np.float32, np.uint32), (np.float64, np.uint64)):
        v = rng.standard_normal(5_000_003).astype(dt)
        v[rng.integers(0, v.size, 1000)] = np.nan
        v[rng.integers(0, v.size, 500)] = np.inf
        v[rng.integers(0, v.size, 500)] = -np.inf
        special = np.array([0.0, -0.0, np.finfo(dt).max, -np.finfo(dt).max, np.finfo(dt).tiny / 4, np.nan, -np.nan], dtype=dt)
        v[:special.size] = special
        v[special.size:special.size + 2] = np.array([np.iinfo(bits).max, np.iinfo(bits).max >> 1], dtype=bits).view(dt)   # NaNs with full payloads
        want = np.isfinite(v)
        n, mask = spatialstats._count_finite(v)
        assert n == int(want.sum()) and mask is None
        n, mask = spatialstats._count_finite(v, want_mask=True)
        assert n == int(want.sum()) and mask.dtype == np.bool_ and np.array_equal(mask, want)
        n, mask = spatialstats._count_finite(v[::2], want_mask=True)   # (not contiguous: NumPy)
        assert n == int(want[::2].sum()) and np.array_equal(mask, want[::2])
    assert spatialstats._count_finite(np.empty(0, dtype=np.float32)) == (0, None)
    assert spatialstats._count_finite(np.arange(10, dtype=np.float16))[0] == 10


def test_nextprod_fft_rule_of_the_texture_path():
    """xdem_amd.terrain.freq._nextprod_fft (the reference's helper of the same name, xdem/terrain/freq.py:33-61): the known answers
    of the reference's own test (tests/test_terrain/test_freq.py:198-215) and the oracle's restatement over a range."""
    import terrain_oracle as to
    import xdem_amd

    f = xdem_amd.terrain.freq._nextprod_fft
    assert [f(n) for n in (0, 1, 2, 3, 10, 20, 100, 1000, 1024)] == [1, 1, 2, 4, 16, 32, 128, 1024, 1024]
    for n in list(range(1, 1300)) + [1025, 2047, 2049, 4801, 10007, 16385, 40000, 65537]:
        m = f(n)
        assert m == to._next_fft_len(n) and m >= n
        if n > 1024:
            r = m
            for p in (2, 3, 5, 7):
                while r % p == 0:
                    r //= p
            assert r == 1


def test_host_half_of_binned_statistics_equals_scipy():
    """xdem_amd/_binstat_host.binned_statistic_host -- what nd_binning / NuthKaab apply to bin numbers from the GPU for statistics the
    device does not evaluate -- against scipy.stats.binned_statistic / _2d themselves: SciPy's names, the NumPy function objects it
    answers itself, callables (one that fails on an empty bin, one that answers it), float32 and float64 values, bit for bit."""
    import scipy.stats

    from xdem_amd._binstat_host import binned_statistic_host as host

    def p90(a):
        return np.percentile(a, 90)

    def spread(a):
        return float(np.max(a) - np.min(a)) if len(a) else -1.0

    rng = np.random.default_rng(0)
    x, y = rng.uniform(0, 10, 6000).astype(np.float32), rng.uniform(-1, 1, 6000)
    x[:300] = 9.99   # (a crowded bin; bins 3 and 4 of the 2-D grid stay empty below)
    stats = ["count", "mean", "std", "sum", "min", "max", "median", np.mean, np.std, np.sum, np.min, np.max, np.median, np.nanmean, np.nanstd,
             np.nanmedian, p90, spread, np.ptp]
    for dt in (np.float32, np.float64):
        v = rng.normal(size=x.size).astype(dt)
        for st in stats:
            r = scipy.stats.binned_statistic(x, v, statistic=st, bins=12)
            b = np.minimum(r.binnumber - 1, 11)
            assert np.array_equal(host(st, b, v, 12), r.statistic, equal_nan=True), (dt, st)
            keep = (y < 0.2) | (x > 5)      # leaves bins of the 5 x 4 grid empty
            r2 = scipy.stats.binned_statistic_2d(x[keep], y[keep], v[keep], statistic=st, bins=(5, 4), range=[(0, 10), (-1, 1)], expand_binnumbers=True)
            b2 = (np.minimum(r2.binnumber[0] - 1, 4)) * 4 + np.minimum(r2.binnumber[1] - 1, 3)
            assert np.array_equal(host(st, b2, v[keep], 20), r2.statistic.ravel(), equal_nan=True), (dt, st, "2d")
    with pytest.raises(ValueError, match="invalid statistic 'mode'"):
        host("mode", np.zeros(3, np.intp), np.ones(3), 2)
