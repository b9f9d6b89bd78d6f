"""GPU parity tests of the variogram path: pair kernels (through the C-ABI) vs the CPU oracle.
Integer work (lag-class membership counts) bit-exact; Dowd (exact median) bit-exact; Matheron / Cressie sums within
1e-12 relative (float64 accumulation order differs)."""
import numpy as np
import pytest

from conftest import decided

import variogram_oracle as vo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ss():
    from xdem_amd import spatialstats as s

    return s


def _pts(n, seed, dtype, extent=500.0, grid=True):
    r = np.random.default_rng(seed)
    if grid:
        x = r.integers(0, int(extent), n).astype(np.float64)
        y = r.integers(0, int(extent), n).astype(np.float64)
    else:
        x, y = r.uniform(0, extent, n), r.uniform(0, extent, n)
    v = (np.sin(x / 40.0) + 0.3 * r.normal(size=n)).astype(dtype)
    return x, y, v


EDGES = [float(e) for e in vo.default_bin_edges(1.0, 700.0)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("estimator", ["matheron", "cressie", "dowd"])
@pytest.mark.parametrize("mode", ["pdist", "cdist"])
def test_pairs_vs_oracle(ss, dtype, estimator, mode):
    if mode == "pdist":
        blocks = [_pts(700, 1, dtype), _pts(301, 2, dtype, grid=False), _pts(2, 3, dtype), _pts(1, 4, dtype)]
    else:
        blocks = [_pts(300, 1, dtype) + _pts(1500, 2, dtype), _pts(257, 3, dtype, grid=False) + _pts(4097, 4, dtype, grid=False),
                  _pts(1, 5, dtype) + _pts(3, 6, dtype)]
    exp, count = ss.empirical_variogram_pairs(blocks, EDGES, estimator)
    exp_o, count_o = vo.empirical_variogram_blocks(blocks, EDGES, estimator)
    assert np.array_equal(count, count_o)  # class membership: bit-exact
    assert np.array_equal(np.isnan(exp), np.isnan(exp_o))
    ok = np.isfinite(exp_o)
    if estimator == "dowd":
        assert np.array_equal(exp[ok], exp_o[ok])  # exact medians
    else:
        assert np.allclose(exp[ok], exp_o[ok], rtol=1e-12, atol=0)


def test_edge_inclusivity_and_nan(ss):
    # distances exactly on an edge open the next class; NaN values never pair; beyond the last edge is dropped
    ax, ay, av = np.array([0.0]), np.array([0.0]), np.array([1.0], np.float32)
    bx = np.array([1.0, 2.0, 2.5, 5.0, 3.0, 4.0])
    by = np.zeros(6)
    bv = np.array([2.0, 4.0, 8.0, 16.0, np.nan, 3.0], np.float32)
    edges = [1.0, 2.0, 5.0]
    exp, count = ss.empirical_variogram_pairs([(ax, ay, av, bx, by, bv)], edges, "matheron")
    # ([e_{k-1}, e_k) by default; (e_{k-1}, e_k] where a decision file says vario_edge = 1: d = 1 joins class 0, d = 5 class 2)
    if decided("vario_edge") == 0:
        assert count.tolist() == [0, 1, 3] and np.isnan(exp[0])
    else:
        assert count.tolist() == [1, 1, 3]
    # the oracle has no NaN filter of its own (the reference drops NaN values before pairing): compare on the finite points
    keep = np.isfinite(bv)
    exp_o, count_o = vo.empirical_variogram_blocks([(ax, ay, av, bx[keep], by[keep], bv[keep])], edges, "matheron")
    assert np.array_equal(count, count_o) and np.allclose(exp[1:], exp_o[1:], rtol=1e-14)
    # 3-4-5 triangle: d == 5.0 exactly is outside [.., 5) -- and inside (.., 5]
    exp, count = ss.empirical_variogram_pairs([(np.array([0.0]), np.array([0.0]), av, np.array([3.0]), np.array([4.0]), av)], edges, "dowd")
    assert count.sum() == (0 if decided("vario_edge") == 0 else 1) and np.isnan(exp[:2]).all()


def test_dowd_even_odd_duplicates(ss):
    x = np.arange(40, dtype=np.float64)
    y = np.zeros(40)
    v = np.repeat(np.arange(10), 4).astype(np.float32)  # many duplicate differences
    for n in (5, 6, 39, 40):
        blk = [(x[:n], y[:n], v[:n])]
        exp, count = ss.empirical_variogram_pairs(blk, [2.0, 5.0, 100.0], "dowd")
        exp_o, count_o = vo.empirical_variogram_blocks(blk, [2.0, 5.0, 100.0], "dowd")
        assert np.array_equal(count, count_o) and np.array_equal(exp, exp_o, equal_nan=True)


@pytest.mark.parametrize("method", ["cdist_equidistant", "cdist_point", "pdist_point"])
@pytest.mark.parametrize("estimator", ["matheron", "dowd"])
def test_sample_empirical_variogram_end_to_end(ss, method, estimator):
    from xdem_amd.synth import fbm_numpy

    vals = fbm_numpy((90, 120), hurst=0.3, seed=45, mean=0.0, std=2.0)
    vals[10:14, 20:30] = np.nan
    df = ss.sample_empirical_variogram(vals, gsd=5.0, subsample=120, subsample_method=method, random_state=42,
                                       estimator=estimator)
    # oracle composition with the same host preparation and RNG protocol
    coords, extent, maxlag = vo.grid_coords_extent_maxlag(vals.shape, 5.0)
    edges = vo.default_bin_edges(5.0, maxlag)
    flat = vals.flatten()
    valid = np.isfinite(flat)
    seed = list(np.random.default_rng(42).choice(1, 1, replace=False))[0]
    rng = np.random.default_rng(seed)
    if method == "cdist_equidistant":
        # the draws of the ring sampler are the product's own (raster form: tests/test_cabi_and_host.py checks it against the
        # enumeration of the rings); the pair arithmetic on those blocks is what the oracle checks here
        runs, samples, ratio = vo.choose_cdist_equidistant_sampling_parameters(120, extent, vals.shape)
        img = flat.reshape(vals.shape[1], vals.shape[0])   # upstream's meshgrid convention: shape[0] along x
        blocks = ss.equidistant_blocks_from_raster(img, 5.0, runs, samples, ratio, rng, valid2d=np.isfinite(img))
        assert len(blocks) == runs
    elif method == "cdist_point":
        idx = np.flatnonzero(valid)
        a = rng.choice(idx, 120, replace=False)
        b = rng.choice(idx, 120, replace=False)
        blocks = [(coords[a, 0], coords[a, 1], flat[a], coords[b, 0], coords[b, 1], flat[b])]
    else:
        a = rng.choice(np.flatnonzero(valid), 120, replace=False)
        blocks = [(coords[a, 0], coords[a, 1], flat[a])]
    exp_o, count_o = vo.empirical_variogram_blocks(blocks, edges, estimator)
    assert list(df.columns) == ["exp", "lags", "count", "err_exp"]
    assert len(df) == len(edges) - 1 and df["count"].dtype == np.int64
    assert np.array_equal(df["lags"].values, np.array(edges[:-1]))
    assert np.array_equal(df["count"].values, count_o[:-1])
    assert np.allclose(df["exp"].values, exp_o[:-1], rtol=1e-12, equal_nan=True)
    assert df["err_exp"].isna().all()
    assert df["count"].sum() > 1000


@pytest.mark.parametrize("method", ["pdist_disk", "pdist_ring"])
def test_multi_range_pdist_methods(ss, method):
    """"pdist_disk" / "pdist_ring": one pdist variogram per range, rows of all ranges kept, the last lag of each removed
    (the reference drops by index label after the concat).  Every range against the oracle on the product's own subsample."""
    from xdem_amd.synth import fbm_numpy

    vals = fbm_numpy((90, 120), hurst=0.3, seed=45, mean=0.0, std=2.0)
    vals[10:14, 20:30] = np.nan
    gsd = 5.0
    df = ss.sample_empirical_variogram(vals, gsd=gsd, subsample=150, subsample_method=method, random_state=3, estimator="dowd")
    coords, extent, maxlag = vo.grid_coords_extent_maxlag(vals.shape, gsd)
    edges = vo.default_bin_edges(gsd, maxlag)
    flat = vals.flatten()
    seed = list(np.random.default_rng(3).choice(1, 1, replace=False))[0]
    sels = ss._pdist_multi_range_subsamples(np.isfinite(flat), vals.shape, 150, method, gsd, maxlag, None, seed)
    ranges = [50.0, 100.0, 200.0, maxlag]  # 10 gsd doubling while below maxlag / 2 (= 372), then maxlag
    assert len(sels) == len(ranges) and all(np.isfinite(flat[s_]).all() for s_ in sels)
    nb = len(edges)
    assert len(df) == len(sels) * (nb - 1)
    for j, sel in enumerate(sels):
        exp_o, count_o = vo.empirical_variogram_blocks([(coords[sel, 0], coords[sel, 1], flat[sel])], edges, "dowd")
        part = df.iloc[j * (nb - 1):(j + 1) * (nb - 1)]
        assert np.array_equal(part["lags"].values, np.array(edges[:-1]))
        assert np.array_equal(part["count"].values, count_o[:-1])
        assert np.allclose(part["exp"].values, exp_o[:-1], rtol=1e-12, equal_nan=True)
    # (no geometric claim on the lags of a range: for a non-square grid upstream's meshgrid coordinates do not follow the
    # C-order flattening of the values -- reproduced as is, see sample_empirical_variogram -- so a disk of pixels is not
    # compact in coordinate space)


def test_pair_passes_split_over_several_launches(ss):
    """A pass over more tiles than one HIP dispatch holds (total work-items are a 32-bit quantity: 5e13 pairs, SURVEY 8d's C5
    reading A, need it) goes out as several launches.  Forced here with a small per-launch cap: every estimator must match
    the single-launch result (integer work bit for bit)."""
    from xdem_amd import _lib

    rng = np.random.default_rng(21)
    blocks = []
    for _ in range(3):
        ax, ay = rng.uniform(0, 4000, 1500), rng.uniform(0, 4000, 1500)
        bx, by = rng.uniform(0, 4000, 20000), rng.uniform(0, 4000, 20000)
        av = np.round(np.sin(ax / 300) + 0.3 * rng.normal(size=1500), 2).astype(np.float32)
        bv = np.round(np.sin(bx / 300) + 0.3 * rng.normal(size=20000), 2).astype(np.float32)
        blocks.append((ax, ay, av, bx, by, bv))
    edges = np.geomspace(np.sqrt(2), 5700.0, 25)
    ctx = _lib.default_context()
    want = {est: ss.empirical_variogram_pairs(blocks, edges, est) for est in ("matheron", "cressie", "dowd")}
    try:
        for cap in (1, 7):
            ctx.set_option("pairs_launch_cap", cap)
            for est, (e0, c0) in want.items():
                e1, c1 = ss.empirical_variogram_pairs(blocks, edges, est)
                assert np.array_equal(c1, c0), (cap, est)
                if est == "dowd":
                    assert np.array_equal(e1, e0, equal_nan=True), cap
                else:
                    assert np.allclose(e1, e0, rtol=1e-12, equal_nan=True), (cap, est)
    finally:
        ctx.set_option("pairs_launch_cap", 0)


def test_multiple_runs_aggregate(ss):
    from xdem_amd.synth import fbm_numpy

    vals = fbm_numpy((64, 64), hurst=0.3, seed=1, mean=0.0, std=1.0)
    df = ss.sample_empirical_variogram(vals, gsd=1.0, subsample=60, n_variograms=3, random_state=7)
    assert np.isfinite(df["err_exp"].values).any() and (df["count"] >= 0).all()
    with pytest.raises(ValueError, match="2D when using"):
        ss.sample_empirical_variogram(vals.ravel(), gsd=1.0)
    with pytest.raises(TypeError, match="subsampling method"):
        ss.sample_empirical_variogram(vals, gsd=1.0, subsample_method="nope")


def test_named_binning_even(ss):
    """scikit-gstat's named binning 'even' (skgstat.binning.even_width_lags: n_lags classes of equal width up to maxlag CLIPPED to the
    largest sampled pair distance; n_lags defaults to 10): equally wide classes whose top is a distance between two sampled points
    -- below the extent diagonal upstream passes as maxlag -- and the same call as the Iterable of those right edges; an explicit
    maxlag below every sampled distance is kept; the binnings that depend on the sampled distances in other ways are refused by
    name (reference: xdem/spatialstats.py:1396-1403 warns about exactly those).  (`_max_pair_distance` against brute force: CPU suite.)"""
    from xdem_amd.synth import fbm_numpy

    vals = fbm_numpy((64, 64), hurst=0.3, seed=2, mean=0.0, std=1.0)
    maxlag = float(np.sqrt(63.0**2 + 63.0**2))
    for kw, n in (({}, 10), ({"n_lags": 7}, 7)):
        a = ss.sample_empirical_variogram(vals, gsd=1.0, subsample=80, random_state=3, bin_func="even", **kw)
        top = float(a["lags"].values[0]) * n
        assert 0.5 * maxlag < top < maxlag and abs(top * top - round(top * top)) < 1e-6   # a distance between two lattice points
        b = ss.sample_empirical_variogram(vals, gsd=1.0, subsample=80, random_state=3, bin_func=np.linspace(0, top, n + 1)[1:])
        # (n - 1 rows: upstream drops the last lag class, xdem/spatialstats.py:1541)
        assert len(a) == len(b) == n - 1 and np.allclose(a["lags"].values, np.linspace(0, top, n + 1)[1:-1], rtol=1e-15, atol=0)
        # (Matheron sums are float64 atomics: equal to rounding between two calls, not bit for bit)
        assert np.allclose(a["exp"].values, b["exp"].values, rtol=1e-12, atol=0, equal_nan=True) and np.array_equal(a["count"].values, b["count"].values)
    c = ss.sample_empirical_variogram(vals, gsd=1.0, subsample=80, random_state=3, bin_func="even", maxlag=20.0)
    assert np.array_equal(c["lags"].values, np.linspace(0, 20.0, 11)[1:-1])
    with pytest.raises(NotImplementedError, match="only 'even'"):
        ss.sample_empirical_variogram(vals, gsd=1.0, subsample=80, bin_func="uniform")


def test_large_pair_count_properties(ss):
    """2e4-point pdist (2e8 pairs): counts sum to N(N-1)/2; a cdist block gives the same classes with A and B swapped."""
    x, y, v = _pts(20000, 9, np.float32, extent=4000.0, grid=False)
    edges = [float(e) for e in vo.default_bin_edges(1.0, 7000.0)]
    exp, count = ss.empirical_variogram_pairs([(x, y, v)], edges, "matheron")
    assert count.sum() == 20000 * 19999 // 2
    h = 10000
    e1, c1 = ss.empirical_variogram_pairs([(x[:h], y[:h], v[:h], x[h:], y[h:], v[h:])], edges, "dowd")
    e2, c2 = ss.empirical_variogram_pairs([(x[h:], y[h:], v[h:], x[:h], y[:h], v[:h])], edges, "dowd")
    assert np.array_equal(c1, c2) and np.array_equal(e1, e2, equal_nan=True)
    assert c1.sum() == h * h


def test_C5_full_size_properties(ss):
    """BASELINE config[4] scale (reading B of SURVEY 8d): 1e7 sampled points in 100 centre x ring blocks
    (100 x 9091 x 90910 = 8.3e10 pairs), 50 lag classes, Matheron and exact Dowd."""
    rng = np.random.default_rng(45)
    runs, samples, rings = 100, 9091, 10
    L = 20000.0
    blocks = []
    for _ in range(runs):
        ax, ay = rng.uniform(0, L, samples), rng.uniform(0, L, samples)
        bx, by = rng.uniform(0, L, samples * rings), rng.uniform(0, L, samples * rings)
        av = (np.sin(ax / 900.0) + 0.2 * rng.normal(size=samples)).astype(np.float32)
        bv = (np.sin(bx / 900.0) + 0.2 * rng.normal(size=samples * rings)).astype(np.float32)
        blocks.append((ax, ay, av, bx, by, bv))
    maxlag = float(np.hypot(L, L))
    edges = np.geomspace(np.sqrt(2), maxlag, 50)
    total = runs * samples * samples * rings
    exp_m, cnt_m = ss.empirical_variogram_pairs(blocks, edges, "matheron")
    exp_d, cnt_d = ss.empirical_variogram_pairs(blocks, edges, "dowd")
    assert np.array_equal(cnt_m, cnt_d)                      # two independent kernels, same class membership
    assert cnt_m.sum() == total                               # every pair is below the diagonal maxlag, none is lost
    assert np.all(np.diff(cnt_m[10:40].astype(np.float64)) > 0)  # annulus areas grow geometrically below the domain scale
    ok = cnt_m > 1000
    assert np.all(exp_m[ok] > 0) and np.all(exp_d[ok] > 0)
    # a 1 % subset of the blocks against the oracle (exact counts; Dowd exact)
    import variogram_oracle as vo

    sub = [tuple(a[:300] if i in (0, 1, 2) else a[:1200] for i, a in enumerate(b)) for b in blocks[:2]]
    e_g, c_g = ss.empirical_variogram_pairs(sub, edges, "dowd")
    e_o, c_o = vo.empirical_variogram_blocks(sub, edges, "dowd")
    assert np.array_equal(c_g, c_o) and np.array_equal(e_g, e_o, equal_nan=True)


def test_C5_sampler_geometry_lattice_bracket_kernels(ss):
    """BASELINE C5 (reading B) on the input SURVEY 8d names -- fBm(H = 0.3) values on a 20000^2 grid, points drawn by the
    equidistant disk / ring sampler, i.e. INTEGER-LATTICE coordinates and >= 4e9 pairs: the configuration bench.py times, whose
    exact-Dowd route runs `pairs_kernel<float, OP_BRACKET, GRID>` (lattice distances x bracketed counting pass).
    (i) lattice kernels (option vario_grid = 1) == float64-coordinate kernels (0): class counts and exact Dowd medians
        identical, in every selection mode (0 bracketed, 1 plain digit passes, 2 forced bracket miss) and with the pair
        passes split into many launches (pairs_launch_cap);
    (ii) a ~1 % subset of the blocks against the CPU oracle (counts and medians exact, Matheron to 1e-12);
    (iii) Matheron and Dowd routes count the same pairs per class and together every pair below the last edge."""
    import torch

    from xdem_amd import _lib
    from xdem_amd.synth import c5_variogram_blocks

    blocks, edges = c5_variogram_blocks("cuda", runs=100, samples=9091)
    assert len(blocks) == 100 and all(b[0].size == 9091 and 5 * 9091 <= b[3].size <= 11 * 9091 for b in blocks)
    for b in blocks[:3]:
        assert all(np.array_equal(c, np.round(c)) for c in (b[0], b[1], b[3], b[4]))  # raster pixels: lattice coordinates
    ctx = _lib.default_context()
    total = sum(b[0].size * b[3].size for b in blocks)
    assert total >= 4_000_000_000
    res = {}
    try:
        for grid in (1, 0):
            ctx.set_option("vario_grid", grid)
            ps = ss.PairSet(blocks, edges, ctx)
            try:
                assert ps.n_pairs == total
                s_m, c_m = ps.sums(0)
                for mode in (0, 1, 2):
                    ctx.set_option("selection", mode)
                    res[(grid, mode)] = ss.class_medians(ps)
                    assert np.array_equal(res[(grid, mode)][1], c_m), (grid, mode)      # (iii) two routes, same membership
                ctx.set_option("selection", 0)
                if grid == 1:
                    ctx.set_option("pairs_launch_cap", 3000)
                    res["cap"] = ss.class_medians(ps)
                    s_cap, c_cap = ps.sums(0)
                    ctx.set_option("pairs_launch_cap", 0)
                    assert np.array_equal(c_cap, c_m) and np.allclose(s_cap, s_m, rtol=1e-12, atol=0)
                    # the counting pass per pair on the sorted copy (round 3's form) instead of run-length (round 4)
                    ctx.set_option("vario_runs", 0)
                    res["per-pair counters"] = ss.class_medians(ps)
                    ctx.set_option("vario_runs", 1)
                    # ... and on the caller's order (no sorted copy linked)
                    ctx.check(ctx._L.xdemhip_pairs_link_sorted(ps.handle_sel, None))
                    res["unlinked"] = ss.class_medians(ps)
                    ctx.check(ctx._L.xdemhip_pairs_link_sorted(ps.handle_sel, ps.handle))
                res[("sums", grid)] = (s_m, c_m)
            finally:
                ps.close()
    finally:
        ctx.set_option("vario_grid", 1)
        ctx.set_option("selection", 0)
        ctx.set_option("pairs_launch_cap", 0)
        ctx.set_option("vario_runs", 1)
    med0, cnt0 = res[(1, 0)]
    for k, (med, cnt) in res.items():
        if k[0] == "sums":
            continue
        assert np.array_equal(cnt, cnt0), k
        assert np.array_equal(med, med0, equal_nan=True), k
    assert np.array_equal(res[("sums", 1)][1], res[("sums", 0)][1])
    assert np.allclose(res[("sums", 1)][0], res[("sums", 0)][0], rtol=1e-12, atol=0)
    # every pair lies below the extent diagonal (= the last edge) except pairs AT it (none: corner to corner only)
    assert cnt0.sum() == total
    # (ii) a subset against the oracle: 3 blocks cut to 400 x 8000 points (9.6e6 pairs)
    sub = [tuple(a[:400] if i < 3 else a[:: max(1, a.size // 8000)][:8000] for i, a in enumerate(b)) for b in blocks[:3]]
    for est in ("dowd", "matheron"):
        e_g, c_g = ss.empirical_variogram_pairs(sub, edges, est)
        e_o, c_o = vo.empirical_variogram_blocks(sub, edges, est)
        assert np.array_equal(c_g, c_o)
        if est == "dowd":
            assert np.array_equal(e_g, e_o, equal_nan=True)
        else:
            assert np.allclose(e_g, e_o, rtol=1e-12, atol=0, equal_nan=True)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_class_medians_routes_agree(ss, dtype):
    """xdemhip_pairs_medians (device-resident selection; bracketed for >= 4e9 pairs) vs the host-driven pass-by-pass
    selection over the same C-ABI histograms, in every selection mode (0 auto, 1 plain passes, 2 bracket-miss fall-back):
    identical medians and counts.  4.4e9 pairs (4 blocks of 16384 x 67000 points), values with many ties."""
    from xdem_amd import _lib

    rng = np.random.default_rng(31)
    blocks = []
    for _ in range(4):
        na, nbp = 16384, 67000
        ax, ay = rng.uniform(0, 20000, na), rng.uniform(0, 20000, na)
        bx, by = rng.uniform(0, 20000, nbp), rng.uniform(0, 20000, nbp)
        av = np.round(np.sin(ax / 900.0) + 0.2 * rng.normal(size=na), 3).astype(dtype)
        bv = np.round(np.sin(bx / 900.0) + 0.2 * rng.normal(size=nbp), 3).astype(dtype)
        blocks.append((ax, ay, av, bx, by, bv))
    edges = np.geomspace(np.sqrt(2), np.hypot(20000, 20000), 50)
    ctx = _lib.default_context()
    ps = ss.PairSet(blocks, edges, ctx)
    try:
        assert ps.n_pairs >= 4_000_000_000
        ref_med, ref_cnt = ss.class_medians_host_driven(ps)
        assert ref_cnt.sum() == ps.n_pairs
        for mode in (0, 1, 2):
            ctx.set_option("selection", mode)
            med, cnt = ss.class_medians(ps)
            assert np.array_equal(cnt, ref_cnt), mode
            assert np.array_equal(med, ref_med, equal_nan=True), mode
    finally:
        ctx.set_option("selection", 0)
        ps.close()


def test_class_medians_through_reduction_hook_single_rank(ss):
    """The sharded route (integer histograms / counters / successor keys all-reduced through the library hook with
    torch.distributed) on a 1-rank NCCL group equals the plain single-process result."""
    import os

    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29613")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        blocks = [_pts(300, 1, np.float32) + _pts(1500, 2, np.float32), _pts(700, 3, np.float32)[:3] + _pts(901, 4, np.float32)]
        want = ss.empirical_variogram_pairs(blocks, EDGES, "dowd")
        got = ss.empirical_variogram_pairs(blocks, EDGES, "dowd", group=dist.group.WORLD)
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0], equal_nan=True)
        got_m = ss.empirical_variogram_pairs(blocks, EDGES, "matheron", group=dist.group.WORLD)
        want_m = ss.empirical_variogram_pairs(blocks, EDGES, "matheron")
        assert np.array_equal(got_m[1], want_m[1]) and np.allclose(got_m[0], want_m[0], rtol=1e-12, atol=0, equal_nan=True)  # (float64 atomics: order varies)
    finally:
        if created:
            dist.destroy_process_group()


def test_randomised_pair_sets_vs_oracle(ss):
    """Seeded sweep: random block structures (pdist / cdist, empty and 1-point blocks), both dtypes, all estimators, edge lists
    that take the class-lookup table (geometric), the binary search (dense linear edges, > 127 classes) and few-class paths,
    NaN values, duplicated points (zero distances) -- counts bit-exact, Dowd bit-exact, sums within 1e-12."""
    rng = np.random.default_rng(777)
    for trial in range(40):
        dtype = rng.choice([np.float32, np.float64])
        mode = rng.choice(["pdist", "cdist"])
        nblk = int(rng.integers(1, 5))
        extent = float(rng.choice([50.0, 500.0, 5000.0]))

        def pts(n):
            if rng.uniform() < 0.5:
                x, y = rng.integers(0, int(extent), n).astype(np.float64), rng.integers(0, int(extent), n).astype(np.float64)
            else:
                x, y = rng.uniform(0, extent, n), rng.uniform(0, extent, n)
            v = np.round(np.sin(x / (extent / 8)) + 0.3 * rng.normal(size=n), int(rng.integers(1, 6))).astype(dtype)
            if n > 3 and rng.uniform() < 0.3:
                v[rng.integers(0, n)] = np.nan
            return x, y, v

        blocks = []
        for _ in range(nblk):
            na = int(rng.choice([0, 1, 2, 37, 300, 1100]))
            if mode == "pdist":
                blocks.append(pts(na))
            else:
                blocks.append(pts(na) + pts(int(rng.choice([0, 1, 5, 260, 4100]))))
        kind = rng.integers(0, 4)
        if kind == 0:
            edges = [float(e) for e in vo.default_bin_edges(1.0, 1.5 * extent)]
        elif kind == 1:
            edges = list(np.linspace(extent / 300, 1.2 * extent, 200))      # > 127 classes: binary search, 2 LDS sweeps
        elif kind == 2:
            edges = list(np.linspace(extent / 40, extent, 40))              # dense linear edges: several thresholds per binade cell
        else:
            edges = [extent / 10, extent / 3, extent]
        est = str(rng.choice(["matheron", "cressie", "dowd"]))
        # the oracle pairs every value; NaN values never pair in the reference (dropped beforehand): filter them for the oracle
        def drop_nan(b):
            if len(b) == 3:
                k = np.isfinite(b[2])
                return (b[0][k], b[1][k], b[2][k])
            ka, kb = np.isfinite(b[2]), np.isfinite(b[5])
            return (b[0][ka], b[1][ka], b[2][ka], b[3][kb], b[4][kb], b[5][kb])

        got_e, got_c = ss.empirical_variogram_pairs(blocks, edges, est)
        ref_e, ref_c = vo.empirical_variogram_blocks([drop_nan(b) for b in blocks], edges, est)
        assert np.array_equal(got_c, ref_c), (trial, mode, kind, est)
        assert np.array_equal(np.isnan(got_e), np.isnan(ref_e)), (trial, mode, kind, est)
        ok = np.isfinite(ref_e)
        if est == "dowd":
            assert np.array_equal(got_e[ok], ref_e[ok]), (trial, mode, kind)
        else:
            assert np.allclose(got_e[ok], ref_e[ok], rtol=1e-12, atol=0), (trial, mode, kind, est)


def test_number_of_effective_samples(ss):
    """Callers of the fitted variogram: neff_exact / neff_hugonnet_approx (double covariance sums on the GPU) against the
    reference's vectorised NumPy expression restated here (xdem/spatialstats.py:2226-2231, 2297-2300), for every device
    model; closed-form vs numerical disk integration (the reference's own consistency test)."""
    import pandas as pd
    from scipy.spatial.distance import cdist

    rng = np.random.default_rng(12)
    n = 1500
    coords = rng.uniform(0, 2000, (n, 2))
    errors = rng.uniform(0.5, 2.0, n)
    for models in (pd.DataFrame({"model": ["spherical", "gaussian"], "range": [150.0, 900.0], "psill": [0.7, 0.3]}),
                   pd.DataFrame({"model": ["exponential"], "range": [400.0], "psill": [1.2]}),
                   pd.DataFrame({"model": ["cubic", "stable"], "range": [300.0, 1200.0], "psill": [0.5, 0.5], "smooth": [np.nan, 1.4]})):
        rho = ss.correlation_from_variogram(models)
        d = cdist(coords, coords)
        var = np.sum(errors.reshape((-1, 1)) @ errors.reshape((1, -1)) * rho(d.flatten()).reshape(d.shape))
        want = float(np.mean(errors)) ** 2 / (var / n**2)
        got = ss.neff_exact(coords, errors, models)
        assert got == pytest.approx(want, rel=1e-11)
        sub = np.random.default_rng(5).choice(n, size=200, replace=False)
        d2 = cdist(coords, coords[sub])
        var2 = np.sum(errors.reshape((-1, 1)) @ errors[sub].reshape((1, -1)) * rho(d2.flatten()).reshape(d2.shape))
        want2 = float(np.mean(errors)) ** 2 / (var2 / (n * 200))
        assert ss.neff_hugonnet_approx(coords, errors, models, subsample=200, random_state=5) == pytest.approx(want2, rel=1e-11)
    with pytest.raises(NotImplementedError, match="not available on the HIP engine"):
        ss.neff_exact(coords, errors, pd.DataFrame({"model": ["matern"], "range": [100.0], "psill": [1.0], "smooth": [0.5]}))
    m = pd.DataFrame({"model": ["spherical", "gaussian", "exponential", "cubic"], "range": [200.0, 500.0, 800.0, 1000.0],
                      "psill": [0.2, 0.3, 0.1, 0.4]})
    for area in (1e3, 1e5, 1e7):
        assert ss.neff_circular_approx_numerical(area, m) == pytest.approx(ss.neff_circular_approx_theoretical(area, m), rel=1e-6)
    # a regular grid of uncorrelated-scale points: about one effective sample per correlation patch
    big = np.stack(np.meshgrid(np.arange(300.0), np.arange(300.0)), -1).reshape(-1, 2) * 10.0
    ne = ss.neff_exact(big, np.ones(len(big)), m.iloc[:1])   # 90 000 points = 8.1e9 ordered pairs
    assert 100 < ne < len(big)


@pytest.mark.parametrize("gsd,x0", [(1.0, 0.0), (10.0, 502810.0), (0.5, -37.25), (2.5, 1e6), (30.0, 7.0)])
@pytest.mark.parametrize("estimator", ["matheron", "dowd", "cressie"])
def test_integer_lattice_kernels_equal_float64_kernels(ss, gsd, x0, estimator):
    """Raster-sampled points (coordinates x0 + gsd * integer, as the reference's samplers draw them, xdem/spatialstats.py:
    1413-1416) run the integer-lattice pair kernels: packed int16 indexes, one v_pk_sub_i16 + one v_dot2_i32_i16 per squared
    distance, integer thresholds that are the exact pre-images of the float64 ones.  Classes, counts and estimates must be
    IDENTICAL to the float64-coordinate kernels (option "vario_grid" = 0) and equal the oracle -- including pairs exactly on a
    class edge (edges = gsd * integers and gsd * sqrt(2)-geometric)."""
    from xdem_amd import _lib

    ctx = _lib.default_context()
    r = np.random.default_rng(17)

    def pts(n, extent):
        ix, iy = r.integers(0, extent, n), r.integers(0, extent, n)
        v = np.round(np.sin(ix / 40.0) + 0.3 * r.normal(size=n), 3).astype(np.float32)
        return x0 + gsd * ix, (x0 / 3) + gsd * iy, v

    blocks_c = [pts(300, 900) + pts(2500, 900), pts(257, 30000) + pts(4097, 30000), pts(1, 5) + pts(3, 5)]
    blocks_p = [pts(700, 600), pts(1025, 20000)]
    edges = sorted(set([gsd * e for e in (1.0, 2.0, 3.0, 5.0, 13.0, 25.0, 100.0)] + [gsd * np.sqrt(2) ** k for k in range(1, 30)]))
    for blocks in (blocks_c, blocks_p):
        got = {}
        for grid in (1, 0):
            ctx.set_option("vario_grid", grid)
            try:
                got[grid] = ss.empirical_variogram_pairs(blocks, edges, estimator, ctx)
            finally:
                ctx.set_option("vario_grid", 1)
        assert np.array_equal(got[1][1], got[0][1])
        if estimator == "dowd":
            assert np.array_equal(got[1][0], got[0][0], equal_nan=True)
        else:
            assert np.allclose(got[1][0], got[0][0], rtol=1e-13, atol=0, equal_nan=True)
        small = [b for b in blocks if b[0].size * (b[3].size if len(b) == 6 else b[0].size) < 3e6]
        e1, c1 = ss.empirical_variogram_pairs(small, edges, estimator, ctx)
        eo, co = vo.empirical_variogram_blocks(small, edges, estimator)
        assert np.array_equal(c1, co)
        ok = np.isfinite(eo)
        assert np.allclose(e1[ok], eo[ok], rtol=1e-12, atol=0)


@pytest.mark.parametrize("estimator", ["matheron", "cressie", "dowd"])
def test_morton_ordered_copy_equals_callers_order(ss, estimator):
    """Round 3: the sum passes run over a second device copy of the pair set whose points are in Morton order (run-length
    accumulation in registers).  Order must not matter: counts and Dowd medians identical, float64 sums to rounding, against
    the one-copy route (option "vario_sort" = 0) and the oracle -- on a spatially correlated field with NaN values, points that
    coincide, a block of a single point and pdist blocks, i.e. where runs of one lag class are long and where they are not."""
    from xdem_amd import _lib

    ctx = _lib.default_context()
    r = np.random.default_rng(29)

    def pts(n, extent, x0=0.0):
        ix, iy = r.integers(0, extent, n), r.integers(0, extent, n)
        v = (np.sin(ix / 25.0) * np.cos(iy / 31.0) + 0.05 * r.normal(size=n)).astype(np.float32)
        v[r.random(n) < 0.03] = np.nan
        return x0 + 5.0 * ix, 5.0 * iy, v

    dense = pts(3000, 40)                       # ~2 points per lattice cell: coincident points, zero distances
    blocks_c = [pts(700, 300) + pts(5000, 300), dense + dense, pts(1, 9) + pts(300, 9)]
    blocks_p = [pts(1500, 200), pts(513, 4000, x0=1e5)]
    edges = [float(e) for e in vo.default_bin_edges(5.0, 30000.0)]
    for blocks in (blocks_c, blocks_p):
        got = {}
        for srt in (1, 0):
            ctx.set_option("vario_sort", srt)
            try:
                got[srt] = ss.empirical_variogram_pairs(blocks, edges, estimator, ctx)
            finally:
                ctx.set_option("vario_sort", 1)
        assert np.array_equal(got[1][1], got[0][1])
        if estimator == "dowd":
            assert np.array_equal(got[1][0], got[0][0], equal_nan=True)
        else:
            assert np.allclose(got[1][0], got[0][0], rtol=1e-12, atol=0, equal_nan=True)
        # (the oracle takes finite values only: pairs with a NaN value are dropped by the kernels, cf. test_edge_inclusivity_and_nan)
        fin = []
        for b in blocks:
            f = []
            for k in range(0, len(b), 3):
                keep = np.isfinite(b[k + 2])
                f += [b[k][keep], b[k + 1][keep], b[k + 2][keep]]
            fin.append(tuple(f))
        eo, co = vo.empirical_variogram_blocks(fin, edges, estimator)
        assert np.array_equal(got[1][1], co)
        ok = np.isfinite(eo)
        assert np.allclose(got[1][0][ok], eo[ok], rtol=1e-12, atol=0)


@pytest.mark.parametrize("kind,dtype", [("pdist", np.float32), ("cdist", np.float64)])
def test_run_length_counting_pass_on_lattice_sets(ss, kind, dtype):
    """The run-length counting pass of the exact Dowd selection (round 4) outside the bench's configuration: all i < j pairs of
    one lattice point set (tiles at and below the diagonal keep the per-pair path) and float64 values, >= 4e9 pairs each.
    Medians and counts identical with run-length counting, per-pair counting on the sorted copy, the caller's order, and the
    plain digit passes; a spatially correlated field with ties."""
    from xdem_amd import _lib

    rng = np.random.default_rng(11)
    L = 20000
    f = lambda x, y: np.round(np.sin(x / 700.0) * np.cos(y / 900.0) + 0.05 * rng.normal(size=x.size), 3).astype(dtype)  # noqa: E731
    if kind == "pdist":
        n = 92000
        x, y = rng.integers(0, L, n).astype(np.float64), rng.integers(0, L, n).astype(np.float64)
        blocks = [(x, y, f(x, y))]
        total = n * (n - 1) // 2
    else:
        blocks = []
        for _ in range(3):
            na, nbp = 20000, 70000
            ax, ay = rng.integers(0, L, na).astype(np.float64), rng.integers(0, L, na).astype(np.float64)
            bx, by = rng.integers(0, L, nbp).astype(np.float64), rng.integers(0, L, nbp).astype(np.float64)
            blocks.append((ax, ay, f(ax, ay), bx, by, f(bx, by)))
        total = 3 * 20000 * 70000
    assert total >= 4_000_000_000
    edges = np.geomspace(np.sqrt(2), np.hypot(L, L), 40)
    ctx = _lib.default_context()
    ps = ss.PairSet(blocks, edges, ctx)
    res = {}
    try:
        assert ps.n_pairs == total
        s_m, c_m = ps.sums(0)
        res["run-length"] = ss.class_medians(ps)
        ctx.set_option("vario_runs", 0)
        res["per pair, sorted"] = ss.class_medians(ps)
        ctx.set_option("vario_runs", 1)
        ctx.check(ctx._L.xdemhip_pairs_link_sorted(ps.handle_sel, None))
        res["per pair, caller's order"] = ss.class_medians(ps)
        ctx.check(ctx._L.xdemhip_pairs_link_sorted(ps.handle_sel, ps.handle))
        ctx.set_option("selection", 1)
        res["plain"] = ss.class_medians(ps)
    finally:
        ctx.set_option("selection", 0)
        ctx.set_option("vario_runs", 1)
        ps.close()
    med0, cnt0 = res["plain"]
    assert np.array_equal(cnt0, c_m) and cnt0.sum() <= total
    for k, (med, cnt) in res.items():
        assert np.array_equal(cnt, cnt0), k
        assert np.array_equal(med, med0, equal_nan=True), k


@pytest.mark.parametrize("kind", ["cdist lattice", "pdist lattice", "cdist float64 coordinates"])
def test_float64_differences_of_float32_values_through_the_float32_shadow(ss, kind):
    """Option "vario_diff" = 1 (|dv| in float64: what SciPy's pdist does to float32 values) on float32 inputs, >= 4e9 pairs: the
    exact-median route runs its sampled passes and its counting pass on a float32 SHADOW of the set -- every pair classified by its
    float32 difference, rounding being monotone -- and stages / selects the candidates as exact float64 differences
    (xdemhip_pairs_link_shadow, WIDE kernels).  Medians and counts must be those of the float64 set's own route bit for bit (shadow
    unlinked: float64 kernels; plain digit passes) -- and differ from the float32-difference medians in the last digits (the
    conventions do differ).  Values with ties, a correlated field, differences that are no float32 numbers."""
    from xdem_amd import _lib

    rng = np.random.default_rng(23)
    L = 20000
    lattice = "lattice" in kind
    coord = (lambda n: rng.integers(0, L, n).astype(np.float64)) if lattice else (lambda n: rng.uniform(0, L, n))
    # (values around zero, all binades: float32 differences of such values round -- around 1000 m they would all be exact)
    f = lambda x, y: (40.0 * np.sin(x / 700.0) * np.cos(y / 900.0) + rng.normal(scale=1.5, size=x.size)).astype(np.float32)  # noqa: E731
    if kind.startswith("pdist"):
        n = 92000
        x, y = coord(n), coord(n)
        blocks = [(x, y, f(x, y))]
        total = n * (n - 1) // 2
    else:
        blocks = []
        for _ in range(3):
            na, nbp = 20000, 70000
            ax, ay, bx, by = coord(na), coord(na), coord(nbp), coord(nbp)
            blocks.append((ax, ay, f(ax, ay), bx, by, f(bx, by)))
        total = 3 * 20000 * 70000
    assert total >= 4_000_000_000
    edges = np.geomspace(np.sqrt(2), np.hypot(L, L), 40)
    ctx = _lib.default_context()
    res = {}
    try:
        ctx.set_option("vario_diff", 0)
        ps32 = ss.PairSet(blocks, edges, ctx)          # float32 differences
        res["float32 differences"] = ss.class_medians(ps32)
        ps32.close()
        ctx.set_option("vario_diff", 1)
        ps = ss.PairSet(blocks, edges, ctx)
        assert ps.shadow_sel is not None and ps.vdtype == np.float64
        try:
            res["shadow"] = ss.class_medians(ps)
            ctx.set_option("vario_runs", 0)
            res["shadow, per-pair counting"] = ss.class_medians(ps)
            ctx.set_option("vario_runs", 1)
            ctx.check(ctx._L.xdemhip_pairs_link_shadow(ps.handle_sel, None))
            res["float64 kernels"] = ss.class_medians(ps)
            ctx.set_option("selection", 1)
            res["float64 plain passes"] = ss.class_medians(ps)
            ctx.set_option("selection", 0)
            ctx.check(ctx._L.xdemhip_pairs_link_shadow(ps.handle_sel, ps.shadow_sel))
        finally:
            ps.close()
    finally:
        ctx.set_option("vario_diff", decided("vario_diff"))
        ctx.set_option("vario_runs", 1)
        ctx.set_option("selection", 0)
    med0, cnt0 = res["float64 plain passes"]
    for k in ("shadow", "shadow, per-pair counting", "float64 kernels"):
        med, cnt = res[k]
        assert np.array_equal(cnt, cnt0), k
        assert np.array_equal(med, med0, equal_nan=True), (k, np.nanmax(np.abs(med - med0)))
    m32, c32 = res["float32 differences"]
    assert np.array_equal(c32, cnt0)
    ok = np.isfinite(med0) & (cnt0 > 1000)
    assert np.all(np.abs(m32[ok] - med0[ok]) <= 2e-7 * np.abs(med0[ok])) and np.any(m32[ok] != med0[ok])


def test_link_sorted_refuses_a_different_pair_set(ss):
    """xdemhip_pairs_link_sorted: the companion must hold the same blocks (sizes, dtype, edges); anything else is refused and
    the set stays unlinked."""
    from xdem_amd import _lib

    rng = np.random.default_rng(5)
    mk = lambda na, nbp: (rng.uniform(0, 100, na), rng.uniform(0, 100, na), rng.normal(size=na).astype(np.float32),
                          rng.uniform(0, 100, nbp), rng.uniform(0, 100, nbp), rng.normal(size=nbp).astype(np.float32))
    edges = np.linspace(10, 150, 8)
    ctx = _lib.default_context()
    a, b, c = ss.PairSet([mk(300, 500)], edges, ctx), ss.PairSet([mk(300, 501)], edges, ctx), ss.PairSet([mk(300, 500)], edges * 1.5, ctx)
    try:
        for other in (b, c):
            assert ctx._L.xdemhip_pairs_link_sorted(a.handle_sel, other.handle_sel) == -1  # XDEMHIP_EINVAL
        assert ctx._L.xdemhip_pairs_link_sorted(a.handle_sel, a.handle) == 0
        med, cnt = ss.class_medians(a)
        assert cnt.sum() > 0
    finally:
        a.close(); b.close(); c.close()


def test_lattice_path_refuses_what_it_cannot_represent(ss):
    """Off-lattice points, lattices wider than 32767 cells and spacings whose squares are not exact fall back to the float64
    kernels (same results as with the lattice kernels disabled)."""
    from xdem_amd import _lib

    ctx = _lib.default_context()
    r = np.random.default_rng(5)
    v = r.normal(size=900).astype(np.float32)
    cases = [
        (r.uniform(0, 500, 900), r.uniform(0, 500, 900)),                       # not a lattice
        (r.integers(0, 40000, 900).astype(float), r.integers(0, 300, 900).astype(float)),  # too wide for int16
        (r.integers(0, 500, 900) * 0.1, r.integers(0, 500, 900) * 0.1),          # 0.1 is not a dyadic multiple: rounded coordinates
    ]
    edges = [float(e) for e in vo.default_bin_edges(0.1, 60000.0)]
    for x, y in cases:
        res = {}
        for grid in (1, 0):
            ctx.set_option("vario_grid", grid)
            try:
                res[grid] = ss.empirical_variogram_pairs([(x, y, v)], edges, "matheron", ctx)
            finally:
                ctx.set_option("vario_grid", 1)
        assert np.array_equal(res[1][1], res[0][1]) and np.allclose(res[1][0], res[0][0], rtol=1e-13, atol=0, equal_nan=True)
        eo, co = vo.empirical_variogram_blocks([(x, y, v)], edges, "matheron")
        assert np.array_equal(res[1][1], co)


def test_switchable_scikit_gstat_conventions(ss):
    """The two conventions of scikit-gstat that nothing readable offline pins are options, each checked against the oracle's
    implementation of the same choice: "vario_edge" (a distance exactly on an edge: [e_{k-1}, e_k) vs (e_{k-1}, e_k]) and
    "vario_diff" (|dv| in the value dtype vs float64)."""
    from xdem_amd import _lib

    ctx = _lib.default_context()
    r = np.random.default_rng(8)
    ix, iy = r.integers(0, 60, 500), r.integers(0, 60, 500)
    x, y = 2.0 * ix, 2.0 * iy
    v = (np.sin(ix / 9.0) * 1000 + r.normal(size=500)).astype(np.float32)
    edges = [2.0, 4.0, 10.0, 20.0, 26.0, 100.0, 200.0]      # 3-4-5 multiples: many pairs exactly on an edge
    for edge in (0, 1):
        for diff in (0, 1):
            ctx.set_option("vario_edge", edge)
            ctx.set_option("vario_diff", diff)
            try:
                for est in ("matheron", "dowd"):
                    e, c = ss.empirical_variogram_pairs([(x, y, v)], edges, est, ctx)
                    eo, co = vo.empirical_variogram_blocks([(x, y, v)], edges, est, right_closed=bool(edge), diff_f64=bool(diff))
                    assert np.array_equal(c, co), (edge, diff, est)
                    assert np.allclose(e, eo, rtol=1e-12, atol=0, equal_nan=True)
            finally:
                ctx.set_option("vario_edge", decided("vario_edge"))
                ctx.set_option("vario_diff", decided("vario_diff"))
    ctx.set_option("vario_edge", 0)
    e0, c0 = ss.empirical_variogram_pairs([(x, y, v)], edges, "matheron", ctx)
    ctx.set_option("vario_edge", 1)
    try:
        e1, c1 = ss.empirical_variogram_pairs([(x, y, v)], edges, "matheron", ctx)
    finally:
        ctx.set_option("vario_edge", decided("vario_edge"))
    assert not np.array_equal(c0, c1) and c0.sum() != 0   # the two conventions do differ on lattice data
