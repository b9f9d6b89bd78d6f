"""C-ABI threading contract ("ctypes calls release the GIL; library must be re-entrant per context handle"): two
contexts on one GPU, driven from two Python threads at the same time with interleaved terrain, Nuth-Kaab and variogram calls,
must give exactly the results of the same calls made serially.  (The reference's own parallelism is process-based --
mp.Pool across variogram runs, xdem/spatialstats.py:1502 -- so a thread per context is the analogous unit here.)"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATTRS = ["slope", "aspect", "hillshade", "profile_curvature", "max_curvature", "topographic_position_index",
         "terrain_ruggedness_index"]


def _work(ctx, seed, out, rounds=6):
    from xdem_amd import _lib

    with _lib.use_context(ctx):
        _work_in_context(ctx, seed, out, rounds)


def _work_in_context(ctx, seed, out, rounds):
    from xdem_amd import coreg
    from xdem_amd import spatialstats as ss
    from xdem_amd import terrain
    from xdem_amd.synth import fbm_numpy

    rng = np.random.default_rng(seed)
    res = []
    for r in range(rounds):
        dem = fbm_numpy((300 + 17 * r, 400 + 29 * seed), seed=seed * 100 + r)
        dem[5 + r, 7] = np.nan
        res.append(("terrain", [a.copy() for a in terrain.get_terrain_attribute(dem, ATTRS, resolution=5.0)]))
        x, y = rng.integers(0, 400, 1500).astype(float), rng.integers(0, 400, 1500).astype(float)
        v = rng.normal(size=1500).astype(np.float32)
        edges = [2.0 * 1.5**k for k in range(14)]
        res.append(("dowd", ss.empirical_variogram_pairs([(x, y, v)], edges, "dowd", ctx)))
        res.append(("matheron_counts", ss.empirical_variogram_pairs([(x, y, v)], edges, "matheron", ctx)[1]))
        tba = (np.roll(dem, (1, -1), (0, 1)) + 1.0).astype(np.float32)
        plan = coreg.NKPlan(dem, tba, None, ctx)
        d = plan.step(3.0, -2.0, (5.0, 5.0), 72)
        plan.close()
        res.append(("nk", (d["vshift"], d["n_valid"], d["counts"].copy(), d["medians"].copy())))
    out[seed] = res


def _same(a, b):
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return np.array_equal(a, b, equal_nan=True)
    return a == b or (a != a and b != b)


def test_two_contexts_two_threads_equal_serial():
    from xdem_amd import _lib

    if not hasattr(_lib.lib().xdemhip_terrain, "argtypes"):
        pytest.skip("binding not loaded")
    ctx_a, ctx_b = _lib.Context(0), _lib.Context(0)
    try:
        serial, threaded = {}, {}
        _work(ctx_a, 1, serial)
        _work(ctx_b, 2, serial)
        errors = []

        def run(ctx, seed):
            try:
                _work(ctx, seed, threaded)
            except Exception as e:  # pragma: no cover
                errors.append(repr(e))

        for _ in range(2):  # twice: contexts keep state (events, scratch) between calls
            threaded.clear()
            ts = [threading.Thread(target=run, args=(ctx_a, 1)), threading.Thread(target=run, args=(ctx_b, 2))]
            for t in ts:
                t.start()
            for t in ts:
                t.join(timeout=300)
            assert not errors, errors
            assert all(not t.is_alive() for t in ts)
            for seed in (1, 2):
                for (ka, va), (kb, vb) in zip(serial[seed], threaded[seed]):
                    assert ka == kb and _same(va, vb), (seed, ka)
    finally:
        ctx_a.close()
        ctx_b.close()


def test_context_takes_decided_conventions_from_the_decision_file(tmp_path, monkeypatch):
    """A decision file with NON-default conventions (what oracle/pin_thirdparty.py writes once the third-party packages can be
    run) must configure every new context: options mirrored in `ctx.options`, and the behaviour really switched -- a pair
    whose distance falls exactly on a class edge moves to the other lag class under vario_edge = 1."""
    import json

    from xdem_amd import _lib
    from xdem_amd import spatialstats as ss

    path = tmp_path / "thirdparty_decision.json"
    path.write_text(json.dumps({"nk_nan_rule": 3, "vario_edge": 1, "_source": "test"}))
    monkeypatch.setenv("XDEM_THIRDPARTY_DECISION", str(path))
    assert _lib.thirdparty_decision() == {"nk_nan_rule": 3, "vario_edge": 1}
    ctx = _lib.Context(0)
    try:
        assert ctx.options == {"nk_nan_rule": 3, "vario_edge": 1}
        x = np.array([0.0, 2.0, 5.0])
        y = np.zeros(3)
        v = np.array([0.0, 1.0, 3.0], dtype=np.float32)
        edges = [2.0, 4.0, 8.0]   # distances 2, 3, 5: the first lies on an edge
        with _lib.use_context(ctx):
            _, count_right_closed = ss.empirical_variogram_pairs([(x, y, v)], edges, "matheron")
        monkeypatch.delenv("XDEM_THIRDPARTY_DECISION")
        plain = _lib.Context(0)
        try:
            assert plain.options == {}
            with _lib.use_context(plain):
                _, count_default = ss.empirical_variogram_pairs([(x, y, v)], edges, "matheron")
        finally:
            plain.close()
        assert count_right_closed.tolist() == [1, 1, 1] and count_default.tolist() == [0, 2, 1]
    finally:
        ctx.close()
