"""GPU parity of the two public functions added in round 6's last session, through the C-ABI:
* xdem_amd.spatialstats.convolution (csrc/convolve.hip) vs the reference's own outputs for BOTH of its engines
  (tests/golden/conv_golden.npz) and vs the pinned CPU oracle on larger images / filters -- BIT-EXACT;
* xdem_amd.spatialstats.get_perbin_nd_binning (csrc/perbin.hip) vs the reference's outputs on its own
  nd_binning DataFrames, their CSV round trip, overlapping hand-made intervals, and the errors upstream raises -- BIT-EXACT."""
import io

import numpy as np
import pandas as pd
import pytest

import conv_oracle as co
from test_oracle_conv_golden import CONV_CASES, ERRS, PERBIN_RUNS, Z, perbin_frame, perbin_vars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ss():
    from xdem_amd import spatialstats

    return spatialstats


@pytest.mark.parametrize("name", CONV_CASES)
def test_convolution_equals_the_reference_in_both_engines(ss, name):
    imgs, filters = Z[f"conv|{name}|imgs"], Z[f"conv|{name}|filters"]
    for method in ("scipy", "numba"):
        got = ss.convolution(imgs, filters, method=method)
        want = Z[f"conv|{name}|{method}"]
        assert got.dtype == np.float64 and got.shape == want.shape
        assert np.array_equal(got, want, equal_nan=True), (name, method)


def test_convolution_on_device_tensors_equals_the_host_form(ss):
    import torch

    imgs, filters = Z["conv|float32|5x5|imgs"], Z["conv|float32|5x5|filters"]
    t = torch.from_numpy(imgs).cuda()
    for method in ("scipy", "Numba"):
        out = ss.convolution(t, filters, method=method)
        torch.cuda.synchronize()
        assert out.is_cuda and out.dtype == torch.float64
        assert np.array_equal(out.cpu().numpy(), Z[f"conv|float32|5x5|{method.lower()}"], equal_nan=True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_convolution_larger_images_and_filters_vs_oracle(ss, dtype):
    """Several workgroups per side, ragged edges, a filter larger than the 64 x 16 block, and one whose LDS patch would pass 64 KiB
    (the global-memory loop): same bits as the oracle (itself pinned to the reference and to SciPy)."""
    rng = np.random.default_rng(11)
    imgs = (800.0 + np.cumsum(np.cumsum(rng.normal(scale=0.2, size=(2, 203, 331)), axis=1), axis=2)).astype(dtype)
    imgs[0, 50:53, 100] = np.nan
    imgs[1, 7, 7] = np.inf
    for shape in ((3, 3), (5, 5), (7, 7), (2, 2), (9, 4), (21, 35)):   # (3, 5, 7: the register-window kernels; the rest: runtime tap lists)
        filters = rng.normal(size=(3,) + shape)
        filters[1][rng.uniform(size=shape) < 0.5] = 0.0
        for method in ("scipy", "numba"):
            got = ss.convolution(imgs, filters, method=method)
            with np.errstate(invalid="ignore", over="ignore"):
                want = co.convolution(imgs, filters, method)
            assert np.array_equal(got, want, equal_nan=True), (shape, method)
    small = imgs[:1, :70, :90]
    big = np.zeros((1, 65, 67))           # sparse: the oracle's cost is per non-zero tap; the patch (80 x 130 values) leaves LDS for float64
    big[0, rng.integers(0, 65, 40), rng.integers(0, 67, 40)] = rng.normal(size=40)
    for method in ("scipy", "numba"):
        got = ss.convolution(small, big, method=method)
        with np.errstate(invalid="ignore", over="ignore"):
            want = co.convolution(small, big, method)
        assert np.array_equal(got, want, equal_nan=True), method


def test_convolution_reproduces_the_surface_fit_coefficients(ss):
    """The reference's own use (surfit.py:1107, tests/test_terrain/test_surfit.py:542-563): the Florinsky stencil tables divided by
    their resolution terms, convolved in one call -- equal to the oracle's coefficient planes of the terrain path."""
    import terrain_oracle as to

    rng = np.random.default_rng(5)
    dem = (1000 + np.cumsum(np.cumsum(rng.normal(scale=0.2, size=(90, 130)), 0), 1)).astype(np.float32)
    dem[30, 40] = np.nan
    names = ["zx", "zy", "zxx", "zyy", "zxy"]
    ks = to.conv_kernels("florinsky")
    filters = []
    for n in names:
        tab, (const, power) = ks[n]
        filters.append(tab.astype(np.float64) / (const * 10.0**power))
    got = ss.convolution(dem[None], np.stack(filters), method="scipy")
    want = to.surface_coefficients(dem, 10.0, "florinsky", names)
    for j, n in enumerate(names):
        assert np.array_equal(got[0, j], want[n], equal_nan=True), n


def test_convolution_refusals(ss):
    with pytest.raises(ValueError, match=ERRS["method_name"]["message"].replace('"', '.')):
        ss.convolution(np.zeros((1, 4, 4)), np.ones((1, 3, 3)), method="fft")
    with pytest.raises(TypeError, match="float32 or float64"):
        ss.convolution(np.zeros((1, 4, 4), dtype=np.int32), np.ones((1, 3, 3)))
    with pytest.raises(ValueError, match="three dimensions"):
        ss.convolution(np.zeros((4, 4)), np.ones((1, 3, 3)))
    assert ss.convolution(np.zeros((0, 4, 4), np.float32), np.ones((2, 3, 3))).shape == (0, 2, 4, 4)


@pytest.mark.parametrize("key", sorted(PERBIN_RUNS))
def test_perbin_lookup_equals_the_reference(ss, key):
    d, vkeys, names, stat, mc = PERBIN_RUNS[key]
    got = ss.get_perbin_nd_binning(perbin_frame(d), perbin_vars(vkeys), names, statistic=stat, min_count=mc)
    want = Z[f"perbin|{key}|out"]
    assert got.dtype == np.float64 and got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)


def test_perbin_lookup_callable_statistic_csv_round_trip_and_overlaps(ss):
    qs, qc = Z["perbin|q_slope"], Z["perbin|q_curv"]
    got = ss.get_perbin_nd_binning(perbin_frame("2"), [qs, qc], ["slope", "curv"], statistic=np.nanmedian, min_count=0)
    assert np.array_equal(got, Z["perbin|callable|out"], equal_nan=True)
    # intervals read back from text: Python floats, which NumPy compares with float32 arrays in float32
    df_csv = pd.read_csv(io.StringIO(str(Z["perbin|csv_text"])))
    got = ss.get_perbin_nd_binning(df_csv, [qs, qc], ["slope", "curv"], statistic="nmad", min_count=10)
    assert np.array_equal(got, Z["perbin|csv|out"], equal_nan=True)
    df_csv1 = pd.read_csv(io.StringIO(str(Z["perbin|csv1_text"])))
    got = ss.get_perbin_nd_binning(df_csv1, [qs], ["slope"], statistic="nmad", min_count=0)
    assert np.array_equal(got, Z["perbin|csv1_f32|out"], equal_nan=True)
    hand = pd.DataFrame({"x": [pd.Interval(0.0, 5.0, closed="left"), pd.Interval(3.0, 8.0, closed="left"), pd.Interval(2.0, 4.0, closed="left")],
                         "count": [10.0, 1.0, 7.0], "val": [1.5, 2.5, 3.5]})
    x = Z["perbin|overlap|x"]
    assert np.array_equal(ss.get_perbin_nd_binning(hand, [x], ["x"], statistic="val", min_count=0), Z["perbin|overlap|out_min0"], equal_nan=True)
    assert np.array_equal(ss.get_perbin_nd_binning(hand, [x], ["x"], statistic="val", min_count=5), Z["perbin|overlap|out_min5"], equal_nan=True)


def test_perbin_lookup_raster_sized_vs_oracle(ss):
    rng = np.random.default_rng(21)
    n = 30000
    a, b = rng.gamma(2.0, 8.0, n).astype(np.float32), np.abs(rng.normal(0, 1.5, n))
    v = (rng.normal(size=n) * (1 + 0.1 * a)).astype(np.float32)
    df = ss.nd_binning(v, [a, b], ["a", "b"], list_var_bins=(7, 5), statistics=["count", np.nanmedian])
    df = df[df.nd == 2]
    qa = rng.gamma(2.0, 9.0, (1500, 1700)).astype(np.float32)
    qb = np.abs(rng.normal(0, 1.7, (1500, 1700)))
    qa[::97, ::89] = np.nan
    got = ss.get_perbin_nd_binning(df, [qa, qb], ["a", "b"], statistic="nanmedian", min_count=20)
    want = co.get_perbin_nd_binning(df, [qa, qb], ["a", "b"], "nanmedian", 20)
    assert np.array_equal(got, want, equal_nan=True) and np.isfinite(got).mean() > 0.5


def test_perbin_lookup_errors_are_upstreams(ss):
    qs, qc = Z["perbin|q_slope"], Z["perbin|q_curv"]
    df1, df2, df3 = perbin_frame("1"), perbin_frame("2"), perbin_frame("3")
    calls = {
        "fewer_variables_than_the_binning": lambda: ss.get_perbin_nd_binning(df3, [qs], ["slope"], statistic="nmad"),
        "lengths_differ": lambda: ss.get_perbin_nd_binning(df1, [qs, qc], ["slope"], statistic="nmad"),
        "unknown_variable": lambda: ss.get_perbin_nd_binning(df1, [qs], ["aspect"], statistic="nmad"),
        "unknown_statistic": lambda: ss.get_perbin_nd_binning(df1, [qs], ["slope"], statistic="mean"),
        "no_count_column": lambda: ss.get_perbin_nd_binning(df1.drop(columns="count"), [qs], ["slope"], statistic="nmad"),
        "empty_dataframe": lambda: ss.get_perbin_nd_binning(df1.iloc[:0], [qs], ["slope"], statistic="nmad"),
        "min_count_none": lambda: ss.get_perbin_nd_binning(df1, [qs], ["slope"], statistic="nmad", min_count=None),
        "bin_without_row": lambda: ss.get_perbin_nd_binning(df2[df2.nd == 2].drop(index=df2[df2.nd == 2]["count"].idxmax()), [qs, qc], ["slope", "curv"],
                                                            statistic="nmad"),
    }
    for label, fn in calls.items():
        rec = ERRS[label]
        with pytest.raises(Exception) as ei:
            fn()
        assert type(ei.value).__name__ == rec["type"] and str(ei.value) == rec["message"], (label, ei.value)
    # upstream's unraised ValueError for cells that are neither intervals nor their text form
    with pytest.raises(ValueError, match="should be pandas.Interval"):
        ss.get_perbin_nd_binning(pd.DataFrame({"x": [1.0, 2.0], "count": [3.0, 4.0], "val": [0.1, 0.2]}), [np.arange(3.0)], ["x"], statistic="val")
