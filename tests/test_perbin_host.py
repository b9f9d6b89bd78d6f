"""Host half of xdem_amd.spatialstats.get_perbin_nd_binning on the CPU: the tables it hands to xdemhip_perbin_lookup -- sorted unique
intervals, the ends as NumPy compares them (float32 arrays against text-parsed ends), statistic / decision per bin of the product,
first row of a bin, the disjointness flag -- checked against the reference's own outputs (tests/golden/conv_golden.npz) with the
KERNEL emulated in NumPy by this test (masks over the handed-over tables: test code, not a fall-back of the product).  The real
kernel is tests/test_convolution_gpu.py's business."""
import ctypes
import io
import itertools
import threading

import numpy as np
import pandas as pd
import pytest

from test_oracle_conv_golden import ERRS, PERBIN_RUNS, Z, perbin_frame, perbin_vars


class _EmulatedLibrary:
    seen_disjoint = []

    def xdemhip_perbin_lookup(self, handle, ptrs, dts, n_var, n, nint, left, right, table, passb, disjoint, out, missing, memspace):
        counts = [nint[k] for k in range(n_var)]
        n_edges, n_bins = sum(counts), int(np.prod(counts))
        lo, hi = np.ctypeslib.as_array(left, (n_edges,)), np.ctypeslib.as_array(right, (n_edges,))
        tab = np.ctypeslib.as_array(table, (n_bins,))
        kind = np.frombuffer(ctypes.string_at(passb, n_bins), dtype=np.uint8)
        var = [np.ctypeslib.as_array(ctypes.cast(ptrs[k], ctypes.POINTER(ctypes.c_float if dts[k] == 0 else ctypes.c_double)), (n,)).astype(np.float64)
               for k in range(n_var)]
        res = np.ctypeslib.as_array(ctypes.cast(out, ctypes.POINTER(ctypes.c_double)), (n,))
        res[:] = np.nan
        off = np.cumsum([0] + counts)
        miss = 0
        for b, combo in enumerate(itertools.product(*[range(c) for c in counts])):
            m = np.ones(n, bool)
            for k, j in enumerate(combo):
                m &= (var[k] >= lo[off[k] + j]) & (var[k] < hi[off[k] + j])
            if kind[b] == 1:
                res[m] = tab[b]
            miss += int(m.sum()) if kind[b] == 2 else 0
        missing._obj.value = miss
        self.seen_disjoint.append(bool(disjoint))
        return 0


class _Ctx:
    _L = _EmulatedLibrary()
    handle = None
    call_lock = threading.RLock()

    def check(self, rc):
        assert rc == 0


@pytest.fixture()
def ss():
    from xdem_amd import spatialstats

    return spatialstats


@pytest.mark.parametrize("key", sorted(PERBIN_RUNS))
def test_tables_reproduce_the_reference(ss, key):
    d, vkeys, names, stat, mc = PERBIN_RUNS[key]
    got = ss.get_perbin_nd_binning(perbin_frame(d), perbin_vars(vkeys), names, statistic=stat, min_count=mc, ctx=_Ctx())
    assert np.array_equal(got, Z[f"perbin|{key}|out"], equal_nan=True)
    assert _Ctx._L.seen_disjoint[-1] is True


def test_text_intervals_overlaps_and_errors(ss):
    qs, qc = Z["perbin|q_slope"], Z["perbin|q_curv"]
    ctx = _Ctx()
    df_csv = pd.read_csv(io.StringIO(str(Z["perbin|csv_text"])))
    assert np.array_equal(ss.get_perbin_nd_binning(df_csv, [qs, qc], ["slope", "curv"], statistic="nmad", min_count=10, ctx=ctx), Z["perbin|csv|out"], equal_nan=True)
    df_csv1 = pd.read_csv(io.StringIO(str(Z["perbin|csv1_text"])))
    assert np.array_equal(ss.get_perbin_nd_binning(df_csv1, [qs], ["slope"], statistic="nmad", min_count=0, ctx=ctx), Z["perbin|csv1_f32|out"], equal_nan=True)
    hand = pd.DataFrame({"x": [pd.Interval(0.0, 5.0, closed="left"), pd.Interval(3.0, 8.0, closed="left"), pd.Interval(2.0, 4.0, closed="left")],
                         "count": [10.0, 1.0, 7.0], "val": [1.5, 2.5, 3.5]})
    x = Z["perbin|overlap|x"]
    assert np.array_equal(ss.get_perbin_nd_binning(hand, [x], ["x"], statistic="val", min_count=0, ctx=ctx), Z["perbin|overlap|out_min0"], equal_nan=True)
    assert _Ctx._L.seen_disjoint[-1] is False
    assert np.array_equal(ss.get_perbin_nd_binning(hand, [x], ["x"], statistic="val", min_count=5, ctx=ctx), Z["perbin|overlap|out_min5"], equal_nan=True)
    df1, df2, df3 = perbin_frame("1"), perbin_frame("2"), perbin_frame("3")
    calls = {
        "fewer_variables_than_the_binning": lambda: ss.get_perbin_nd_binning(df3, [qs], ["slope"], statistic="nmad", ctx=ctx),
        "lengths_differ": lambda: ss.get_perbin_nd_binning(df1, [qs, qc], ["slope"], statistic="nmad", ctx=ctx),
        "unknown_variable": lambda: ss.get_perbin_nd_binning(df1, [qs], ["aspect"], statistic="nmad", ctx=ctx),
        "unknown_statistic": lambda: ss.get_perbin_nd_binning(df1, [qs], ["slope"], statistic="mean", ctx=ctx),
        "no_count_column": lambda: ss.get_perbin_nd_binning(df1.drop(columns="count"), [qs], ["slope"], statistic="nmad", ctx=ctx),
        "empty_dataframe": lambda: ss.get_perbin_nd_binning(df1.iloc[:0], [qs], ["slope"], statistic="nmad", ctx=ctx),
        "min_count_none": lambda: ss.get_perbin_nd_binning(df1, [qs], ["slope"], statistic="nmad", min_count=None, ctx=ctx),
        "bin_without_row": lambda: ss.get_perbin_nd_binning(df2[df2.nd == 2].drop(index=df2[df2.nd == 2]["count"].idxmax()), [qs, qc],
                                                            ["slope", "curv"], statistic="nmad", ctx=ctx),
    }
    for label, fn in calls.items():
        with pytest.raises(Exception) as ei:
            fn()
        assert type(ei.value).__name__ == ERRS[label]["type"] and str(ei.value) == ERRS[label]["message"], (label, ei.value)


def test_interval_text_form(ss):
    """_pandas_str_to_interval (xdem/spatialstats.py:221-234): the four closures, a float cell, an interval pandas refuses."""
    f = ss._pandas_str_to_interval
    assert f("[0.5, 2.0)") == pd.Interval(0.5, 2.0, closed="left") and f("(0.5, 2.0]") == pd.Interval(0.5, 2.0, closed="right")
    assert f("[1, 3]") == pd.Interval(1.0, 3.0, closed="both") and f("(1, 3)") == pd.Interval(1.0, 3.0, closed="neither")
    assert np.isnan(f(np.nan)) and np.isnan(f("[3.0, 1.0)"))
