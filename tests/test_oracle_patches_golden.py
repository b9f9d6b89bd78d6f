"""oracle/patches_oracle.py against the outputs of the reference itself (tests/golden/patches_golden.npz, written by
oracle/gen_golden_patches.py from /root/reference): the mean filter bit for bit -- means, valid counts, kernel pixel counts,
every kernel size / shape incl. even sizes and the circular mask -- and both forms of the patches method."""
import os

import numpy as np
import pytest

import patches_oracle as po

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "patches_golden.npz"))


def test_mean_filter_nan_bit_exact():
    n = 0
    for key in Z.files:
        if not key.startswith("mean|"):
            continue
        _, name, shape, p = key.split("|")
        img = Z[f"img|{name}"]
        mean, valid, npx = po.mean_filter_nan(img, int(p), shape)
        assert npx == int(Z[f"npx|{name}|{shape}|{p}"])
        assert np.array_equal(valid, Z[f"valid|{name}|{shape}|{p}"]), key
        assert np.array_equal(mean, Z[key], equal_nan=True), key
        n += 1
    assert n >= 60
    # the reference's int8 count wraps beyond 127 kernel pixels (why the product refuses such kernels)
    assert int(Z["wrap|npx"]) == 144 and float(Z["wrap|valid"][10, 10]) == -112.0


def test_patches_method_forms():
    vals = Z["patches|values"]
    for key in Z.files:
        if key.startswith("pconv|"):
            _, shape, area = key.split("|")
            stat, nb, exact, df = po.patches_convolution(vals, 2.0, float(area), 80.0, shape)
            assert np.array_equal(np.array([stat, nb, exact]), Z[key]), key
            assert np.array_equal(df, Z[f"pconv_df|{shape}|{area}"], equal_nan=True)
        if key.startswith("pquad|"):
            _, shape, area, seed = key.split("|")
            stat, nb, exact, tiles, df = po.patches_loop_quadrants(vals, 2.0, float(area), shape, 12, 80.0, random_state=int(seed))
            assert np.array_equal(np.array([stat, nb, exact]), Z[key], equal_nan=True), key
            assert list(Z[f"pquad_tiles|{shape}|{area}|{seed}"]) == tiles
            if tiles:
                assert np.array_equal(df, Z[f"pquad_df|{shape}|{area}|{seed}"])
