"""GPU parity tests of the patches method: the HIP mean filter against the reference's own outputs
(tests/golden/patches_golden.npz) and the CPU oracle -- bit-exact (float64 sums in SciPy's order, integer counts)."""
import os

import numpy as np
import pytest

import patches_oracle as po

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "patches_golden.npz"))


@pytest.fixture(scope="module")
def ss():
    from xdem_amd import spatialstats as s

    return s


def test_mean_filter_nan_reference_fixtures(ss):
    n = 0
    for key in Z.files:
        if not key.startswith("mean|"):
            continue
        _, name, shape, p = key.split("|")
        mean, valid, npx = ss.mean_filter_nan(Z[f"img|{name}"], int(p), shape)
        assert npx == int(Z[f"npx|{name}|{shape}|{p}"]), key
        assert mean.dtype == np.float64 and valid.dtype == np.float64
        assert np.array_equal(valid, Z[f"valid|{name}|{shape}|{p}"]), key
        assert np.array_equal(mean, Z[key], equal_nan=True), key
        n += 1
    assert n >= 60


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mean_filter_nan_larger_vs_oracle(ss, dtype):
    """Rasters spanning many 64 x 16 blocks (ragged edges), every kernel the int8 count allows, holes of all sizes."""
    rng = np.random.default_rng(3)
    img = (500.0 + np.cumsum(rng.normal(size=(203, 331)), axis=1)).astype(dtype)
    img[rng.uniform(size=img.shape) < 0.1] = np.nan
    img[60:90, 100:180] = np.nan
    img[5, 7] = np.inf
    for shape, sizes in (("square", (1, 2, 7, 10, 11)), ("circular", (2, 3, 7, 10, 12, 13))):
        for p in sizes:
            mean, valid, npx = ss.mean_filter_nan(img, p, shape)
            m0, v0, n0 = po.mean_filter_nan(img, p, shape)
            assert npx == n0 and np.array_equal(valid, v0) and np.array_equal(mean, m0, equal_nan=True), (shape, p)
    for shape, p in (("square", 12), ("circular", 14), ("square", 40)):
        with pytest.raises(NotImplementedError, match="int8"):
            ss.mean_filter_nan(img, p, shape)
    with pytest.raises(ValueError, match="Kernel shape"):
        ss.mean_filter_nan(img, 3, "hexagon")


def test_patches_method_forms_vs_reference(ss):
    vals = Z["patches|values"]
    for key in Z.files:
        if key.startswith("pconv|"):
            _, shape, area = key.split("|")
            stat, nb, exact, df = ss._patches_convolution(vals, 2.0, float(area), 80.0, shape, statistic_between_patches=po.nmad,
                                                          return_in_patch_statistics=True)
            assert np.array_equal(np.array([stat, nb, exact]), Z[key]), key
            assert np.array_equal(np.stack([df["nanmean"].values, df["count"].values]), Z[f"pconv_df|{shape}|{area}"], equal_nan=True)
        if key.startswith("pquad|"):
            _, shape, area, seed = key.split("|")
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                stat, nb, exact, df = ss._patches_loop_quadrants(vals, 2.0, float(area), shape, 12, 80.0, statistic_between_patches=po.nmad,
                                                                 random_state=int(seed), return_in_patch_statistics=True)
            assert np.array_equal(np.array([stat, nb, exact], dtype=np.float64), Z[key], equal_nan=True), key
            tiles = list(Z[f"pquad_tiles|{shape}|{area}|{seed}"])
            if tiles:
                assert list(df["tile"].values) == tiles
                assert np.array_equal(np.stack([df["nanmean"].values.astype(np.float64), df["count"].values.astype(np.float64)]),
                                      Z[f"pquad_df|{shape}|{area}|{seed}"])


def test_patches_method_public_entry(ss):
    """The reference's own data-free expectations (tests/test_spatialstats.py:1344-1404): summary columns, one row per area,
    exact areas within 20 % of the requested ones, masks honoured; both the vectorized and the quadrant form."""
    rng = np.random.default_rng(11)
    vals = rng.normal(0, 2, (300, 360)).astype(np.float32)
    unstable = np.zeros(vals.shape, dtype=bool)
    unstable[100:180, 50:200] = True
    gsd = 20.0   # (the reference's test runs on a 20 m DEM: areas of 10000 / 20000 m^2 = circular kernels of 6 / 8 pixels across)
    df = ss.patches_method(vals, gsd=gsd, areas=[10000, 20000], unstable_mask=unstable, vectorized=True, convolution_method="scipy")
    assert df.shape == (2, 4) and list(df.columns) == ["nmad", "nb_indep_patches", "exact_areas", "areas"]
    assert df["exact_areas"][0] == pytest.approx(df["areas"][0], rel=0.2)
    assert df["nmad"][0] > df["nmad"][1] > 0            # the standard error of the mean falls with the patch area
    npx = int(po.kernel_of(6, "circular").sum())
    assert df["exact_areas"][0] == npx * gsd**2 and abs(df["nmad"][0] - 2.0 / np.sqrt(npx)) < 0.1   # white noise: sigma / sqrt(pixels)
    df2, full = ss.patches_method(vals, gsd=gsd, areas=[10000], unstable_mask=unstable, vectorized=False, n_patches=7, random_state=42,
                                  return_in_patch_statistics=True)
    assert df2.shape == (1, 4) and df2["nb_indep_patches"][0] == 7 and full.shape == (7, 5)
    assert all(full["count"].values > 0.8 * np.max(full["count"].values))
    with pytest.raises(NotImplementedError, match="int8"):
        ss.patches_method(vals, gsd=10.0, areas=[20000])   # a 16-pixel circle: 197 pixels, beyond the reference's int8 count
    # masked terrain never enters a patch mean: filling it with garbage changes nothing
    vals2 = vals.copy()
    vals2[unstable] = 1e6
    dfb = ss.patches_method(vals2, gsd=gsd, areas=[10000, 20000], unstable_mask=unstable)
    assert np.array_equal(dfb.values, df.values)
