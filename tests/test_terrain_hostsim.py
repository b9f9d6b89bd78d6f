"""CPU check of the GPU kernel's own math: xdem_amd/csrc/terrain_math.h (the per-column marcher and the float64
attribute formulas the HIP kernel instantiates) compiled for the host by tests/hostsim and compared with the oracle.
Guards the stencil algebra, window rotation, NaN poisoning and polynomial accuracy on machines without a GPU."""
import numpy as np
import pytest

import terrain_oracle as to
from hostsim_util import hostsim_terrain
from parity import assert_parity

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
        assert_parity(g, r, f"{fit}/{cm}/{a}", min_exact=0.999)


@pytest.mark.parametrize("fit,attrs", [("Florinsky", HOT11), ("ZevenbergThorne", HOT11),
                                       ("Horn", ["slope", "aspect", "hillshade", "topographic_position_index",
                                                 "terrain_ruggedness_index"])])
def test_specialised_kernel_math(fit, attrs):
    """The compile-time specialised instantiations (what bench.py runs)."""
    dem = _dem(seed=8)
    got = hostsim_terrain(dem, attrs, resolution=10.0, surface_fit=fit)
    ref = to.terrain_attributes(dem, attrs, resolution=10.0, surface_fit=fit)
    for a, g, r in zip(attrs, got, ref):
        assert_parity(g, r, f"{fit}/{a}", min_exact=0.9999)


def test_float64_and_tile_boundaries():
    dem = _dem((70, 530), seed=5, dtype=np.float64)  # three column tiles, three row tiles of 32
    got = hostsim_terrain(dem, HOT11, resolution=2.0)
    ref = to.terrain_attributes(dem, HOT11, resolution=2.0)
    for a, g, r in zip(HOT11, got, ref):
        assert_parity(g, r, a)


def test_exact_zero_on_planar_terrain():
    """Where the reference only returns rounding noise (|zx| ~ 1e-15 from non-cancelling weights) the kernel is exact."""
    ramp = np.add.outer(np.arange(40.0), 2 * np.arange(45.0)).astype(np.float32)
    flat = np.full((30, 31), 7.0, np.float32)
    for fit in ("Florinsky", "ZevenbergThorne"):
        got = hostsim_terrain(ramp, HOT11, resolution=2.0, surface_fit=fit)
        for a, g in zip(HOT11, got):
            if "curvature" in a or a == "topographic_position_index":
                assert np.nanmax(np.abs(g)) == 0.0, (fit, a)
        s, asp = hostsim_terrain(flat, ["slope", "aspect"], resolution=2.0, surface_fit=fit)
        assert np.nanmax(s) == 0.0 and np.nanmax(asp) == 0.0
