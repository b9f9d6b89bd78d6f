"""Pins the reference-owned host preparation of the variogram path (oracle/variogram_oracle.py) against vectors
recorded from the reference; the scikit-gstat part is 'parity unpinned' (see the oracle header)."""
import os

import numpy as np

import variogram_oracle as vo
from conftest import GOLDEN


def test_T7_sampling_parameter_table():
    z = np.load(os.path.join(GOLDEN, "vario_golden.npz"))
    n_ok = 0
    for subsample, nx, ny, gsd, runs, samples, ratio in z["T7|params"]:
        shape = (int(nx), int(ny))
        extent = (0.0, (shape[0] - 1) * gsd, 0.0, (shape[1] - 1) * gsd)
        if runs < 0:
            try:
                vo.choose_cdist_equidistant_sampling_parameters(int(subsample), extent, shape)
                raise AssertionError("expected ValueError")
            except ValueError:
                continue
        got = vo.choose_cdist_equidistant_sampling_parameters(int(subsample), extent, shape)
        assert got[0] == int(runs) and got[1] == int(samples)  # integer work: exact
        assert got[2] == ratio                                  # same float expression: bit-exact
        n_ok += 1
    assert n_ok >= 30
    # the survey's probe: N0 = 1e7 -> runs = 100, samples = 223607
    got = vo.choose_cdist_equidistant_sampling_parameters(10**7, (0, 19999, 0, 19999), (20000, 20000))
    assert got[:2] == (100, 223607)


def test_default_bin_edges_and_grid():
    coords, extent, maxlag = vo.grid_coords_extent_maxlag((30, 40), 2.0)
    assert coords.shape == (1200, 2) and extent == (0.0, 58.0, 0.0, 78.0)
    e = vo.default_bin_edges(2.0, maxlag)
    assert e[0] == np.sqrt(2) * 2.0 and e[-1] == maxlag and np.all(np.diff(e) > 0)
    assert np.allclose(np.array(e[1:-1]) / np.array(e[:-2]), np.sqrt(2))


def test_estimators_known_values():
    d = np.array([1.0, 2.0, 3.0, 4.0])
    assert vo._estimate(d, "matheron") == (1 + 4 + 9 + 16) / 8
    assert vo._estimate(d, "dowd") == 2.198 * 2.5**2 / 2
    assert np.isnan(vo._estimate(np.array([]), "matheron"))
    # lag classes are [e_{k-1}, e_k): a distance equal to an edge opens the next class
    args = (np.array([0.0]), np.array([0.0]), np.array([1.0, 2.0, 2.5, 5.0]), np.zeros(4), [1.0, 2.0, 5.0])
    assert vo.pair_groups(*args, right_closed=0).tolist() == [[1, 2, 2, -1]]
    # ... and (e_{k-1}, e_k] under the other convention a decision file may select (option "vario_edge" = 1)
    assert vo.pair_groups(*args, right_closed=1).tolist() == [[0, 1, 2, 2]]


def test_T7_masks_and_multi_range_subsamples():
    """The product's circular / ring masks (host logic of the "pdist_disk" / "pdist_ring" methods) against the masks recorded
    from the reference; the multi-range subsampling on top of them: range list, ring disjointness, sizes, seeding."""
    from xdem_amd import spatialstats as ss

    z = np.load(os.path.join(GOLDEN, "vario_golden.npz"))
    assert np.array_equal(ss._create_circular_mask((30, 41), center=(12, 20), radius=9.5), z["T7|circ"])
    assert np.array_equal(ss._create_ring_mask((30, 41), center=(12, 20), in_radius=4.0, out_radius=11.0), z["T7|ring"])
    shape, gsd = (200, 260), 2.0
    maxlag = float(np.hypot(199 * gsd, 259 * gsd))
    valid = np.ones(shape[0] * shape[1], dtype=bool)
    valid[::7] = False
    rings = ss._pdist_multi_range_subsamples(valid, shape, 500, "pdist_ring", gsd, maxlag, None, 11)
    disks = ss._pdist_multi_range_subsamples(valid, shape, 500, "pdist_disk", gsd, maxlag, None, 11)
    assert len(rings) == len(disks) == 6  # 20, 40, 80, 160, 320 (< maxlag / 2 = 326.6), maxlag
    rng = np.random.default_rng(11)
    cx, cy = rng.choice(shape[0], 1)[0], rng.choice(shape[1], 1)[0]  # one centre for all ranges of a seeded run
    rr, cc = np.unravel_index(np.arange(valid.size), shape)
    dist = np.sqrt((cc - cx) ** 2 + (rr - cy) ** 2)  # (the reference's axis convention)
    bounds = [0.0, 10.0, 20.0, 40.0, 80.0, 160.0, maxlag / gsd]
    for j, (r_, d_) in enumerate(zip(rings, disks)):
        assert valid[r_].all() and valid[d_].all() and len(set(r_.tolist())) == r_.size <= 500
        assert (dist[r_] >= bounds[j]).all() and (dist[r_] < bounds[j + 1]).all()
        assert (dist[d_] < bounds[j + 1]).all()
    assert ss._pdist_multi_range_subsamples(valid, shape, 500, "pdist_ring", gsd, maxlag, [30.0, 90.0], 11)[1].size == 500
