"""CPU oracle for the empirical-variogram path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What the reference does (xdem/spatialstats.py:1295-1546 ``sample_empirical_variogram``): host preparation in its
own code -- grid coordinates, maxlag, sqrt(2)-geometric right bin edges, subsampling parameters
(``_choose_cdist_equidistant_sampling_parameters`` 1104-1183) -- then the pairwise work is delegated to
``skgstat.Variogram`` / ``skgstat.RasterEquidistantMetricSpace`` (spatialstats.py:1091, 1247-1255) of the
THIRD-PARTY package scikit-gstat (``scikit-gstat>=1.0.18`` in setup.cfg:56; un-vendored, not installed here).

This oracle restates (own NumPy code):
  * the reference's own host preparation (pinned against golden vectors recorded from the reference:
    ``default_bin_edges`` and ``choose_cdist_equidistant_sampling_parameters``, tests/test_oracle_vario_golden.py);
  * scikit-gstat's published algorithm for the pairwise part, anchored on the reference's call sites:
      - pair distance = Euclidean distance of the coordinates (scipy pdist / cdist), pair value = |v_i - v_j|
        in the values' dtype (Variogram._calc_diff);
      - lag classes from explicit right edges e_0 < ... < e_{n-1} (``bin_func`` given as an iterable,
        spatialstats.py:1439-1449): class k holds e_{k-1} <= d < e_k with e_{-1} = 0; d >= e_{n-1} is dropped
        (Variogram._calc_groups);
      - estimators (skgstat.estimators): matheron = sum(d^2) / (2 n); cressie = 0.5 (mean sqrt d)^4 /
        (0.457 + 0.494/n + 0.045/n^2); dowd = 2.198 median(d)^2 / 2 (doc/source/robust_estimators.md:76-85 and the
        <= 1.0.0 halving fix at spatialstats.py:1529-1538); empty class -> NaN; ``bin_count`` = n.

PARITY UNPINNED for the scikit-gstat part: the package is absent, so the lag-edge inclusivity, the estimator
constants and the equidistant ring sampling cannot be checked against it offline; the reference's only pins
(tests/test_spatialstats.py:507-518, 543-630) need scikit-gstat and downloaded data.  The conventions above are
what both this oracle and the HIP kernel implement.
"""
from __future__ import annotations

import numpy as np

import _conventions

ESTIMATORS = ("matheron", "cressie", "dowd")


def default_bin_edges(gsd: float, maxlag: float) -> list[float]:
    """Right bin edges sqrt(2) gsd (sqrt 2)^k ... < maxlag, then maxlag (spatialstats.py:1439-1449)."""
    edges = []
    right = np.sqrt(2) * gsd
    while right < maxlag:
        edges.append(right)
        right *= np.sqrt(2)
    edges.append(maxlag)
    return edges


def grid_coords_extent_maxlag(shape: tuple[int, int], gsd: float):
    """coords (N,2), extent, maxlag exactly as spatialstats.py:1413-1431 builds them for a 2-D array."""
    x, y = np.meshgrid(np.arange(0, shape[0] * gsd, gsd), np.arange(0, shape[1] * gsd, gsd))
    coords = np.dstack((x.flatten(), y.flatten())).squeeze()
    extent = (np.min(coords[:, 0]), np.max(coords[:, 0]), np.min(coords[:, 1]), np.max(coords[:, 1]))
    maxlag = np.sqrt((extent[1] - extent[0]) ** 2 + (extent[3] - extent[2]) ** 2)
    return coords, extent, maxlag


def choose_cdist_equidistant_sampling_parameters(subsample: int, extent, shape, nb_rings: int = 10):
    """(runs, samples, ratio_subsample) of spatialstats.py:1104-1183."""
    min_subsample = np.ceil(np.sqrt(2 * nb_rings * 2**2) + 1)
    if subsample < min_subsample:
        raise ValueError(f"The number of subsamples needs to be at least {min_subsample:.0f}.")
    pairwise_comp_per_disk = np.ceil(subsample**2 / (2 * nb_rings))
    if pairwise_comp_per_disk < 10:
        runs = int(pairwise_comp_per_disk / 2**2)
    else:
        runs = int(min(100, 10 * np.ceil((pairwise_comp_per_disk / (2**2 * 10)) ** (1 / 3))))
    samples = int(np.ceil(np.sqrt(pairwise_comp_per_disk / runs)))
    maxdist = np.sqrt((extent[1] - extent[0]) ** 2 + (extent[3] - extent[2]) ** 2)
    res = np.mean([(extent[1] - extent[0]) / (shape[0] - 1), (extent[3] - extent[2]) / (shape[1] - 1)])
    ratio_subsample = res**2 * samples / (np.pi * maxdist**2 / np.sqrt(2) ** (2 * nb_rings))
    return runs, samples, ratio_subsample


def _estimate(diffs: np.ndarray, estimator: str) -> float:
    n = diffs.size
    if n == 0:
        return np.nan
    d = diffs.astype(np.float64)
    if estimator == "matheron":
        return (1.0 / (2 * n)) * np.sum(d**2)
    if estimator == "cressie":
        term1 = (1.0 / n) * np.sum(np.sqrt(d))
        term2 = 0.457 + (0.494 / n) + (0.045 / n**2)
        return 0.5 * term1**4 / term2
    if estimator == "dowd":
        return 2.198 * float(np.median(diffs)) ** 2 / 2
    raise ValueError(estimator)


def pair_groups(ax, ay, bx, by, edges, right_closed: bool | None = None):
    """Lag class of every pair (rows = a, cols = b): k with e_{k-1} <= d < e_k (``right_closed``: e_{k-1} < d <= e_k), or -1.
    (Which of the two scikit-gstat uses is unpinned offline: the product switches with the option "vario_edge".)"""
    if right_closed is None:   # the decided convention (oracle/_conventions.py)
        right_closed = bool(_conventions.decided("vario_edge"))
    d = np.sqrt((ax[:, None] - bx[None, :]) ** 2 + (ay[:, None] - by[None, :]) ** 2)
    g = np.searchsorted(np.asarray(edges, dtype=np.float64), d, side="left" if right_closed else "right")
    g[g >= len(edges)] = -1
    return g


def empirical_variogram_blocks(blocks, edges, estimator: str = "matheron", right_closed: bool | None = None, diff_f64: bool | None = None):
    """exp float64[n], count int64[n] over the union of pair blocks.

    ``blocks`` is a list of (ax, ay, av, bx, by, bv) -- every a paired with every b (cdist) -- or (ax, ay, av)
    -- all pairs i < j inside the set (pdist).  Values keep their dtype: |v_i - v_j| is formed in it, or in float64 with
    ``diff_f64`` (the product's option "vario_diff": SciPy's pdist / cdist widen first).
    """
    if right_closed is None:
        right_closed = bool(_conventions.decided("vario_edge"))
    if diff_f64 is None:
        diff_f64 = bool(_conventions.decided("vario_diff"))
    n = len(edges)
    per_bin: list[list[np.ndarray]] = [[] for _ in range(n)]
    for blk in blocks:
        if len(blk) == 3:
            ax, ay, av = blk
            av = av.astype(np.float64) if diff_f64 else av
            iu = np.triu_indices(ax.size, k=1)
            g = pair_groups(ax, ay, ax, ay, edges, right_closed)[iu]
            diff = np.abs(av[:, None] - av[None, :])[iu]
        else:
            ax, ay, av, bx, by, bv = blk
            if diff_f64:
                av, bv = av.astype(np.float64), bv.astype(np.float64)
            g = pair_groups(ax, ay, bx, by, edges, right_closed).ravel()
            diff = np.abs(av[:, None] - bv[None, :]).ravel()
        for k in range(n):
            sel = diff[g == k]
            if sel.size:
                per_bin[k].append(sel)
    exp = np.full(n, np.nan)
    count = np.zeros(n, dtype=np.int64)
    for k in range(n):
        if per_bin[k]:
            d = np.concatenate(per_bin[k])
            count[k] = d.size
            exp[k] = _estimate(d, estimator)
    return exp, count


def equidistant_blocks(coords: np.ndarray, values: np.ndarray, valid: np.ndarray, gsd: float, runs: int, samples: int,
                       ratio_subsample: float, rng: np.random.Generator, fac: float = np.sqrt(2)):
    """Pair blocks of the centre-disk / equidistant-ring scheme (Hugonnet et al. 2022, Suppl. Fig. 13; the design of
    skgstat.RasterEquidistantMetricSpace restated; PARITY UNPINNED against the package itself until
    oracle/pin_thirdparty.py has run somewhere it is importable): per run one random valid centre; a centre sample of
    up to `samples` valid points with d < r0 = sqrt(samples / (ratio_subsample pi)) gsd; an equidistant sample of up to
    `samples` from each of the rings [0, r0), [r0, r0 f), [r0 f, r0 f^2) ... the last one ending at the extent diagonal
    -- the first ring is the centre disk, sampled a second time; pairs = centre sample x union of ring samples.
    RNG protocol (shared with the product so that seeds reproduce): one ``choice`` for the centre, one
    ``choice(.., samples, replace=False)`` for the centre sample if over-full, then one per over-full ring, inner to outer."""
    x, y = coords[:, 0], coords[:, 1]
    r0 = np.sqrt(samples / (ratio_subsample * np.pi)) * gsd
    diag = np.hypot(x.max() - x.min(), y.max() - y.min())
    bounds = [0.0]
    nxt = r0
    while nxt < diag:
        bounds.append(nxt)
        nxt = nxt * fac
    bounds.append(diag)
    candidates = np.flatnonzero(valid)
    out = []
    for _ in range(runs):
        c = rng.choice(candidates)
        d = np.sqrt((x - x[c]) ** 2 + (y - y[c]) ** 2)
        centre = np.flatnonzero(valid & (d < r0))
        if centre.size > samples:
            centre = rng.choice(centre, samples, replace=False)
        picked = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            members = np.flatnonzero(valid & (d >= lo) & (d < hi))
            if members.size > samples:
                members = rng.choice(members, samples, replace=False)
            picked.append(members)
        b = np.concatenate(picked) if picked else np.empty(0, dtype=np.int64)
        if centre.size and b.size:
            out.append((x[centre], y[centre], values[centre], x[b], y[b], values[b]))
    return out
