"""CPU oracle of the NaN-ignoring mean filter and the patches method (TEST INFRASTRUCTURE ONLY -- never imported by xdem_amd).

Restates ``mean_filter_nan`` (/root/reference/xdem/spatialstats.py:2597-2655), ``_patches_convolution`` (2658-2741) and
``_patches_loop_quadrants`` (2744-2879) in plain NumPy.  PINNED: tests/test_oracle_patches_golden.py reproduces every array
of tests/golden/patches_golden.npz -- outputs of the reference itself (oracle/gen_golden_patches.py) -- bit for bit.

What the reference's two ``scipy.ndimage.convolve(..., mode="constant", cval=nan)`` calls amount to (2626-2644, 2512-2525):
* kernel k = ones((p, p)) or the p x p circular mask of ``_create_circular_mask`` (centre (p // 2, p // 2), radius p // 2,
  strict <), uint8; a true CONVOLUTION: tap (a, b) of the kernel reads image pixel (r + p // 2 - a, c + p // 2 - b), i.e.
  the window spans offsets -(p - 1 - p // 2) .. p // 2 (one more pixel towards larger indexes for even p); zero weights
  are skipped (so the circle's corners never touch the border);
* sum image: non-finite pixels count as 0; accumulated in float64 over the window in row-major IMAGE order (increasing
  row offset, then increasing column offset: SciPy walks the flipped kernel), rounded to the image dtype, stored as float64;
  a window with a non-zero tap outside the raster sums the NaN border value -> NaN;
* count image: int8 ones (0 where non-finite) convolved into an int8 result: the same NaN border value makes the float64
  accumulator NaN, whose cast to int8 is 0 -- border windows count 0; beyond 127 kernel pixels the int8 result wraps (e.g.
  -112 for a 12 x 12 square), which is why the product refuses such kernels;
* mean = sum / count in float64 (0 / 0 -> NaN).
"""
from __future__ import annotations

import numpy as np


def circular_mask(shape, center=None, radius=None):
    """``_create_circular_mask`` (spatialstats.py:880-904), its axis convention included."""
    w, h = shape
    if center is None:
        center = (int(w / 2), int(h / 2))
    if radius is None:
        radius = min(center[0], center[1], w - center[0], h - center[1])
    Y, X = np.ogrid[:w, :h]
    return np.sqrt((X - center[0]) ** 2 + (Y - center[1]) ** 2) < radius


def kernel_of(p: int, shape: str) -> np.ndarray:
    if shape.lower() == "square":
        return np.ones((p, p), dtype=np.uint8)
    if shape.lower() == "circular":
        return circular_mask((p, p)).astype(np.uint8)
    raise ValueError('Kernel shape should be "square" or "circular".')


def window_offsets(kernel: np.ndarray):
    """(row offset, column offset) of every non-zero tap, in the order SciPy accumulates them."""
    p = kernel.shape[0]
    taps = [(p // 2 - a, p // 2 - b) for a in range(p) for b in range(p) if kernel[a, b]]
    return sorted(taps)


def mean_filter_nan(img: np.ndarray, kernel_size: int, kernel_shape: str = "circular"):
    kernel = kernel_of(kernel_size, kernel_shape)
    taps = window_offsets(kernel)
    H, W = img.shape
    fin = np.isfinite(img)
    zeroed = np.where(fin, img, 0).astype(np.float64)
    pad = kernel_size
    zp = np.full((H + 2 * pad, W + 2 * pad), np.nan)
    zp[pad:pad + H, pad:pad + W] = zeroed
    fp = np.full((H + 2 * pad, W + 2 * pad), np.nan)
    fp[pad:pad + H, pad:pad + W] = fin
    s = np.zeros((H, W))
    n = np.zeros((H, W))
    for dy, dx in taps:   # sequential float64 accumulation in SciPy's order
        s = s + zp[pad + dy:pad + dy + H, pad + dx:pad + dx + W]
        n = n + fp[pad + dy:pad + dy + H, pad + dx:pad + dx + W]
    summed = s.astype(img.dtype).astype(np.float64)
    with np.errstate(invalid="ignore"):
        nb_valid = np.where(np.isnan(n), 0.0, n)   # int8(NaN) = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = summed / nb_valid
    return mean, nb_valid, int(np.count_nonzero(kernel))


def nmad(data, nfact: float = 1.4826):
    arr = np.asarray(data)
    return nfact * np.nanmedian(np.abs(arr - np.nanmedian(arr)))


def patches_convolution(values, gsd, area, perc_min_valid=80.0, patch_shape="circular", statistic_between_patches=nmad):
    """``_patches_convolution`` (spatialstats.py:2658-2741) -> (statistic, number of patches, exact area, (nanmean, count) of the
    first independent subset)."""
    if patch_shape.lower() == "circular":
        k = int(np.round(2 * np.sqrt(area / np.pi) / gsd, decimals=0))
    elif patch_shape.lower() == "square":
        k = int(np.round(np.sqrt(area) / gsd, decimals=0))
    else:
        raise ValueError('Kernel shape should be "square" or "circular".')
    mean_img, nb_valid, nb_px = mean_filter_nan(values, k, patch_shape)
    mean_img[nb_valid < nb_px * perc_min_valid / 100.0] = np.nan
    stats, nbs = [], []
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for i in range(k):
            for j in range(k):
                sub = mean_img[i::k, j::k]
                stats.append(statistic_between_patches(sub.ravel()))
                nbs.append(np.count_nonzero(np.isfinite(sub)))
        return (float(np.nanmean(np.asarray(stats))), float(np.nanmean(np.asarray(nbs))), nb_px * gsd**2,
                np.stack([mean_img[::k, ::k].ravel(), nb_valid[::k, ::k].ravel()]))


def patches_loop_quadrants(values, gsd, area, patch_shape="circular", n_patches=1000, perc_min_valid=80.0,
                           statistic_between_patches=nmad, random_state=None):
    """``_patches_loop_quadrants`` (spatialstats.py:2744-2879) with np.nanmean as the in-patch statistic -> (statistic, number of
    patches, exact area, tile names, (nanmean, count) per patch).  Quirks kept: for square patches "the exact number of pixels"
    is nx_sub * ny_sub (the number of QUADRANTS), so no square patch ever qualifies unless k^2 happens to equal it."""
    rng = np.random.default_rng(random_state)
    nx, ny = values.shape
    k = int(np.round(np.sqrt(area) / gsd, decimals=0))
    nx_sub, ny_sub = int(np.floor((nx - 1) / k)), int(np.floor((ny - 1) / k))
    rad = int(np.round(np.sqrt(area / np.pi) / gsd, decimals=0))
    if patch_shape.lower() == "square":
        nb_exact = nx_sub * ny_sub
    elif patch_shape.lower() == "circular":
        nb_exact = np.count_nonzero(circular_mask((nx, ny), radius=rad))
    else:
        raise ValueError("Patch method must be square or circular.")
    exact_area = nb_exact * gsd**2
    quads = [[i, j] for i in range(nx_sub) for j in range(ny_sub)]
    u, remaining = 0, n_patches
    tiles, means, counts = [], [], []
    while len(quads) > 0 and u < n_patches:
        idxs = rng.choice(len(quads), size=min(len(quads), 10 * remaining))
        for iq in idxs:
            i, j = quads[iq]
            if patch_shape.lower() == "square":
                patch = values[k * i:k * (i + 1), k * j:k * (j + 1)].flatten()
            else:
                cx, cy = np.floor(k * (i + 1 / 2)), np.floor(k * (j + 1 / 2))
                patch = values[circular_mask((nx, ny), center=(cx, cy), radius=rad)]
            nb_total, nb_valid = len(patch), int(np.count_nonzero(np.isfinite(patch)))
            if nb_valid >= np.ceil(perc_min_valid / 100.0 * nb_total) and nb_total == nb_exact:
                u += 1
                if u > n_patches:
                    break
                tiles.append(f"{i}_{j}")
                means.append(np.nanmean(patch[np.isfinite(patch)].astype("float64")))
                counts.append(nb_valid)
        remaining = n_patches - u
        drop = set(int(q) for q in idxs)
        quads = [c for q, c in enumerate(quads) if q not in drop]
    if tiles:
        m = np.asarray(means)
        return float(statistic_between_patches(m)), int(np.count_nonzero(np.isfinite(m))), exact_area, tiles, np.stack([m, np.asarray(counts, dtype=np.float64)])
    return float("nan"), 0, exact_area, [], np.array([[np.nan], [np.nan]])
