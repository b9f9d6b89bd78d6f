"""Synthetic inputs for tests and benchmarks: fractional-Brownian DEMs by spectral synthesis."""
from __future__ import annotations

import math

import numpy as np


def fbm_numpy(shape, hurst: float = 0.7, seed: int = 42, mean: float = 1000.0, std: float = 300.0,
              dtype=np.float32) -> np.ndarray:
    """Periodic fBm surface: white noise x |k|^-(H+1) in the Fourier domain, rescaled to (mean, std)."""
    n, m = shape
    rng = np.random.default_rng(seed)
    f = np.fft.rfft2(rng.normal(size=(n, m)))
    ky = np.fft.fftfreq(n)[:, None]
    kx = np.fft.rfftfreq(m)[None, :]
    k = np.sqrt(kx**2 + ky**2)
    k[0, 0] = 1.0
    z = np.fft.irfft2(f * k ** (-(hurst + 1.0)), s=(n, m))
    z = (z - z.mean()) / z.std()
    return (mean + std * z).astype(dtype)


def fbm_torch(H: int, W: int, device, hurst: float = 0.7, seed: int = 42, mean: float = 1000.0, std: float = 300.0,
              tile: int = 8192, row0: int = 0, total_rows: int | None = None):
    """float32 fBm DEM of any size built on the device: one periodic `tile`^2 fBm patch repeated over the raster
    (continuous across repeats) plus a smooth large-scale trend so that repeats differ.  `row0`/`total_rows`
    let a rank generate only its row block of a larger raster (identical values to the full build)."""
    import torch

    total_rows = total_rows or H
    t = min(tile, 1 << max(1, math.ceil(math.log2(max(total_rows, W)))))
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    w = torch.randn((t, t), generator=g, device=device, dtype=torch.float32)
    f = torch.fft.rfft2(w)
    ky = torch.fft.fftfreq(t, device=device)[:, None]
    kx = torch.fft.rfftfreq(t, device=device)[None, :]
    k = torch.sqrt(kx * kx + ky * ky)
    k[0, 0] = 1.0
    f = f * k.pow(-(hurst + 1.0))
    z = torch.fft.irfft2(f, s=(t, t))
    del f, w, k
    z = (z - z.mean()) / z.std()
    out = torch.empty((H, W), device=device, dtype=torch.float32)
    two_pi = 2.0 * math.pi
    xs = torch.arange(W, device=device, dtype=torch.float32)
    trend_x = 50.0 * torch.sin(two_pi * xs / W)
    for y0 in range(0, H, t):
        y1 = min(H, y0 + t)
        gy = torch.arange(row0 + y0, row0 + y1, device=device)
        ys = gy.to(torch.float32)
        trend_y = 50.0 * torch.cos(two_pi * ys / total_rows)
        src_rows = z[gy % t]
        for x0 in range(0, W, t):
            x1 = min(W, x0 + t)
            out[y0:y1, x0:x1] = mean + std * src_rows[:, : x1 - x0] + trend_x[None, x0:x1] + trend_y[:, None]
    return out


def c5_variogram_blocks(device, runs: int = 100, samples: int = 9091, size: int = 20000, seed: int = 45, rings: int = 10):
    """BASELINE C5 input as SURVEY.md 8d states it: a dh-like fBm(H = 0.3) field on a `size`^2 grid (gsd 1, no NaN, seed 45) and,
    per run, the centre-disk x equidistant-ring blocks of the product's own raster sampler
    (spatialstats.equidistant_blocks_from_raster; disk radius = extent diagonal / sqrt(2)^rings as
    `_choose_cdist_equidistant_sampling_parameters` sets it for this grid).  samples = 9091 is reading B (1e7 points drawn in
    total), samples = 223607 reading A (subsample = 1e7 in the reference's sense, 5e13 pairs).  Returns (blocks, right edges):
    50 explicit edges geomspace(sqrt 2, maxlag, 50).  Rings that leave the raster hold fewer points -- that is the geometry."""
    import torch

    from . import spatialstats as ss

    field = fbm_torch(size, size, device, hurst=0.3, seed=seed, mean=0.0, std=1.0).reshape(-1)
    maxdist = math.sqrt(2.0) * (size - 1)
    ratio = samples / (math.pi * maxdist**2 / math.sqrt(2.0) ** (2 * rings))   # res = 1: spatialstats.py:1176-1181

    def values_of(idx):
        return field[torch.from_numpy(np.ascontiguousarray(idx)).to(device)].cpu().numpy()

    rng = np.random.default_rng(seed)
    blocks = ss.equidistant_blocks_from_raster(None, 1.0, runs, samples, ratio, rng, values_of=values_of, shape=(size, size))
    edges = np.geomspace(math.sqrt(2.0), maxdist, 50)
    return blocks, edges
