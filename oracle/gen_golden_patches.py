"""Golden vectors of the NaN-ignoring mean filter and the patches method: runs the reference's own
``xdem.spatialstats.mean_filter_nan`` / ``_patches_convolution`` / ``_patches_loop_quadrants`` (imported from /root/reference
through oracle/_refimport.py) on seeded inputs and records inputs + outputs under tests/golden/patches_golden.npz.
Container-only; re-run with  python oracle/gen_golden.py patches  (geoutils is absent: ``nmad`` is passed in as a local
function of the published definition)."""
from __future__ import annotations

import os

import numpy as np


def nmad(data, nfact: float = 1.4826):
    arr = np.asarray(data)
    return nfact * np.nanmedian(np.abs(arr - np.nanmedian(arr)))


def images():
    rng = np.random.default_rng(91)
    base = np.cumsum(np.cumsum(rng.normal(scale=0.4, size=(47, 61)), axis=0), axis=1)
    for dt in (np.float32, np.float64):
        img = (base + 300.0).astype(dt)
        img[5:9, 10:14] = np.nan
        img[30, 40] = np.nan
        img[0, 0] = np.nan
        img[46, 60] = np.inf        # non-finite counts as nodata (np.isfinite)
        img[20, 3] = -np.inf
        yield np.dtype(dt).name, img
    holes = rng.normal(size=(33, 29)).astype(np.float32)
    holes[rng.uniform(size=holes.shape) < 0.45] = np.nan
    yield "holes", holes


def main(ref, out_dir: str) -> None:
    ss = ref.spatialstats
    rec = {}
    for name, img in images():
        rec[f"img|{name}"] = img
        for shape in ("square", "circular"):
            for p in (1, 2, 3, 4, 5, 6, 8, 9, 11, 12, 13):
                if shape == "square" and p > 11:
                    continue
                mean_img, nb_valid, nb_px = ss.mean_filter_nan(img, p, shape)
                rec[f"mean|{name}|{shape}|{p}"] = np.asarray(mean_img)
                rec[f"valid|{name}|{shape}|{p}"] = np.asarray(nb_valid)
                rec[f"npx|{name}|{shape}|{p}"] = np.int64(nb_px)
    # the int8 wrap-around the reference's count suffers from beyond 127 kernel pixels (documented, not reproduced)
    _, nb_valid, nb_px = ss.mean_filter_nan(np.ones((20, 20), np.float32), 12, "square")
    rec["wrap|valid"] = np.asarray(nb_valid)
    rec["wrap|npx"] = np.int64(nb_px)
    # the patches method on top of it: convolution form (all patches, independent subsets) and the quadrant loop
    rng = np.random.default_rng(5)
    vals = (rng.normal(0, 1, (120, 150)) + 0.3 * np.sin(np.arange(150) / 9.0)[None, :]).astype(np.float32)
    vals[40:60, 70:100] = np.nan
    vals[rng.uniform(size=vals.shape) < 0.03] = np.nan
    rec["patches|values"] = vals
    for shape, area in (("circular", 60.0), ("circular", 250.0), ("square", 100.0), ("square", 36.0)):
        stat, nb, exact, df = ss._patches_convolution(vals, gsd=2.0, area=area, perc_min_valid=80.0, patch_shape=shape,
                                                      statistic_between_patches=nmad, return_in_patch_statistics=True)
        rec[f"pconv|{shape}|{area}"] = np.array([stat, nb, exact], dtype=np.float64)
        rec[f"pconv_df|{shape}|{area}"] = np.stack([df["nanmean"].values.astype(np.float64), df["count"].values.astype(np.float64)])
    for shape, area, seed in (("square", 100.0, 42), ("circular", 100.0, 42), ("square", 400.0, 7)):
        stat, nb, exact, df = ss._patches_loop_quadrants(vals, gsd=2.0, area=area, patch_shape=shape, n_patches=12, perc_min_valid=80.0,
                                                         statistic_between_patches=nmad, random_state=seed,
                                                         return_in_patch_statistics=True)
        rec[f"pquad|{shape}|{area}|{seed}"] = np.array([stat, nb, exact], dtype=np.float64)
        rec[f"pquad_tiles|{shape}|{area}|{seed}"] = np.array([t for t in df["tile"].values]) if "tile" in df else np.array([], dtype=str)
        rec[f"pquad_df|{shape}|{area}|{seed}"] = np.stack([df["nanmean"].values.astype(np.float64), df["count"].values.astype(np.float64)])
    np.savez_compressed(os.path.join(out_dir, "patches_golden.npz"), **rec)
    print(f"patches fixtures written: {len(rec)} arrays")
