"""Per-attribute ulp histogram and true relative error of the terrain kernel against the CPU oracle (measurement tool).

  python tools/ulp_report.py [--gpu] [--size N] [--fit Florinsky] [--json out.json]

Without --gpu the kernel's math header (xdem_amd/csrc/terrain_math.h) runs through the host-compiled harness of
tests/hostsim; with --gpu the HIP kernel runs through the C-ABI.  Reports, per attribute: share of pixels at 0, 1, 2, 3-4,
5-8, >8 ulp from the oracle, the worst ulp distance, and the worst TRUE relative error |got-ref|/|ref| over pixels where
|ref| is not within 1e-3 of the attribute's scale from zero (zero crossings carry no relative information)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index",
        "terrain_ruggedness_index"]


def ulp_stats(got, ref):
    from hostsim_util import ulp_diff

    fin = np.isfinite(ref) & np.isfinite(got)
    d = ulp_diff(got[fin], ref[fin])
    n = d.size
    edges = [0, 1, 2, 4, 8]
    out = {"n": int(n), "ulp0": float(np.mean(d == 0)), "ulp1": float(np.mean(d == 1)), "ulp2": float(np.mean(d == 2)),
           "ulp3_4": float(np.mean((d > 2) & (d <= 4))), "ulp5_8": float(np.mean((d > 4) & (d <= 8))),
           "ulp_gt8": float(np.mean(d > 8)), "max_ulp": int(d.max()) if n else 0}
    r = ref[fin].astype(np.float64)
    g = got[fin].astype(np.float64)
    scale = np.percentile(np.abs(r), 99) if n else 1.0
    away = np.abs(r) > 1e-3 * scale
    out["max_rel_away_from_zero"] = float(np.max(np.abs(g[away] - r[away]) / np.abs(r[away]))) if away.any() else 0.0
    out["max_scaled"] = float(np.max(np.abs(g - r) / np.maximum(np.abs(r), scale))) if n else 0.0
    out["nan_equal"] = bool(np.array_equal(np.isnan(got), np.isnan(ref)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--size", type=int, default=700)
    ap.add_argument("--fit", default="Florinsky")
    ap.add_argument("--res", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--json", default=None)
    ap.add_argument("--tail", type=int, default=2, help="2 lean tail (default), 0 mixed tail of round 2, 1 float64 (GPU only)")
    a = ap.parse_args()
    import terrain_oracle as to
    from xdem_amd.synth import fbm_numpy

    dem = fbm_numpy((a.size, a.size + 57), seed=a.seed, dtype=np.float32)
    attrs = FULL if a.fit != "Horn" else ["slope", "aspect", "hillshade", "topographic_position_index",
                                          "terrain_ruggedness_index"]
    ref = to.terrain_attributes(dem, attrs, resolution=a.res, surface_fit=a.fit)
    if a.gpu:
        from xdem_amd import terrain

        from xdem_amd import _lib

        _lib.default_context().set_option("terrain_math", a.tail)
        got = terrain.get_terrain_attribute(dem, attrs, resolution=a.res, surface_fit=a.fit)
    else:
        from hostsim_util import hostsim_terrain

        got = hostsim_terrain(dem, attrs, resolution=a.res, surface_fit=a.fit, tail=a.tail)
    rep = {}
    print(f"{'attribute':32s} {'0ulp':>7s} {'1ulp':>7s} {'2ulp':>7s} {'3-4':>7s} {'5-8':>7s} {'>8':>7s} {'max':>5s} {'max rel (away from 0)':>22s}")
    for n, g, r in zip(attrs, got, ref):
        s = ulp_stats(g, r)
        rep[n] = s
        print(f"{n:32s} {s['ulp0']:7.4f} {s['ulp1']:7.4f} {s['ulp2']:7.4f} {s['ulp3_4']:7.4f} {s['ulp5_8']:7.4f} "
              f"{s['ulp_gt8']:7.4f} {s['max_ulp']:5d} {s['max_rel_away_from_zero']:22.3e}  nan_equal={s['nan_equal']}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"source": "gpu" if a.gpu else "hostsim", "size": a.size, "fit": a.fit, "res": a.res, "attrs": rep}, f, indent=1)


if __name__ == "__main__":
    main()
