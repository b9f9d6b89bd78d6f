"""A/B timing of runtime options of the fused terrain kernel on one GPU (measurement tool).

  python tools/terrain_opts_bench.py [--size 40000] [--reps 4] [--rounds 3] [--opts "terrain_sync=0,2,4,8;terrain_order=0,1"]

Times the headline launch (Florinsky, 11 attributes, float32, device-resident) for every listed value of every option (the
others at their defaults), interleaved over `rounds` so that the box's clock drift hits all settings alike.  Times are the
library's own HIP events (xdemhip_last_kernel_ms)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index",
        "terrain_ruggedness_index"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=40000)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--opts", default="terrain_sync=0,2,4,8;terrain_order=0,1")
    ap.add_argument("--combos", default="")
    ap.add_argument("--json", default=None)
    ap.add_argument("--planes", default="torch", choices=("torch", "scattered"), help="torch.empty planes or the library's scattered backing")
    a = ap.parse_args()
    import torch

    from xdem_amd import _lib
    from xdem_amd.synth import fbm_torch
    from xdem_amd.terrain import terrain_attributes_device

    n = a.size
    dem = fbm_torch(n, n, "cuda", seed=42)
    ctx = _lib.default_context(0)
    if a.planes == "scattered":
        from xdem_amd.terrain import alloc_planes

        out = alloc_planes(len(FULL), n, n, torch.float32, ctx, torch.device("cuda", 0), backing="auto")
    else:
        out = torch.empty((len(FULL), n, n), dtype=torch.float32, device="cuda")
    # --opts "a=0,1;b=2" times every listed value of a and of b alone (the other options at 0);
    # --combos "a=1+b=2;a=0+b=2" times the named combinations.  Options not named in a setting are set to 0.
    settings = []
    for spec in a.opts.split(";"):
        if not spec:
            continue
        name, vals = spec.split("=")
        for v in vals.split(","):
            settings.append((f"{name}={v}", {name: v}))
    for combo in a.combos.split(";"):
        if combo:
            settings.append((combo, dict(kv.split("=") for kv in combo.split("+"))))
    res = {k: [] for k, _ in settings}
    kw = dict(resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)
    names = sorted({k for _, d in settings for k in d})
    for rnd in range(a.rounds):
        for label, d in settings:
            for k in names:
                ctx.set_option(k, int(d.get(k, 0)))
            for _ in range(a.reps):
                terrain_attributes_device(dem, FULL, out=out, **kw)
                res[label].append(ctx.last_kernel_ms())
    for k in names:
        ctx.set_option(k, {"terrain_math": 2, "terrain_stream": 1}.get(k, 0))   # back to the library defaults
    summary = {}
    for label, _ in settings:
        t = sorted(res[label])
        summary[label] = {"min": round(t[0], 3), "median": round(t[len(t) // 2], 3)}
        print(f"{label:28s} min {t[0]:8.3f} ms   median {t[len(t) // 2]:8.3f} ms   ({48 * n * n / t[len(t) // 2] / 1e6:6.1f} GB/s, frac {48 * n * n / t[len(t) // 2] / 1e6 / 8000:.3f})", flush=True)
    if a.json:
        json.dump(summary, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
