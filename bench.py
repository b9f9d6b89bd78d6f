#!/usr/bin/env python
"""bench.py -- throughput of the terrain hot path (BASELINE.json metric: Mpixels/s, full terrain-attribute set,
40000^2 float32 DEM), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the fused kernel over the whole raster (all 11 attributes: slope, aspect, hillshade,
profile / tangential / planform / flowline / max / min curvature, TPI, TRI; Florinsky fit, geometric curvatures --
the reference defaults), inputs and outputs resident in HBM.  With N > 1 the raster is row-block partitioned
(strong scaling: total work fixed) and every step includes the RCCL halo exchange.  Rank 0 prints ONE JSON line.

Extra objects on that line: "roofline" (algorithmic bytes / HIP-event kernel time vs the 8 TB/s HBM peak) and
"cpu_baseline" (the NumPy oracle = port of the reference's SciPy engine, timed on a bounded sample on this box).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT]

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index",
        "terrain_ruggedness_index"]
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak (about 6.3 TB/s achievable)
BYTES_PER_PIXEL = 4 + 4 * len(FULL)  # 4 B read + 4 B per attribute written = 48 B


def cpu_baseline(n: int = 2048) -> dict:
    """Reference-recipe CPU port (oracle/terrain_oracle.py, single thread NumPy) on a bounded n x n sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import terrain_oracle

    from xdem_amd.synth import fbm_numpy

    dem = fbm_numpy((n, n), seed=42)
    t0 = time.perf_counter()
    terrain_oracle.terrain_attributes(dem, FULL, resolution=10.0)
    dt = time.perf_counter() - t0
    return {"value": round(n * n / dt / 1e6, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": f"{n}x{n} fBm float32 DEM, full 11-attribute set, oracle/terrain_oracle.py (NumPy restatement of "
                      f"the reference SciPy engine), {dt:.1f} s, host has {os.cpu_count()} cores"}


def measured_traffic_bytes(pixels_per_launch: int):
    """HBM bytes per launch from the newest committed PMC profile of the same launch size (profiles/*_pmc.json:
    2 x FETCH_SIZE (gfx950 wide-load correction) + WRITE_SIZE, KiB units), or None."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_terrain_pmc.json"))):
        try:
            d = json.load(open(f))
            if int(d.get("_pixels_per_launch", 40000 * 40000)) == int(pixels_per_launch):
                best = (2.0 * d["FETCH_SIZE"]["mean"] + d["WRITE_SIZE"]["mean"]) * 1024.0
        except Exception:
            pass
    return best


def secondary_metrics(ctx) -> dict:
    """Small fixed-size runs of the other two hot paths (reported next to the headline metric, not part of `value`)."""
    import numpy as np

    from xdem_amd import coreg
    from xdem_amd import spatialstats as ss
    from xdem_amd.synth import fbm_numpy

    out = {}
    # variogram: 65536-point sample x 8192-point sample (5.4e8 pairs), 50 geometric lag classes, float32 values
    rng = np.random.default_rng(45)
    n = 65536
    x, y = rng.uniform(0, 20000, n), rng.uniform(0, 20000, n)
    v = (np.sin(x / 900) + 0.2 * rng.normal(size=n)).astype(np.float32)
    edges = np.geomspace(np.sqrt(2), np.hypot(20000, 20000), 50)
    ps = ss.PairSet([(x[: n // 8], y[: n // 8], v[: n // 8], x, y, v)], edges, ctx)
    ps.sums(0)
    ps.sums(0)
    ms = ctx.last_kernel_ms()
    t0 = time.perf_counter()
    ss.class_medians(ps)
    dt = time.perf_counter() - t0
    out["variogram"] = {"pairs": ps.n_pairs, "lag_classes": 50, "matheron_pass_Gpairs_s": round(ps.n_pairs / ms / 1e6, 1),
                        "dowd_exact_median_Gpairs_s": round(ps.n_pairs / dt / 1e9, 2),
                        "note": "cdist 8192 x 65536 points, f32 values; Dowd = 4 histogram passes + successor pass"}
    ps.close()
    # Nuth-Kaab: 4096^2 pair, 20 % NaN, one iteration step (all grid passes of an iteration, exact medians)
    m = 4096
    ref = fbm_numpy((m, m), seed=42)
    tba = (np.roll(ref, (1, -2), (0, 1)) + 2.0).astype(np.float32)
    hole = fbm_numpy((m, m), seed=44, hurst=1.0, mean=0.0, std=1.0)
    tba[hole < np.percentile(hole, 20)] = np.nan
    plan = coreg.NKPlan(ref, tba, None, ctx)
    plan.step(0.0, 0.0, (10.0, 10.0), 72)
    t0 = time.perf_counter()
    plan.step(3.0, -4.0, (10.0, 10.0), 72)
    dt = time.perf_counter() - t0
    out["nuthkaab"] = {"grid": f"{m}x{m}", "Mpixel_iterations_s": round(m * m / dt / 1e6, 1),
                       "note": "one iteration step: shifted dh, exact nanmedian, 72-bin exact medians (float32)"}
    plan.close()
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=40000, help="raster is size x size (40000 = the metric's DEM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the small variogram / Nuth-Kaab side measurements")
    ap.add_argument("--no-overlap", action="store_true", help="wait for the halo before launching anything")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from xdem_amd import _lib
    from xdem_amd import dist as xdist
    from xdem_amd.synth import fbm_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    n = args.size
    depth = xdist.halo_depth(FULL, "Florinsky", 3)
    block = xdist.RowBlock(n, n, depth, rank, world, dev)
    # each rank synthesises exactly its rows of the global raster (halo rows come from the neighbours)
    block.interior.copy_(fbm_torch(block.rows, n, dev, seed=42, row0=block.r0, total_rows=n))
    out = torch.empty((len(FULL), block.rows, n), device=dev, dtype=torch.float32)
    ctx = _lib.default_context(local_rank)
    kw = dict(resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)

    def step():
        xdist.terrain_row_block(block, FULL, out=out, overlap=not args.no_overlap, **kw)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # Kernel-only duration for the roofline: HIP events recorded by the library on the launch stream around the
    # kernel (xdemhip_last_kernel_ms), averaged over fresh launches of the dominant (interior / whole-block) kernel.
    from xdem_amd.terrain import terrain_attributes_device

    kms = []
    for _ in range(max(3, min(args.steps, 10))):
        terrain_attributes_device(block.buf, FULL, out=out, halo_top=block.halo_top, halo_bottom=block.halo_bottom, **kw)
        kms.append(ctx.last_kernel_ms())
    kernel_ms = sum(kms) / len(kms)
    px_launch = block.rows * n
    achieved = BYTES_PER_PIXEL * px_launch / (kernel_ms * 1e-3) / 1e9

    if rank == 0:
        total_px = float(n) * n
        res = {
            "metric": "Mpixels/s full terrain-attribute set (11 attributes), float32 DEM",
            "value": round(total_px * args.steps / elapsed / 1e6, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{n}x{n} float32 fBm DEM (H=0.7, 1000+-300 m, res 10 m), Florinsky fit, geometric "
                                   f"curvatures, 11 attributes, device-resident in/out",
                       "partition": f"{world} row block(s), halo depth {depth}" + (", RCCL send/recv" if world > 1 else ""),
                       "bytes_per_pixel": BYTES_PER_PIXEL},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": (lambda t: None if t is None else round(t / (kernel_ms * 1e-3) / 1e9, 1))(
                             measured_traffic_bytes(px_launch) if world == 1 else None),
                         "traffic_bytes_per_launch": measured_traffic_bytes(px_launch) if world == 1 else None,
                         "kernel": "terrain_tile_kernel<Florinsky,curv,win,f32,f32>",
                         "kernel_ms": round(kernel_ms, 4), "pixels_per_launch": px_launch},
        }
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        if not args.no_secondary:
            try:
                res["secondary"] = secondary_metrics(ctx)
            except Exception as e:  # the headline line must still be printed
                res["secondary"] = {"error": repr(e)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
