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


def cpu_baseline(n: int = 6144) -> dict:
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


def secondary_cpu_baselines() -> dict:
    """The CPU ports (oracles) of the two other paths on bounded samples (reported as rates, never extrapolated)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np

    import nuthkaab_oracle as nko
    import variogram_oracle as vo
    from xdem_amd.synth import fbm_numpy

    out = {}
    m = 2000
    rng = np.random.default_rng(1)
    ref = fbm_numpy((m, m), seed=42)
    tba = (np.roll(ref, (1, -2), (0, 1)) + 2.0).astype(np.float32)
    tba[rng.uniform(size=(m, m)) < 0.2] = np.nan
    st, asp = nko.aux_vars(ref)
    valid = np.isfinite(ref) & np.isfinite(tba) & np.isfinite(st) & np.isfinite(asp)
    t0 = time.perf_counter()
    nko.iteration_step((0.0, 0.0, 0.0), ref, tba, valid, st, asp, (10.0, 10.0), 72)
    dt = time.perf_counter() - t0
    out["nuthkaab"] = {"value": round(m * m / dt / 1e6, 3), "unit": "Mpixel_iterations_s", "cores": 1, "kind": "port",
                       "sample": f"{m}x{m} pair, one iteration step, oracle/nuthkaab_oracle.py, {dt:.1f} s"}
    n = 5000
    x, y = rng.uniform(0, 20000, n), rng.uniform(0, 20000, n)
    v = (np.sin(x / 900) + 0.2 * rng.normal(size=n)).astype(np.float32)
    edges = np.geomspace(np.sqrt(2), np.hypot(20000, 20000), 50)
    t0 = time.perf_counter()
    _, c = vo.empirical_variogram_blocks([(x, y, v)], edges, "dowd")
    dt = time.perf_counter() - t0
    out["variogram"] = {"value": round(float(c.sum()) / dt / 1e9, 5), "unit": "Gpairs_s", "cores": 1, "kind": "port",
                        "sample": f"pdist of {n} points ({int(c.sum())} pairs), 50 classes, Dowd, oracle/variogram_oracle.py, {dt:.1f} s"}
    return out


def secondary_metrics(ctx, dev) -> dict:
    """The other two hot paths at BASELINE.json's configurations (reported next to the headline metric, not part of
    `value`): C5 reading B of SURVEY.md 8d for the variogram, C3 for Nuth-Kaab."""
    import numpy as np
    import torch

    from xdem_amd import coreg
    from xdem_amd import spatialstats as ss
    from xdem_amd.synth import fbm_torch

    out = {}
    # variogram C5-B: 1e7 sampled points = 100 runs x (9091 centre + 90910 ring points), 8.3e10 pairs, 50 lag classes
    rng = np.random.default_rng(45)
    runs, samples, rings, L = 100, 9091, 10, 20000.0
    blocks = []
    for _ in range(runs):
        ax, ay = rng.uniform(0, L, samples), rng.uniform(0, L, samples)
        bx, by = rng.uniform(0, L, samples * rings), rng.uniform(0, L, samples * rings)
        av = (np.sin(ax / 900.0) + 0.2 * rng.normal(size=samples)).astype(np.float32)
        bv = (np.sin(bx / 900.0) + 0.2 * rng.normal(size=samples * rings)).astype(np.float32)
        blocks.append((ax, ay, av, bx, by, bv))
    edges = np.geomspace(np.sqrt(2), np.hypot(L, L), 50)
    ps = ss.PairSet(blocks, edges, ctx)
    ps.sums(0)
    ps.sums(0)
    ms = ctx.last_kernel_ms()
    t0 = time.perf_counter()
    ss.class_medians(ps)
    dt = time.perf_counter() - t0
    out["variogram"] = {"pairs": ps.n_pairs, "lag_classes": 50, "matheron_pass_Gpairs_s": round(ps.n_pairs / ms / 1e6, 1),
                        "dowd_exact_median_Gpairs_s": round(ps.n_pairs / dt / 1e9, 2),
                        "note": "C5 (reading B): 100 blocks of 9091 x 90910 points, f32 values; Matheron = one pair pass; "
                                "Dowd = exact per-class median of |dv| (bracketed selection: sampled digit passes, one "
                                "counting + compaction pass over all pairs, exact selection among the candidates; wall time)"}
    ps.close()
    del blocks
    # Nuth-Kaab C3: 20000^2 pair, tba = ref shifted + 2 m, 20 % NaN in contiguous gaps; iteration steps on the full grid
    m = 20000
    ref = fbm_torch(m, m, dev, seed=42)
    tba = torch.roll(ref, shifts=(1, -2), dims=(0, 1)) + 2.0
    hole = fbm_torch(m, m, dev, seed=44)
    thr = torch.quantile(hole[::16, ::16].flatten(), 0.2)
    tba[hole < thr] = float("nan")
    del hole
    torch.cuda.synchronize(dev)
    plan = coreg.NKPlan(ref.contiguous(), tba.contiguous(), None, ctx)
    plan.step(0.0, 0.0, (10.0, 10.0), 72)
    t0 = time.perf_counter()
    k = 3
    for i in range(k):
        r = plan.step(3.0 + i, -4.0, (10.0, 10.0), 72)
    dt = (time.perf_counter() - t0) / k
    out["nuthkaab"] = {"grid": f"{m}x{m}", "valid_fraction": round(r["n_valid"] / (m * m), 3),
                       "Mpixel_iterations_s": round(m * m / dt / 1e6, 1), "ms_per_iteration": round(dt * 1e3, 2),
                       "note": "C3: one iteration = shifted dh, exact nanmedian, 72-bin exact medians of dh/slope_tan (float32), "
                               "host 72-point fit excluded"}
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
    # Test knob: XDEM_BENCH_SHARE_GPU=1 runs every rank on GPU 0 over the gloo backend (halo rows staged through the host) so
    # that the multi-rank flow can be exercised on a single-GPU box; RCCL (one rank per GPU) is the measured configuration.
    share = os.environ.get("XDEM_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
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

    if world > 1:  # communicator set-up (RCCL creates its point-to-point channels lazily) is not a step: do it up front
        xdist.RowBlock.wait_all(block.exchange())
        barrier()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device="cpu" if share else dev, dtype=torch.float64)
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
            "metric": f"Mpixels/s full terrain-attribute set, {n}\u00b2 f32 DEM",
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
                       "partition": f"{world} row block(s), halo depth {depth}" +
                                    (", shared-GPU gloo test mode" if share else (", RCCL send/recv" if world > 1 else "")),
                       "bytes_per_pixel": BYTES_PER_PIXEL},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": (lambda t: None if t is None else round(t / (kernel_ms * 1e-3) / 1e9, 1))(
                             measured_traffic_bytes(px_launch) if world == 1 else None),
                         "traffic_bytes_per_launch": measured_traffic_bytes(px_launch) if world == 1 else None,
                         "kernel": "terrain_tile_kernel<Florinsky,curv,win,f32,f32>",
                         "kernel_ms": round(kernel_ms, 4), "pixels_per_launch": px_launch},
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline()
        if not args.no_secondary and world == 1:
            try:
                del out, block
                torch.cuda.empty_cache()
                res["secondary"] = secondary_metrics(ctx, dev)
                if not args.no_cpu_baseline:
                    res["secondary"]["cpu_baseline"] = secondary_cpu_baselines()
            except Exception as e:  # the headline line must still be printed
                res["secondary"] = {"error": repr(e)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
