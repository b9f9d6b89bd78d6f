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
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT]

FULL = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
        "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index",
        "terrain_ruggedness_index"]
C4_SIZE = int(os.environ.get("XDEM_BENCH_C4_SIZE", "65536"))  # test knob: a smaller raster on shared-GPU boxes
C3_SIZE = int(os.environ.get("XDEM_BENCH_C3_SIZE", "20000"))  # test knobs of the secondary legs (BASELINE sizes by default)
C5_RUNS = int(os.environ.get("XDEM_BENCH_C5_RUNS", "100"))
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak (about 6.3 TB/s achievable)
BYTES_PER_PIXEL = 4 + 4 * len(FULL)  # 4 B read + 4 B per attribute written = 48 B


def _cpu_terrain_tile(args):
    """Worker of the all-cores CPU leg: the oracle on one DEM tile (module level: multiprocessing 'spawn' imports it)."""
    seed, rows, cols = args
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"  # one oracle thread per process: the processes are the parallelism
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import terrain_oracle

    rng = np.random.default_rng(seed)
    dem = (1000.0 + np.cumsum(np.cumsum(rng.normal(scale=0.2, size=(rows, cols)), 0), 1)).astype(np.float32)
    t0 = time.perf_counter()
    terrain_oracle.terrain_attributes(dem, FULL, resolution=10.0)
    return time.perf_counter() - t0


def _cpu_nk_step(args):
    """Worker of the all-cores Nuth-Kaab leg: one iteration step of the oracle on its own m x m pair."""
    seed, m = args
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import nuthkaab_oracle as nko

    rng = np.random.default_rng(seed)
    ref = (1000.0 + np.cumsum(np.cumsum(rng.normal(scale=0.2, size=(m, m)), 0), 1)).astype(np.float32)
    tba = (np.roll(ref, (1, -2), (0, 1)) + 2.0).astype(np.float32)
    tba[rng.uniform(size=(m, m)) < 0.2] = np.nan
    st, asp = nko.aux_vars(ref)
    valid = np.isfinite(ref) & np.isfinite(tba) & np.isfinite(st) & np.isfinite(asp)
    t0 = time.perf_counter()
    nko.iteration_step((0.0, 0.0, 0.0), ref, tba, valid, st, asp, (10.0, 10.0), 72)
    return time.perf_counter() - t0


def _cpu_vario_pdist(args):
    """Worker of the all-cores variogram leg: exact-Dowd pdist of its own n points (oracle)."""
    seed, n = args
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import variogram_oracle as vo

    rng = np.random.default_rng(seed)
    x, y = rng.uniform(0, 20000, n), rng.uniform(0, 20000, n)
    v = (np.sin(x / 900) + 0.2 * rng.normal(size=n)).astype(np.float32)
    edges = np.geomspace(np.sqrt(2), np.hypot(20000, 20000), 50)
    t0 = time.perf_counter()
    vo.empirical_variogram_blocks([(x, y, v)], edges, "dowd")
    return time.perf_counter() - t0


def _usable_cores() -> int:
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a container that sees 256 CPUs
    but is granted a handful would only time-share 256 workers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(n: int = 6144, gpu_check: bool = True) -> dict:
    """Reference-recipe CPU port (oracle/terrain_oracle.py = NumPy restatement of the reference's SciPy engine) on bounded
    samples of the same workload, as SURVEY 8d asks: one thread, all host cores (one oracle process per core, one tile each),
    and the reference engine's own primitive -- scipy.ndimage.convolve with the five Florinsky kernels -- where SciPy is there."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import terrain_oracle

    from xdem_amd.synth import fbm_numpy

    dem = fbm_numpy((n, n), seed=42)
    t0 = time.perf_counter()
    ref = terrain_oracle.terrain_attributes(dem, FULL, resolution=10.0)
    dt = time.perf_counter() - t0
    # the oracle as the checker of this leg: the HIP path on the very sample the CPU was timed on (host buffers in and out), NaN
    # masks bit for bit and every plane within 1e-6 of the oracle relative to max(|ref|, the plane's 99th percentile)
    check = None
    if gpu_check:
        from xdem_amd import terrain

        got = terrain.get_terrain_attribute(dem, FULL, resolution=10.0)
        worst = 0.0
        for a, g, r in zip(FULL, got, ref):
            if not np.array_equal(np.isnan(g), np.isnan(r)):
                raise RuntimeError(f"bench.py: NaN mask of {a} differs from the oracle on the CPU-baseline sample")
            fin = np.isfinite(r)
            scale = float(np.percentile(np.abs(r[fin]), 99)) or 1.0
            err = float(np.max(np.abs(g[fin].astype(np.float64) - r[fin]) / np.maximum(np.abs(r[fin]), scale)))
            worst = max(worst, err)
            if err > 1e-6:
                raise RuntimeError(f"bench.py: {a} differs from the oracle by {err:.3e} (scaled) on the CPU-baseline sample")
        check = {"planes": len(FULL), "nan_masks_equal": True, "max_scaled_err": worst, "bar": 1e-6}
    del ref
    out = {"value": round(n * n / dt / 1e6, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port", "gpu_vs_oracle_on_this_sample": check,
           "sample": f"{n}x{n} fBm float32 DEM, full 11-attribute set, oracle/terrain_oracle.py (NumPy restatement of "
                     f"the reference SciPy engine), {dt:.1f} s, host has {os.cpu_count()} cores"}
    # all cores: one process per core (capped), each times the oracle on its own 1024 x 2048 tile; rate = pixels / slowest
    try:
        import multiprocessing as mp

        cores = min(_usable_cores(), 256)
        rows, cols = 512, 2048
        with mp.get_context("spawn").Pool(cores) as pool:
            pool.map(_cpu_terrain_tile, [(i, 8, 64) for i in range(cores)])  # spin the workers up (imports) outside the clock
            t0 = time.perf_counter()
            tiles = 4 * cores  # four tiles per worker: a few seconds of work each, start-up and hand-over amortised
            pool.map(_cpu_terrain_tile, [(100 + i, rows, cols) for i in range(tiles)], chunksize=4)
            wall = time.perf_counter() - t0
        out["all_cores"] = {"value": round(tiles * rows * cols / wall / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
                            "sample": f"{cores} processes (usable cores: affinity mask capped by the cgroup CPU quota; os.cpu_count() = "
                                      f"{os.cpu_count()}) x four {rows}x{cols} float32 tiles each, full 11-attribute set, {wall:.1f} s wall"}
    except Exception as e:  # pragma: no cover - the single-thread figure stands on its own
        out["all_cores"] = {"error": repr(e)}
    try:
        import scipy.ndimage

        m = 4096
        sub = dem[:m, :m]
        ks = terrain_oracle.conv_kernels("florinsky")
        t0 = time.perf_counter()
        for name, (tab, (const, power)) in ks.items():
            scipy.ndimage.convolve(sub, tab.astype(np.float64) / (const * 10.0**power), mode="constant", cval=np.nan)
        dt2 = time.perf_counter() - t0
        out["scipy_convolve"] = {"value": round(m * m / dt2 / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "reference primitive",
                                 "sample": f"scipy.ndimage.convolve x 5 Florinsky kernels on {m}x{m} float32 (the surface-fit half of "
                                           f"the reference engine, xdem/spatialstats.py:2521-2525), {dt2:.1f} s"}
    except Exception as e:  # pragma: no cover
        out["scipy_convolve"] = {"error": repr(e)}
    return out


def measured_traffic_bytes(pixels_per_launch: int):
    """HBM bytes per launch from the newest committed PMC profile of the same launch size (profiles/*_pmc.json:
    2 x FETCH_SIZE (gfx950 wide-load correction) + WRITE_SIZE, KiB units), or None."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_terrain_pmc.json"))):
        try:
            d = json.load(open(f))
            if int(d.get("_pixels_per_launch", 40000 * 40000)) == int(pixels_per_launch):
                # per launch = the strip kernel's + the frame kernel's bytes: per-kernel MEDIANS where the profile has them (the run
                # also holds a 64-row launch of the tile kernel -- the spot check -- and each kernel's first, cold dispatch)
                def per_launch(c):
                    pk = d[c].get("per_kernel")
                    return sum(v.get("median", v["mean"]) for v in pk.values()) if pk else d[c]["mean"]
                best = ((2.0 * per_launch("FETCH_SIZE") + per_launch("WRITE_SIZE")) * 1024.0, os.path.basename(f))
        except Exception:
            pass
    return best


def nk_pass_profile(pixels_per_launch: int):
    """The one data pass of the Nuth-Kaab step (nk_fused_kernel) in the newest committed PMC profile of this launch size
    (profiles/*_nk_fused_pmc.json, tools/profile_nk.sh): HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB units, the gfx950
    correction of MI355X_MICROARCH.md), the share of the launch's cycles its vector instructions were issuing, or None."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_nk_fused_pmc.json"))):
        try:
            d = json.load(open(f))
            if int(d.get("_pixels_per_launch", 0)) != int(pixels_per_launch):
                continue
            med = lambda c: d[c]["per_kernel"]["nk_fused_kernel"]["median"]
            traffic = (2.0 * med("FETCH_SIZE") + med("WRITE_SIZE")) * 1024.0
            cycles = med("GRBM_GUI_ACTIVE") / 8.0                 # (the counter sums the 8 XCDs)
            busy = med("SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0      # (per-SIMD issue cycles: 256 CUs x 4 SIMDs)
            best = {"traffic_bytes_per_launch": round(traffic), "traffic_over_touched_bytes": round(traffic / (13.0 * pixels_per_launch), 3),
                    "vector_instructions_per_64_pixel_row": round(med("SQ_INSTS_VALU") / (pixels_per_launch / 64.0), 1),
                    "vector_issue_share_of_cycles": round(busy / cycles, 3),
                    "source": os.path.basename(f) + ": rocprofv3 --pmc passes of tools/nk_trace.py (a committed profile, not re-measured in this run)"}
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
    # all host cores: one oracle process per usable core, each on its own sample; rate = work of all / slowest wall
    try:
        import multiprocessing as mp

        cores = min(_usable_cores(), 256)
        with mp.get_context("spawn").Pool(cores) as pool:
            pool.map(_cpu_nk_step, [(i, 64) for i in range(cores)])   # spin the workers up (imports) outside the clock
            mm = 1000
            t0 = time.perf_counter()
            pool.map(_cpu_nk_step, [(100 + i, mm) for i in range(2 * cores)], chunksize=2)
            wall = time.perf_counter() - t0
            out["nuthkaab"]["all_cores"] = {"value": round(2 * cores * mm * mm / wall / 1e6, 2), "unit": "Mpixel_iterations_s", "cores": cores, "kind": "port",
                                            "sample": f"{cores} processes x two {mm}x{mm} pairs each, one iteration step, {wall:.1f} s wall"}
            nn = 3000
            t0 = time.perf_counter()
            pool.map(_cpu_vario_pdist, [(200 + i, nn) for i in range(2 * cores)], chunksize=2)
            wall = time.perf_counter() - t0
            out["variogram"]["all_cores"] = {"value": round(2 * cores * (nn * (nn - 1) // 2) / wall / 1e9, 4), "unit": "Gpairs_s", "cores": cores, "kind": "port",
                                             "sample": f"{cores} processes x two pdist of {nn} points each, 50 classes, Dowd, {wall:.1f} s wall"}
    except Exception as e:  # pragma: no cover - the single-thread figures stand on their own
        out["all_cores_error"] = repr(e)
    return out


# MI355X float32 vector peak: 157.3 TFLOP/s (MI355X_MICROARCH.md) = 78.6e12 fused multiply-add lane-operations per second
# (256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz): the unit of SURVEY 8d's "12 VALU operations per pair" model.  What a wave64 stream
# of plain float32 instructions sustains when measured (tools/ubench.hip -> profiles/r03_ubench.txt: 2.9 cycles per wave
# instruction and SIMD at the nominal clock) is 54 T lane-ops/s; it is reported next to the spec-derived peak.
VALU_PEAK_TLANEOPS = 78.6
VALU_MEASURED_TLANEOPS = 54.2
PAIR_OPS_MODEL = 12         # ~12 VALU operations per pair (2 sub, 2 fma, |dv|, <= 6 compares, 1 accumulate)


def _c3_pair(dev, m: int = 20000):
    """BASELINE C3 input: ref = fBm m^2 (seed 42); tba = ref bilinearly shifted by (+1.7, -0.6) px + 2.0 m +
    N(0, 0.5 m) (seed 43); 20 % NaN in contiguous gaps (threshold of an independent smooth field, seed 44) applied to tba."""
    import torch

    from xdem_amd.synth import fbm_torch

    ref = fbm_torch(m, m, dev, seed=42)
    fx, fy = 0.7, 0.6   # fractional parts of the (+1.7 col, -0.6 row) shift; integer parts by roll
    a = torch.roll(ref, shifts=(0, -1), dims=(0, 1))
    tba = ((1 - fy) * ((1 - fx) * a + fx * torch.roll(a, shifts=(0, -1), dims=(0, 1)))
           + fy * torch.roll((1 - fx) * a + fx * torch.roll(a, shifts=(0, -1), dims=(0, 1)), shifts=(1, 0), dims=(0, 1)))
    del a
    g = torch.Generator(device=dev)
    g.manual_seed(43)
    tba += 2.0 + 0.5 * torch.randn((m, m), generator=g, device=dev, dtype=torch.float32)
    hole = fbm_torch(m, m, dev, seed=44)
    thr = torch.quantile(hole[::16, ::16].flatten(), 0.2)
    tba[hole < thr] = float("nan")
    del hole
    torch.cuda.synchronize(dev)
    return ref, tba


def secondary_metrics(ctx, dev, rank: int = 0, world: int = 1, barrier=None, c5a: bool = False) -> dict:
    """The other two hot paths at BASELINE.json's configurations (reported next to the headline metric, not part of
    `value`): C5 (reading B; reading A on request) for the variogram, C3 for Nuth-Kaab, each with its roofline
    object.  With more than one rank the pair blocks are dealt round-robin to the ranks (integer histograms / counters
    all-reduced through the library hook, sums on the host) and the C3 pair is PARTITIONED by row block (halo rows + the
    step's reductions over the process group); every rank runs this function, rank 0's dictionary is printed."""
    import numpy as np
    import torch

    from xdem_amd import coreg
    from xdem_amd import dist as xdist
    from xdem_amd import spatialstats as ss
    from xdem_amd.synth import c5_variogram_blocks

    barrier = barrier or (lambda: torch.cuda.synchronize(dev))
    group = "world" if world > 1 else None
    out = {}

    def variogram_leg(samples: int, label: str, runs: int = C5_RUNS, warm: bool = True) -> dict:
        # the SAME seeded blocks on every rank (fBm(H = 0.3) values on a 20000^2 grid, the product's equidistant disk / ring
        # sampler: raster pixels = integer-lattice coordinates -> the integer-lattice pair kernels); rank r keeps blocks r::world
        blocks, edges = c5_variogram_blocks(dev, runs=runs, samples=samples)
        total = sum(int(b[0].size) * int(b[3].size) for b in blocks)
        mine = blocks[rank::world]
        ps = ss.PairSet(mine, edges, ctx)
        del blocks
        red0 = ctx.reduction_calls()
        try:
            if warm:
                ps.sums(0)                   # warm-up (kernel load, clocks)
            barrier()
            t0 = time.perf_counter()
            s_m, c_m = ps.sums(0)
            s_m, c_m = ss._allreduce(s_m), ss._allreduce(c_m)   # (no-ops on one rank)
            barrier()
            dt_m = time.perf_counter() - t0
            ms_kernel = ctx.last_kernel_ms() if world == 1 else None
            barrier()
            t0 = time.perf_counter()
            med, c_d = ss.class_medians(ps, group)    # first call: also allocates (and first-touches) the candidate buffers
            barrier()
            dt_d_cold = time.perf_counter() - t0
            dt_d = dt_d_cold
            if warm:
                t0 = time.perf_counter()
                med, c_d = ss.class_medians(ps, group)    # timed like every other leg: after a warm-up call
                barrier()
                dt_d = time.perf_counter() - t0
        finally:
            ps.close()
        # validation inside the run: the two routes (sum kernel / bracketed exact selection) must agree on the class membership
        # of every pair, and no pair may be lost (every sampled pair lies below the extent diagonal = the last edge)
        if not np.array_equal(c_m, c_d):
            raise RuntimeError("variogram: Matheron and Dowd routes count different pairs per lag class")
        if int(c_m.sum()) != total:
            raise RuntimeError(f"variogram: {int(c_m.sum())} pairs inside the lags, {total} formed")
        if not (np.all(np.isfinite(med[c_d > 0])) and np.all(s_m[c_m > 0] >= 0)):
            raise RuntimeError("variogram: non-finite class estimate")
        mat_rate = total / (ms_kernel * 1e-3 if ms_kernel else dt_m) / 1e9     # Gpairs/s (kernel time on one GPU, wall over ranks)
        dowd_rate = total / dt_d / 1e9
        red1 = ctx.reduction_calls()
        return {"pairs": total, "lag_classes": int(len(edges)), "n_gpus": world,
                "reductions": {"through_the_host": red1[0] - red0[0], "device": red1[1] - red0[1],
                               "note": "library-hook all-reduces of the exact-Dowd selection over all timed and warm-up calls of this leg (integer histograms, counters, "
                                       "successor keys); the Matheron sums are two torch.distributed all-reduces per call on top"},
                "matheron_pass_Gpairs_s": round(mat_rate, 1), "dowd_exact_median_Gpairs_s": round(dowd_rate, 2),
                "dowd_first_call_Gpairs_s": round(total / dt_d_cold / 1e9, 2), "runs": runs, "points_per_sample": samples,
                "validated": "class counts of the Matheron and exact-Dowd routes identical, their sum = pairs formed",
                "conventions": {k: int(ctx.options.get(k, 0)) for k in ("vario_edge", "vario_diff")},
                "roofline": {"bound": "valu", "model": f"{PAIR_OPS_MODEL} VALU lane-operations per pair (SURVEY 8d); bytes per pair ~ 0",
                             "achieved": round(PAIR_OPS_MODEL * mat_rate / 1e3, 2), "achieved_dowd": round(PAIR_OPS_MODEL * dowd_rate / 1e3, 2),
                             "peak": VALU_PEAK_TLANEOPS * world, "unit": "T lane-ops/s",
                             "frac": round(PAIR_OPS_MODEL * mat_rate / 1e3 / (VALU_PEAK_TLANEOPS * world), 4),
                             "frac_dowd": round(PAIR_OPS_MODEL * dowd_rate / 1e3 / (VALU_PEAK_TLANEOPS * world), 4),
                             "peak_source": "157.3 TFLOP/s float32 vector (MI355X_MICROARCH.md) / 2 flop per lane-op",
                             "measured_issue_peak": VALU_MEASURED_TLANEOPS * world,
                             "measured_issue_peak_source": "tools/ubench.hip (profiles/r03_ubench.txt): v_fma_f32 2.9 cycles per wave "
                                                           "instruction and SIMD at the nominal 2.4 GHz",
                             "limiter": VARIO_LIMITER_NOTE},
                "note": label}

    out["variogram"] = variogram_leg(9091, "C5 reading B (SURVEY 8d): fBm(H=0.3) values on a 20000^2 grid, 100 runs of the equidistant "
                                            "sampler (centre disk x rings, 9091 points each; rings that leave the raster hold fewer), 50 "
                                            "edges geomspace(sqrt 2, maxlag); Matheron = one pair pass (kernel time), Dowd = exact per-class "
                                            "median of |dv| by bracketed selection (wall time)")
    # C5 in SURVEY 8d's PRIMARY reading A (subsample = 1e7 in the reference's sense -> runs x (centre disk x 10 rings of 223607
    # points), xdem/spatialstats.py:1104-1183): every default run carries 20 of its 100 runs (1e13 pairs, ~25 s, same in-run
    # validation; rates per pair do not depend on the number of runs), --c5a the full 100 runs / 5e13 pairs (minutes)
    a_runs = max(1, min(C5_RUNS, 100 if c5a else 20))
    out["variogram_c5a"] = variogram_leg(223607, f"C5 reading A (SURVEY 8d): subsample = 1e7 in the reference's sense -> {a_runs} of the 100 "
                                                 "runs x 223607-point samples (5e13 pairs for the 100; --c5a runs them all); each "
                                                 "route timed once, after reading B has warmed the kernels", runs=a_runs, warm=False)
    # Nuth-Kaab C3
    m = C3_SIZE
    res = (10.0, 10.0)
    ref, tba = _c3_pair(dev, m)
    px = float(m) * m
    routes = None
    import scipy.optimize

    if world == 1:
        plan = coreg.NKPlan(ref.contiguous(), tba.contiguous(), None, ctx)
        plan.step(0.0, 0.0, res, 72)
        t0 = time.perf_counter()
        k = 3
        for i in range(k):
            r = plan.step(3.0 + i, -4.0, res, 72)
        dt = (time.perf_counter() - t0) / k
        t0 = time.perf_counter()
        offsets = coreg._iterate(plan, res, 0.0, 10, 72, scipy.optimize.curve_fit, True)
        dt_fit = (time.perf_counter() - t0) / 10
        # SETTLED steps (round 6): what the later iterations of a fit look like -- the offsets move by ~1e-4 px per step -- timed at the
        # fit's end point; such steps take their brackets from the previous step's exact medians (xdemhip_nk_predict_counts)
        k = 5
        t0 = time.perf_counter()
        for i in range(k):
            plan.step(offsets[0] + 2e-3 * ((i * 7) % 5 - 2), offsets[1] + 1.5e-3 * ((i * 3) % 5 - 2), res, 72)
        dt_settled = (time.perf_counter() - t0) / k
        nk_red = None
        n_valid = r["n_valid"]
        routes = plan.route_counts()
        plan.close()
        how = "one GPU"
    else:
        r0, r1 = xdist.row_block(m, world, rank)
        ref_rows, tba_rows = ref[r0:r1].contiguous(), tba[r0:r1].contiguous()
        del ref, tba
        torch.cuda.empty_cache()
        xdist.nuth_kaab_row_blocks(ref_rows, tba_rows, m, res, halo=8, ctx=ctx, tolerance=0.0, max_iterations=2)   # warm-up
        barrier()
        t0 = time.perf_counter()
        nk_info = {}
        offsets, n_valid = xdist.nuth_kaab_row_blocks(ref_rows, tba_rows, m, res, halo=8, ctx=ctx, tolerance=0.0, max_iterations=10, info=nk_info)
        barrier()
        dt_fit = (time.perf_counter() - t0) / 10
        dt = dt_fit
        dt_settled = None
        routes = nk_info.get("routes")
        red = nk_info.get("reductions", (0, 0))
        nk_red = {"through_the_host": int(red[0]), "device": int(red[1]), "per_iteration": round((red[0] + red[1]) / 10, 1)}
        how = (f"row blocks of {world} ranks + halo rows, every reduction of a step through the process group "
               f"({(red[0] + red[1]) / 10:.1f} all-reduces per iteration: {red[0]} staged through the host, {red[1]} enqueued on the device)")
    # full data passes per iteration: ONE on the one-pass step of round 4 (dh, the counting for its median and the aspect-bin
    # counting against sample brackets in the same pass), two on the queued route of rounds 2-3 (row-partitioned fits)
    onepass = bool(routes and routes["onepass"] > 0 and routes["plain"] == 0)
    passes = 1 if onepass else 2
    alg_bpp = 16 if onepass else 8 * passes   # 8 B/pixel/pass recomputing the aux rasters, 16 B/pixel/pass with stored aux arrays
    touched = 13 if onepass else NK_TOUCHED_BYTES
    # the roofline fraction is priced at what the kernels can at most be credited with: the SMALLER of SURVEY's algorithmic figure
    # and the bytes the step really touches (one-pass: 13 < 16 -- pricing it at 16 would credit bytes that never move, and could
    # exceed the physical peak; two passes: 16 < 21).  The 16 B figure stays next to it, labelled, for comparison across rounds.
    roof_bpp = min(alg_bpp, touched)
    # validation inside the run: the fit must find the shift the pair was built with
    sx, sy, sz = -offsets[0] / res[0], -offsets[1] / res[1], offsets[2]
    if not (abs(sx - 1.7) < 0.05 and abs(sy - 0.6) < 0.05 and abs(sz + 2.0) < 0.05):
        raise RuntimeError(f"Nuth-Kaab: fitted shift ({sx:.3f}, {sy:.3f}, {sz:.3f}) px / m is not the (1.7, 0.6, -2.0) the pair was built with")
    out["nuthkaab"] = {"grid": f"{m}x{m}", "n_gpus": world, "partition": how, "valid_fraction": round(n_valid / (m * m), 3),
                       "Mpixel_iterations_s": round(px / dt / 1e6, 1), "ms_per_iteration": round(dt * 1e3, 2),
                       "ms_per_iteration_whole_fit": round(dt_fit * 1e3, 2),
                       "ms_per_iteration_settled": None if dt_settled is None else round(dt_settled * 1e3, 3),
                       "settled_roofline_frac": None if dt_settled is None else round(roof_bpp * px / dt_settled / 1e9 / (HBM_PEAK_GBPS * world), 4),
                       "fitted_shift_px": [round(sx, 3), round(sy, 3), round(sz, 3)],
                       "validated": "the 10-iteration fit recovers the (+1.7, +0.6) px, -2.0 m shift the pair was built with",
                       "routes": routes, "nk_nan_rule": int(ctx.options.get("nk_nan_rule", 0)),
                       "reductions": nk_red,
                       "roofline": {"bound": "hbm", "model": ("SURVEY 8d: 16 B/pixel per data pass with stored aux arrays (ref 4 + tba 4 + slope tangent + aspect bin) "
                                                              "x P passes; P = 1: the one-pass step counts for the median of dh and for the 72 bin medians in the "
                                                              "same pass, against brackets from a 1/64 sample (8 B/pixel/pass x 2 passes in rounds 2-3: the same 16)"
                                                              if onepass else
                                                              "SURVEY 8d: 8 B/pixel (ref + tba) per data pass x P passes required by the exact "
                                                              "medians; P = 2 here (bracketed selections: one counting pass each for the global median and "
                                                              "the 72 aspect bins; plain radix passes would need 12)"),
                                    "passes": passes, "algorithmic_bytes_per_pixel": alg_bpp, "roofline_bytes_per_pixel": roof_bpp,
                                    "achieved": round(roof_bpp * px / dt / 1e9, 1), "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                                    "frac": round(roof_bpp * px / dt / 1e9 / (HBM_PEAK_GBPS * world), 4),
                                    "frac_at_survey_bytes": round(alg_bpp * px / dt / 1e9 / (HBM_PEAK_GBPS * world), 4),
                                    "touched_bytes_per_pixel": touched,
                                    "touched_GBps": round(touched * px / dt / 1e9, 1),
                                    "data_pass": nk_pass_profile(m * m) if (onepass and world == 1) else None,
                                    "note": NK_ONEPASS_NOTE if onepass else NK_TOUCHED_NOTE},
                       "note": "ms_per_iteration = steps whose shift moves by 0.1 px (the early iterations of a fit: sampled brackets); ms_per_iteration_settled = "
                               "steps at the fit's end point, shift changes of ~2e-4 px (the later iterations: brackets predicted from the previous step's exact "
                               "medians, routes.predicted counts them; results identical either way).  "
                               "C3: one iteration = shifted dh, exact nanmedian, 72-bin exact medians of dh/slope_tan (float32); "
                               "ms_per_iteration = grid work of a step (host 72-point fit excluded; with more than one rank: the whole fit "
                               "per iteration), ms_per_iteration_whole_fit = NuthKaab's 10-iteration loop incl. scipy curve_fit, per iteration"}
    return out


VARIO_LIMITER_NOTE = ("vector-instruction issue in both passes.  Matheron (round 4): a lane keeps the d^2 interval of its run's lag class, a pair that stays "
                      "in the class costs a subtract and a compare instead of the class lookup; only class changes (5 % of the lane-pairs, a third of "
                      "the wave-pairs on this geometry) look up, flush the run to LDS and reload.  Exact Dowd = three sampled digit passes for both "
                      "bracket ends (3.7 ms) + ONE counting / compaction pass over all pairs (38 ms) + the selection among the 0.22 % of the pairs "
                      "inside the brackets (1.6 ms); the counting pass reads the Morton-ordered copy and counts run-length as well (round 4, "
                      "profiles/r04_vario_counting_pmc.json: 17.4 vector instructions per wave-pair against 20.8 per pair-wise form, VALU busy 0.82)")
NK_ONEPASS_NOTE = ("the one pass touches 13 B/pixel: masked reference copy 4 + tba 4 + slope tangent 4 + cached aspect-bin id 1 (2 until round 6); no dh raster is "
                   "written or re-read (22 B/pixel in two passes in round 3); candidates of the medians (a few percent) leave as (dh, slope "
                   "tangent, bin) triples; from the second step of a plan on the sample brackets are half as wide as the rule for fully "
                   "correlated sample lines when the rank offsets measured in the earlier steps allow it (exact either way)")
NK_TOUCHED_BYTES = 21
NK_TOUCHED_NOTE = ("the two passes touch 21 B/pixel (dh pass: masked reference copy 4 + tba 4 + dh out 4 -- min / max aspect come from the "
                   "plan's lists of extreme-aspect pixels, the inlier mask is folded into the reference copy as NaN; bin pass: dh 4 + "
                   "slope_tan 4 + cached aspect-bin id 1): the aux rasters are stored, not recomputed; 27 B/pixel in round 2")


def public_functions_leg(ctx, dev, n: int = 16384) -> dict:
    """Two public functions next to the paths (last paragraph), device-resident, never part of `value`:
    `spatialstats.convolution` (SURVEY 8a row a5 as upstream exposes it) with five dense 5 x 5 filters and with three 3 x 3 filters in one
    call each, and the per-bin lookup behind `get_perbin_nd_binning` (8f row f3) on two float32 variables; kernel times from the
    context's events, a crop of the convolution checked against torch's float64 conv2d (plumbing check, not the parity claim)."""
    import ctypes

    import numpy as np
    import torch

    from xdem_amd import _lib
    from xdem_amd import spatialstats as ss
    from xdem_amd.synth import fbm_torch

    out = {"grid": f"{n}x{n} float32"}
    dem = fbm_torch(n, n, dev, seed=9)[None].contiguous()
    rng = np.random.default_rng(17)
    for label, filt in (("five_5x5_filters", rng.normal(size=(5, 5, 5))), ("three_3x3_filters", rng.normal(size=(3, 3, 3)))):
        res = ss.convolution(dem, filt, method="scipy", ctx=ctx)
        torch.cuda.synchronize(dev)
        ms = []
        for _ in range(3):
            res = ss.convolution(dem, filt, method="scipy", ctx=ctx)
            torch.cuda.synchronize(dev)
            ms.append(ctx.last_kernel_ms())
        m = filt.shape[1]
        crop = dem[:, 1000:1100, 2000:2100].double()[:, None]
        want = torch.nn.functional.conv2d(crop, torch.from_numpy(filt[:, None, ::-1, ::-1].copy()).to(dev))   # (conv2d correlates: flip for a convolution)
        got = res[0, :, 1000 + m // 2:1100 - m // 2, 2000 + m // 2:2100 - m // 2]
        err = float((got - want[0]).abs().max() / want.abs().max())
        if not err < 1e-6:   # (float32-rounded sums against unrounded float64 ones)
            raise RuntimeError(f"convolution leg: crop differs from conv2d by {err:.3e}")
        bpp = 4 + 8 * filt.shape[0]
        out[label] = {"ms": round(min(ms), 4), "bytes_per_pixel": bpp, "GBps": round(bpp * n * n / min(ms) / 1e6, 1),
                      "frac_of_hbm_peak": round(bpp * n * n / min(ms) / 1e6 / HBM_PEAK_GBPS, 4), "crop_vs_conv2d_rel": err}
        del res, got, want
    a = torch.rand((n, n), device=dev) * 40
    b = torch.rand((n, n), device=dev) * 5
    res = torch.empty((n, n), dtype=torch.float64, device=dev)
    na = nb = 10
    left = np.concatenate([np.arange(na) * 4.0, np.arange(nb) * 0.5])
    right = np.concatenate([(np.arange(na) + 1) * 4.0, (np.arange(nb) + 1) * 0.5])
    table = rng.normal(size=na * nb)
    kind = np.ones(na * nb, dtype=np.uint8)
    dp = ctypes.POINTER(ctypes.c_double)
    miss = ctypes.c_int64()
    ms = []
    for _ in range(3):
        ctx.check(ctx._L.xdemhip_perbin_lookup(ctx.handle, (ctypes.c_void_p * 2)(a.data_ptr(), b.data_ptr()), (ctypes.c_int * 2)(_lib.F32, _lib.F32), 2, n * n,
                                               (ctypes.c_int * 2)(na, nb), left.ctypes.data_as(dp), right.ctypes.data_as(dp), table.ctypes.data_as(dp),
                                               kind.ctypes.data_as(ctypes.c_char_p), 1, res.data_ptr(), ctypes.byref(miss), _lib.DEVICE))
        ms.append(ctx.last_kernel_ms())
    ia, ib = (a[5, :7] / 4.0).long().cpu().numpy(), (b[5, :7] / 0.5).long().cpu().numpy()
    if not np.array_equal(res[5, :7].cpu().numpy(), table[ia * nb + ib]):
        raise RuntimeError("per-bin lookup leg: wrong bin values")
    out["perbin_lookup_two_variables"] = {"ms": round(min(ms), 4), "bytes_per_pixel": 16, "GBps": round(16 * n * n / min(ms) / 1e6, 1),
                                          "frac_of_hbm_peak": round(16 * n * n / min(ms) / 1e6 / HBM_PEAK_GBPS, 4), "bins": na * nb}
    return out


def isa_cycles() -> dict:
    """Vector-issue cycles per output row of the streaming kernels of the launches timed here (newest profiles/*_isa_cycles.json,
    written by tools/isa_ledger.py --json from the compiled kernels at the per-instruction costs of tools/ubench.hip).  The cost
    table is in cycles at the nominal 2.4 GHz, i.e. seconds x 2.4e9: issue time of a launch = cycles_per_row x (pixels / 64) /
    (1024 SIMDs) / 2.4e9.  (Check of the model: the headline kernel with its stores compiled out -- pure vector work -- runs in
    8.09-8.12 ms, the model says 8.09: profiles/r06_terrain_bound_variants.txt.)"""
    import glob

    f = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_isa_cycles.json")))
    if not f:
        return {}
    try:
        return {"file": os.path.basename(f[-1]), **json.load(open(f[-1]))}
    except Exception:
        return {}


def issue_bound(name: str, pixels: float, kernel_ms: float, hbm_frac: float, cyc: dict) -> dict | None:
    """Fraction of the launch's time its vector instructions need to issue (the roofline that binds the small attribute sets: they are
    float64-issue-bound, not HBM-bound) next to the HBM fraction, and which of the two is nearer."""
    e = (cyc.get("sets") or {}).get(name)
    if not e:
        return None
    ms = e["cycles_per_row"] * (pixels / 64.0) / 1024.0 / 2.4e9 * 1e3
    return {"issue_ms": round(ms, 3), "issue_frac": round(ms / kernel_ms, 4), "vector_instructions_per_row": e["vector_instructions_per_row"],
            "float64_class_per_row": e["float64_class_per_row"], "nearer_bound": "vector issue" if ms / kernel_ms > hbm_frac else "hbm",
            "source": "profiles/" + cyc.get("file", "?")}


def terrain_sets(ctx, dev, dem, kw, steps: int) -> dict:
    """The SMALL attribute sets users ask for far more often than all eleven planes -- DEM.slope(), slope + aspect, a hillshade
    -- and the full set with directional curvatures, on the headline raster: rate, bytes per pixel (4 read + 4 per plane
    written) and fraction of the HBM roofline per launch (HIP events on the launch stream, planes from terrain.alloc_planes)."""
    import torch

    from xdem_amd import terrain

    H, W = dem.shape
    cases = [("slope", ["slope"], dict(surface_fit="Florinsky")),
             ("slope+aspect Horn (DEM.slope() / aspect() defaults of the reference's examples)", ["slope", "aspect"], dict(surface_fit="Horn")),
             ("slope+aspect Florinsky", ["slope", "aspect"], dict(surface_fit="Florinsky")),
             ("hillshade", ["hillshade"], dict(surface_fit="Florinsky")),
             ("slope+aspect+hillshade Florinsky", ["slope", "aspect", "hillshade"], dict(surface_fit="Florinsky")),
             ("full 11, directional curvatures", FULL, dict(surface_fit="Florinsky", curv_method="directional")),
             ("full 11, ZevenbergThorne fit (3x3)", FULL, dict(surface_fit="ZevenbergThorne")),
             ("full 11, ZevenbergThorne fit, directional curvatures", FULL, dict(surface_fit="ZevenbergThorne", curv_method="directional"))]
    out = {}
    cyc = isa_cycles()
    for name, attrs, extra in cases:
        planes = terrain.alloc_planes(len(attrs), H, W, torch.float32, ctx, dev)
        k = dict(kw)
        k.update(extra)
        for _ in range(2):
            terrain.terrain_attributes_device(dem, attrs, out=planes, **k)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a_, b_ in ev:
            a_.record()
            terrain.terrain_attributes_device(dem, attrs, out=planes, **k)
            b_.record()
        torch.cuda.synchronize(dev)
        ms = sorted(a_.elapsed_time(b_) for a_, b_ in ev)
        med = ms[len(ms) // 2]
        bpp = 4 + 4 * len(attrs)
        gbps = bpp * float(H) * W / (med * 1e-3) / 1e9
        out[name] = {"planes": len(attrs), "bytes_per_pixel": bpp, "kernel_ms_median": round(med, 4), "kernel_ms_min": round(ms[0], 4),
                     "Mpixels_s": round(float(H) * W / (med * 1e-3) / 1e6, 1), "achieved_GBps": round(gbps, 1),
                     "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4),
                     "issue": issue_bound(name, float(H) * W, med, gbps / HBM_PEAK_GBPS, cyc)}
        del planes
    return {"raster": f"{H}x{W} float32 (the headline DEM)", "steps": steps, "sets": out,
            "note": "one launch per step, device-resident; fractions against the 8 TB/s HBM peak at 4 + 4 K bytes per pixel; `issue` = the share of "
                    "the launch's time its vector instructions need to issue (static count of the compiled kernel x measured cost per instruction): "
                    "the one-to-three-plane sets sit at 0.73-0.83 of THAT bound and 0.44-0.60 of the HBM one"}


def device_state() -> list:
    """What rocm-smi says about the GPU the line was measured on (partition modes, power cap, memory / fabric clock levels): the
    same binary runs the headline launch in 12.7-13.3 ms on most boxes of the pool and 14.5-15.0 ms on others (DESIGN.md section 1,
    profiles/r03_box_variance.txt) -- recorded so that a line can be read against the box it came from.  Best effort."""
    import subprocess

    try:
        txt = subprocess.run(["rocm-smi", "--showmemorypartition", "--showcomputepartition", "--showmaxpower", "--showclocks", "--showtemp"],
                             capture_output=True, text=True, timeout=15).stdout
        keep = ("Partition", "Max Graphics", "mclk", "fclk", "sclk", "junction")
        return [" ".join(l.split()) for l in txt.splitlines() if l.startswith("GPU[0]") and any(k in l for k in keep)]
    except Exception:
        return []


class GpuSampler:
    """Shader clock / memory clock / socket power / temperatures of the GPU DURING a timed loop, so that a slow line can be read
    against the state of the box it came from (the same binary runs the headline launch at 13.1 ms on most boxes and 13.9 ms on
    some; an idle rocm-smi snapshot cannot tell a power-managed clock from a slow placement of the planes).  A thread reads the
    amdgpu sysfs files of the device every `period` seconds -- `pp_dpm_sclk` / `pp_dpm_mclk` (the starred level = the current
    clock), hwmon `freq1_input`, `power1_average` / `power1_input`, `temp*_input` -- no subprocess, no SMI library, nothing the
    GPU executes.  Best effort: whatever is unreadable is left out, and an empty result says so."""

    def __init__(self, local_rank: int = 0, period: float = 0.01):
        import glob
        import threading

        self.period = period
        self.samples = []          # (t, {name: value})
        self._stop = threading.Event()
        self._thread = None
        cards = []
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(d, "vendor")).read().strip() == "0x1002" and glob.glob(os.path.join(d, "hwmon", "hwmon*")):
                    cards.append(d)
            except Exception:
                pass
        self.dev = cards[local_rank] if local_rank < len(cards) else (cards[0] if cards else None)
        self.files = {}
        if self.dev:
            hw = (glob.glob(os.path.join(self.dev, "hwmon", "hwmon*")) or [None])[0]
            cand = {"sclk_dpm": os.path.join(self.dev, "pp_dpm_sclk"), "mclk_dpm": os.path.join(self.dev, "pp_dpm_mclk"),
                    "busy_pct": os.path.join(self.dev, "gpu_busy_percent")}
            if hw:
                cand.update({"sclk_hz": os.path.join(hw, "freq1_input"), "mclk_hz": os.path.join(hw, "freq2_input"),
                             "power_avg_uW": os.path.join(hw, "power1_average"), "power_in_uW": os.path.join(hw, "power1_input"),
                             "power_cap_uW": os.path.join(hw, "power1_cap"),
                             "temp1_mC": os.path.join(hw, "temp1_input"), "temp2_mC": os.path.join(hw, "temp2_input"),
                             "temp3_mC": os.path.join(hw, "temp3_input")})
            for k, f in cand.items():
                try:
                    open(f).read()
                    self.files[k] = f
                except Exception:
                    pass

    @staticmethod
    def _parse(name, txt):
        txt = txt.strip()
        if name.endswith("_dpm"):   # "0: 132Mhz\n1: 2400Mhz *": the starred level is the current clock
            for line in txt.splitlines():
                if line.rstrip().endswith("*"):
                    m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
                    return float(m.group(1)) if m else None
            return None
        try:
            return float(txt)
        except ValueError:
            return None

    def _read(self):
        out = {}
        for k, f in self.files.items():
            try:
                v = self._parse(k, open(f).read())
                if v is not None:
                    out[k] = v
            except Exception:
                pass
        return out

    def _loop(self):
        while not self._stop.is_set():
            self.samples.append((time.perf_counter(), self._read()))
            self._stop.wait(self.period)

    def start(self):
        import threading

        if self.files:
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)

    def summary(self, t0: float, t1: float) -> dict:
        """Statistics of the samples taken inside [t0, t1] (perf_counter times of the timed region)."""
        if not self.files:
            return {"available": False, "why": "no readable amdgpu sysfs files (/sys/class/drm/card*/device)"}
        inside = [s for t, s in self.samples if t0 <= t <= t1] or [s for _, s in self.samples]

        def stat(key, scale):
            v = sorted(s[key] * scale for s in inside if key in s)
            return None if not v else {"mean": round(sum(v) / len(v), 1), "min": round(v[0], 1), "max": round(v[-1], 1)}

        out = {"available": True, "samples_in_timed_region": len([1 for t, _ in self.samples if t0 <= t <= t1]),
               "period_ms": self.period * 1e3, "source": "amdgpu sysfs (" + self.dev + ")",
               "sclk_MHz": stat("sclk_hz", 1e-6) or stat("sclk_dpm", 1.0), "mclk_MHz": stat("mclk_hz", 1e-6) or stat("mclk_dpm", 1.0),
               "power_W": stat("power_avg_uW", 1e-6) or stat("power_in_uW", 1e-6), "power_cap_W": stat("power_cap_uW", 1e-6),
               "temp_C": {k: stat(k, 1e-3) for k in ("temp1_mC", "temp2_mC", "temp3_mC") if stat(k, 1e-3)},
               "gpu_busy_pct": stat("busy_pct", 1.0)}
        return out


def end_to_end_host_path(ctx, n: int = 16384) -> dict:
    """SURVEY 8d "separate end-to-end number including H2D/D2H": BASELINE C2 (16384^2 float32, 11 attributes) through the call
    users make -- get_terrain_attribute(ndarray) -> list of ndarrays -- host buffers in and out, PCIe both ways."""
    import numpy as np

    from xdem_amd import terrain
    from xdem_amd.synth import fbm_numpy

    tile = fbm_numpy((4096, 4096), seed=42)
    dem = np.tile(tile, (n // 4096, n // 4096)) + np.linspace(0, 50, n, dtype=np.float32)[None, :]
    terrain.get_terrain_attribute(dem[:1024], FULL, resolution=10.0)   # warm-up: library load, pinned staging buffers
    t0 = time.perf_counter()
    out = terrain.get_terrain_attribute(dem, FULL, resolution=10.0)
    dt = time.perf_counter() - t0
    ok = all(o.shape == dem.shape for o in out)
    gb = (4 + 4 * len(FULL)) * float(n) * n / 1e9
    return {"workload": f"C2: {n}x{n} float32 DEM, 11 attributes, NumPy arrays in and out (H2D + kernel + D2H)", "seconds": round(dt, 3),
            "Mpixels_s": round(float(n) * n / dt / 1e6, 1), "effective_GBps_over_PCIe": round(gb / dt, 1), "shapes_ok": bool(ok),
            "note": "never the reported `value` (that is device-resident); PCIe Gen5 x16 = 63 GB/s per direction"}


def end_to_end_calls(ctx, dev) -> dict:
    """The other two paths through the calls users make, HOST arrays in (never the reported rates): sample_empirical_variogram on a
    20000^2 NumPy raster (subsample 1e6, one variogram = 100 runs, exact Dowd: sampling of the equidistant metric space, gather,
    Morton-ordered copies and lattice packing in the library's native host code, upload, pair passes, exact medians) and
    NuthKaab(subsample=1).fit on the C3 pair as NumPy arrays (uploads, plan creation, ten iterations, ten host curve fits)."""
    import torch

    from xdem_amd import coreg, spatialstats
    from xdem_amd.synth import fbm_torch

    out = {}
    dh = fbm_torch(20000, 20000, dev, seed=3, hurst=0.3).cpu().numpy()
    kw = dict(gsd=10.0, estimator="dowd", random_state=42)
    spatialstats.sample_empirical_variogram(dh[:2000, :2000], subsample=1000, **kw)   # (first launches)
    t0 = time.perf_counter()
    df = spatialstats.sample_empirical_variogram(dh, subsample=1000000, **kw)
    dt = time.perf_counter() - t0
    pairs = float(df["count"].sum())
    out["variogram"] = {"workload": "sample_empirical_variogram(20000x20000 float32 NumPy raster, subsample=1e6, estimator='dowd'): host raster in, DataFrame out",
                        "seconds": round(dt, 3), "pairs_in_kept_classes": int(pairs), "Gpairs_s": round(pairs / dt / 1e9, 1),
                        "note": "profiles/r06_vario_end_to_end.txt has the C5 size (subsample 1e7: 22 s, 2.4 Tpairs/s end to end) and the breakdown"}
    del dh
    ref, tba = _c3_pair(dev, 20000)
    ref, tba = ref.cpu().numpy(), tba.cpu().numpy()
    torch.cuda.empty_cache()
    coreg.NuthKaab(subsample=1, max_iterations=2).fit(ref[:2000, :2000].copy(), tba[:2000, :2000].copy(), None, resolution=10.0)
    t0 = time.perf_counter()
    nk = coreg.NuthKaab(subsample=1, max_iterations=10, offset_threshold=0.0).fit(ref, tba, None, resolution=10.0)
    dt = time.perf_counter() - t0
    a = nk.meta["outputs"]["affine"]
    out["nuthkaab"] = {"workload": "NuthKaab(subsample=1, max_iterations=10).fit on the C3 pair as NumPy arrays (20000x20000 float32 each)",
                       "seconds": round(dt, 3), "shift": [round(float(a["shift_x"]), 3), round(float(a["shift_y"]), 3), round(float(a["shift_z"]), 3)]}
    # ... and with the reference's DEFAULT subsample (5e5 of the valid pixels, drawn once: ranks on the host, pixels on the device)
    t0 = time.perf_counter()
    nk = coreg.NuthKaab(max_iterations=10, offset_threshold=0.0).fit(ref, tba, None, resolution=10.0, random_state=42)
    dt = time.perf_counter() - t0
    a = nk.meta["outputs"]["affine"]
    out["nuthkaab_default_subsample"] = {"workload": "NuthKaab(max_iterations=10).fit (subsample=5e5, the reference's default) on the same NumPy arrays",
                                         "seconds": round(dt, 3), "subsample_final": int(nk.meta["outputs"]["random"]["subsample_final"]),
                                         "shift": [round(float(a["shift_x"]), 3), round(float(a["shift_y"]), 3), round(float(a["shift_z"]), 3)]}
    t0 = time.perf_counter()
    moved = coreg.apply_translation(tba, 17.0, 6.0, 2.0, 10.0)
    out["apply_translation"] = {"workload": "apply_translation of the 20000x20000 float32 NumPy raster (upload, bilinear resample, download)",
                                "seconds": round(time.perf_counter() - t0, 3), "shape_ok": bool(moved.shape == tba.shape)}
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
    ap.add_argument("--c5a", action="store_true", help="also run the variogram at C5 reading A (5e13 pairs: about two minutes)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-buffer (PCIe-inclusive) C2 figure")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from xdem_amd import _lib, terrain
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

    if world > 1 and dist.get_world_size() != args.gpus:
        raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    if not share and torch.cuda.device_count() < min(args.gpus, 8):
        raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible: one rank per GPU is the measured configuration")
    depth = xdist.halo_depth(FULL, "Florinsky", 3)
    ctx = _lib.default_context(local_rank)
    kw = dict(resolution=10.0, surface_fit="Florinsky", curv_method="geometric", ctx=ctx)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def partitioned_run(n, steps, warmup, backing=None, block=None):
        """`steps` timed passes of the 11-attribute set over the n x n raster held as `world` row blocks; max over ranks."""
        fresh = block is None
        if fresh:
            block = xdist.RowBlock(n, n, depth, rank, world, dev)
        # each rank synthesises exactly its rows of the global raster (halo rows come from the neighbours)
        if fresh:
            block.interior.copy_(fbm_torch(block.rows, n, dev, seed=42, row0=block.r0, total_rows=n))
        # resident planes from the library's allocator (XDEM_BENCH_PLANES = torch | scattered | contiguous | chunked
        # forces one backing for measurements).  The default, "auto" WITH A PROBE: three to eight candidate placements are tried in turn --
        # one virtual range over 32 MiB physical pieces in pseudo-random order, an ordinary allocation, and again, until a second
        # candidate has come within 3 % of the fastest --, this rank's
        # own launch is timed on each (two untimed + three timed launches, no halo exchange: every rank decides alone, no rank waits
        # for another) and the fastest is kept: which placement the memory controller likes better depends on the state of the
        # box's free device memory and differs from allocation to allocation (terrain.alloc_planes).  All of it before the warm-up.
        which = backing or os.environ.get("XDEM_BENCH_PLANES", "auto")
        out = terrain.alloc_planes(len(FULL), block.rows, n, torch.float32, ctx, dev, backing=which,
                                   probe=(lambda planes: terrain.terrain_attributes_device(block.buf, FULL, out=planes, halo_top=block.halo_top,
                                                                                           halo_bottom=block.halo_bottom, **kw)) if which == "auto" else None)
        chosen = {"backing": getattr(out, "_xdem_backing", which), "calibration_ms": getattr(out, "_xdem_calibration_ms", None),
                  "calibration_s": getattr(out, "_xdem_calibration_s", None), "calibration_launches_per_candidate": getattr(out, "_xdem_calibration_launches", None)}

        def step():
            xdist.terrain_row_block(block, FULL, out=out, overlap=not args.no_overlap, **kw)

        if world > 1:  # communicator set-up (RCCL creates its point-to-point channels lazily) is not a step: do it up front
            xdist.RowBlock.wait_all(block.exchange())
            barrier()
        sampler = GpuSampler(local_rank).start()   # clock / power / temperature while the warm-up and the timed steps run
        # HIP events around every step ON THE LAUNCH STREAM (the library launches on torch's current stream here), recorded
        # inside the timed region and read after it: the same launches under both clocks
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        # the shader clock UNDER the launch: one sleeping wave per timed step on a side stream (xdemhip_clock_probe: ticks of the
        # shader-clock counter against the constant 100 MHz one; ~7 ms each, inside its step) -- nothing the timed stream waits for
        side = torch.cuda.Stream(device=dev)
        ticks = torch.zeros((steps, 2), dtype=torch.int64, device=dev)
        # (events, side stream and probe buffer exist BEFORE the warm-up: between the warm-up's last launch and the first timed one
        # there is only the contract's barrier + synchronize -- session r06aq: with the set-up in that gap the GPU sat idle long enough
        # to leave its clock state, and the first timed step took 14-15 ms against 12.7-12.8 ms for the rest, kernel_ms_series)
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for i, (a_, b_) in enumerate(ev):
            a_.record()
            step()
            b_.record()
            try:
                ctx.clock_probe(side, ticks[i], sleeps=1600)
            except Exception:
                pass
        barrier()
        t1 = time.perf_counter()
        sampler.stop()
        tk = ticks.cpu().numpy().astype("float64")
        mhz = [100.0 * c / w for c, w in tk if w > 0]
        clock = None if not mhz else {"mean_GHz": round(sum(mhz) / len(mhz) / 1e3, 4), "min_GHz": round(min(mhz) / 1e3, 4),
                                      "max_GHz": round(max(mhz) / 1e3, 4), "probes": len(mhz),
                                      "source": "xdemhip_clock_probe: one sleeping wave per timed step next to the launch, s_memtime "
                                                "(shader clock) / s_memrealtime (100 MHz)"}
        t = torch.tensor([t1 - t0], device="cpu" if share else dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        step_ms = [a_.elapsed_time(b_) for a_, b_ in ev]
        state = sampler.summary(t0, t1)
        state["shader_clock_under_load"] = clock
        busy = (state.get("gpu_busy_pct") or {}).get("mean")
        if busy is not None and busy < 50.0:
            state["sysfs_note"] = (f"gpu_busy_percent read {busy} % while this GPU ran back-to-back launches: on this box the sysfs sensors do "
                                   "not follow this GPU's load (stale or another device's) -- power / sclk from sysfs are not evidence here")
        state["planes"] = chosen
        return float(t.item()), block, out, step_ms, state

    n = args.size
    elapsed, block, out, step_ms, gpu_state = partitioned_run(n, args.steps, args.warmup)
    kernel_ms = sum(step_ms) / len(step_ms)
    # A/B of the plane backing inside the same process (one GPU): the same launches on planes a CALLER would bring -- an ordinary
    # allocation (torch.empty = hipMalloc), which on some boxes is one physically contiguous block -- after the library's
    # scattered backing above.  Every driver run is thereby a data point of "is >= 0.70 the kernel's or the allocator's"
    # (XDEM_BENCH_AB=0 skips it).
    # Spot check of the TIMED output (not a parity claim -- that is the test suite's and the cpu_baseline leg's): 64 rows from the
    # middle of this rank's planes against a second, small launch over just those rows + halo (another launch geometry: its own
    # strips and frame tiles, halo rows instead of raster rows above and below) -- bit for bit; and the planes hold no value the
    # set cannot produce
    lo = max(depth, (block.rows // 2) & ~31)
    hi = min(lo + 64, block.rows - depth)
    spot = None
    if hi - lo >= 8:
        crop = terrain.terrain_attributes_device(block.interior[lo - depth:hi + depth], FULL, halo_top=depth, halo_bottom=depth, **kw)
        torch.cuda.synchronize(dev)
        same = bool(torch.equal(torch.nan_to_num(crop, nan=-7.0e33), torch.nan_to_num(out[:, lo:hi], nan=-7.0e33)))
        # (the outer `depth` columns have windows that leave the raster: NaN there, values inside)
        sl, hs = out[0, lo:hi, depth:n - depth], out[2, lo:hi, depth:n - depth]
        slope_ok = (bool(((sl >= 0) & (sl <= 90)).all()) and bool(((hs >= 0) & (hs <= 255)).all())
                    and bool(torch.isnan(out[0, lo:hi, :depth]).all()) and bool(torch.isnan(out[0, lo:hi, n - depth:]).all()))
        spot = {"rows": [int(lo), int(hi)], "planes_equal_a_separate_launch_bit_for_bit": same, "slope_and_hillshade_in_range_nan_only_in_the_border_columns": slope_ok}
        if not (same and slope_ok):
            raise SystemExit(f"bench.py: spot check of the timed planes failed: {spot}")
        del crop
    ab = None
    gpu_state2 = None
    if world == 1 and os.environ.get("XDEM_BENCH_AB", "1") == "1" and os.environ.get("XDEM_BENCH_PLANES", "auto") == "auto":
        del out
        import gc

        gc.collect()
        # (the OTHER placement: the ordinary allocation when the calibration kept the scattered pieces, the scattered pieces when it kept
        # the ordinary allocation -- every line shows both)
        other = "scattered" if (gpu_state.get("planes") or {}).get("backing") == "torch" else "torch"
        _, _, out, ms2, gpu_state2 = partitioned_run(n, args.steps, args.warmup, backing=other, block=block)
        k2 = sum(ms2) / len(ms2)
        ab = {"planes": "torch.empty (ordinary hipMalloc: what a caller of the C-ABI brings)" if other == "torch" else
                        "the library's scattered 32 MiB-piece backing (the calibration of this run kept the ordinary allocation)",
              "backing": other, "kernel_ms": round(k2, 4),
              "kernel_ms_min": round(min(ms2), 4), "kernel_ms_max": round(max(ms2), 4),
              "achieved": round(BYTES_PER_PIXEL * block.rows * n / (k2 * 1e-3) / 1e9, 1),
              "frac": round(BYTES_PER_PIXEL * block.rows * n / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
              "gpu_state_during_timed_steps": gpu_state2}

    # Kernel duration for the roofline: the mean of the HIP-event times of the K timed steps themselves (events recorded on the
    # launch stream around each step; one step = the streaming kernel over the raster interior + the tile kernel over its frame
    # of edge tiles, plus the halo exchange when the raster is partitioned).  By construction kernel_ms <= ms_per_step.
    px_launch = block.rows * n
    achieved = BYTES_PER_PIXEL * px_launch / (kernel_ms * 1e-3) / 1e9

    # C4 (BASELINE.json configs[3]): 65536^2 over the row blocks of all ranks with the halo exchange -- run by every rank whenever
    # the job has more than one, reported under "secondary" (the headline stays the metric's 40000^2 raster)
    c4 = None
    if world > 1 and not args.no_secondary and n != C4_SIZE:
        del out, block
        torch.cuda.empty_cache()
        c4_steps = max(2, min(args.steps, 5))
        c4_elapsed, block, out, _, _ = partitioned_run(C4_SIZE, c4_steps, max(1, min(args.warmup, 2)))
        c4 = {"workload": f"C4: {C4_SIZE}x{C4_SIZE} float32 fBm DEM, 11 attributes, {world} row blocks, halo depth {depth}, "
                          + ("shared-GPU gloo test mode" if share else "RCCL send/recv over xGMI"),
              "value": round(float(C4_SIZE) ** 2 * c4_steps / c4_elapsed / 1e6, 1), "unit": "Mpixels/s", "n_gpus": world,
              "steps": c4_steps, "ms_per_step": round(c4_elapsed / c4_steps * 1e3, 4)}

    if rank == 0:
        total_px = float(n) * n
        res = {
            "metric": f"Mpixels/s full terrain-attribute set, {n}\u00b2 f32 DEM; variogram Gpairs/s"
                      + ("" if not args.no_secondary else " (this line: terrain half only)"),
            "value": round(total_px * args.steps / elapsed / 1e6, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 in/out, mixed f64/f32 arithmetic (f64 where cancellation demands: stencil sums, curvature numerators, discriminants)",
            "data": "synthetic",
            "config": {"workload": f"{n}x{n} float32 fBm DEM (H=0.7, 1000+-300 m, res 10 m), Florinsky fit, geometric "
                                   f"curvatures, 11 attributes, device-resident in/out (planes: " + ({"scattered": "the library's scattered 32 MiB-piece backing", "torch": "an ordinary allocation"}.get((gpu_state.get("planes") or {}).get("backing"), str((gpu_state.get("planes") or {}).get("backing"))) + (", kept by the allocator's calibration of both placements before the warm-up" if (gpu_state.get("planes") or {}).get("calibration_ms") else "")) + ")",
                       # what a step exchanges: `depth` rows of 4-byte pixels with each neighbour, sent and received (interior ranks: 2 neighbours)
                       "halo_bytes_per_step_and_rank": 0 if world == 1 else int(2 * min(2, world - 1) * depth * n * 4),
                       "partition": f"{world} row block(s), halo depth {depth}" +
                                    (", shared-GPU gloo test mode" if share else (", RCCL send/recv" if world > 1 else "")),
                       "bytes_per_pixel": BYTES_PER_PIXEL},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": (lambda t: None if t is None else round(t[0] / (kernel_ms * 1e-3) / 1e9, 1))(
                             measured_traffic_bytes(px_launch) if world == 1 else None),
                         "traffic_bytes_per_launch": (lambda t: None if t is None else t[0])(
                             measured_traffic_bytes(px_launch) if world == 1 else None),
                         "traffic_source": (lambda t: None if t is None else
                                            f"profiles/{t[1]}: rocprofv3 --pmc passes of this command (2 x FETCH_SIZE + WRITE_SIZE), "
                                            "a committed profile, not re-measured in this run")(
                             measured_traffic_bytes(px_launch) if world == 1 else None),
                         "kernel": "terrain_strip_kernel<Florinsky,curv,win,lean tail> (raster interior, 98.6 % of the pixels) + "
                                   "terrain_tile_kernel (frame of edge tiles)",
                         "kernel_ms_source": "mean HIP-event time of the timed steps themselves (events on the launch stream)",
                         "kernel_ms": round(kernel_ms, 4), "kernel_ms_min": round(min(step_ms), 4),
                         "kernel_ms_max": round(max(step_ms), 4), "kernel_ms_series": [round(x, 3) for x in step_ms],
                         "pixels_per_launch": px_launch, "output_spot_check": spot,
                         # what the GPU was doing WHILE the timed steps ran (sysfs samples: see GpuSampler)
                         "issue": issue_bound("headline: full 11, Florinsky, geometric curvatures", float(px_launch), kernel_ms, achieved / HBM_PEAK_GBPS, isa_cycles()),
                         "clock_GHz": (lambda c: None if not c else c["mean_GHz"])((gpu_state or {}).get("shader_clock_under_load")),
                         "clock_GHz_caller_planes": (lambda c: None if not c else c["mean_GHz"])((gpu_state2 or {}).get("shader_clock_under_load")),
                         "power_W": (lambda c: None if not c else c["mean"])(None if (gpu_state or {}).get("sysfs_note") else (gpu_state or {}).get("power_W")),
                         "gpu_state_during_timed_steps": gpu_state},
        }
        res["roofline"]["planes"] = gpu_state.get("planes")   # which placement the timed planes have, and the calibration that chose it
        if ab is not None:
            # frac_caller_planes = the ORDINARY allocation (what a caller of the C-ABI brings), frac_scattered_planes = the library's pieces:
            # one of the two is the headline's own `frac`, the other the A/B leg's
            main_is_torch = (gpu_state.get("planes") or {}).get("backing") == "torch"
            res["roofline"]["frac_caller_planes"] = round(achieved / HBM_PEAK_GBPS, 4) if main_is_torch else ab["frac"]
            res["roofline"]["kernel_ms_caller_planes"] = round(kernel_ms, 4) if main_is_torch else ab["kernel_ms"]
            res["roofline"]["frac_scattered_planes"] = ab["frac"] if main_is_torch else round(achieved / HBM_PEAK_GBPS, 4)
            res["roofline"]["kernel_ms_scattered_planes"] = ab["kernel_ms"] if main_is_torch else round(kernel_ms, 4)
            res["roofline"]["other_planes"] = ab
        res["config"]["rccl_ranks"] = world if world == 1 else dist.get_world_size()
        res["config"]["visible_gpus"] = torch.cuda.device_count()
        res["device_state"] = device_state()
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline()
        if c4 is not None:
            res["secondary"] = {"c4_terrain_row_blocks": c4}
    sec = None
    failed = []
    if not args.no_secondary:   # every rank runs the secondary legs (they shard over the ranks); rank 0 reports
        sets = None
        if world == 1:
            try:
                del out
                sets = terrain_sets(ctx, dev, block.interior, kw, max(3, min(args.steps, 5)))
            except Exception as e:
                import traceback

                traceback.print_exc()
                sets = {"error": repr(e)}
                failed.append("terrain_sets")
        try:
            out = block = None
            import gc

            gc.collect()
            torch.cuda.empty_cache()
            sec = secondary_metrics(ctx, dev, rank, world, barrier, c5a=args.c5a)
        except Exception as e:  # the headline line must still be printed -- and the process must not look healthy afterwards
            import traceback

            traceback.print_exc()
            sec = {"error": repr(e)}
            failed.append("secondary")
        if sets is not None:
            sec["terrain_sets"] = sets
        if world == 1 and "error" not in sec:
            try:
                sec["public_functions"] = public_functions_leg(ctx, dev)
            except Exception as e:
                import traceback

                traceback.print_exc()
                sec["public_functions"] = {"error": repr(e)}
                failed.append("public_functions")
    if rank == 0:
        if sec is not None:
            res.setdefault("secondary", {}).update(sec)
            if not args.no_cpu_baseline and world == 1 and "error" not in sec:
                res["secondary"]["cpu_baseline"] = secondary_cpu_baselines()
        if world == 1 and not args.no_end_to_end and not args.no_secondary:
            try:
                res["end_to_end"] = end_to_end_host_path(ctx)
            except Exception as e:
                res["end_to_end"] = {"error": repr(e)}
                failed.append("end_to_end")
            try:
                res["end_to_end_calls"] = end_to_end_calls(ctx, dev)
            except Exception as e:
                res["end_to_end_calls"] = {"error": repr(e)}
                failed.append("end_to_end_calls")
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if failed:   # a broken secondary path may not produce a green-looking record: the line is out, the exit code says so
        raise SystemExit(f"bench.py: failed legs: {failed}")


if __name__ == "__main__":
    main()
