"""Multi-rank terrain path on real kernels: 2 and 3 processes share the one GPU of the test box (gloo rendezvous, halo rows
staged through the host -- RCCL refuses several ranks on one device), each computes its row block exactly as bench.py does
(`RowBlock` + `terrain_row_block`, with and without the exchange/compute overlap), and the gathered result must be
bit-identical to the single-process full raster."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

ATTRS = ["slope", "aspect", "hillshade", "profile_curvature", "tangential_curvature", "planform_curvature",
         "flowline_curvature", "max_curvature", "min_curvature", "topographic_position_index", "terrain_ruggedness_index"]


def _worker(rank, world, port, n, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xdem_amd import _lib
        from xdem_amd import dist as xd
        from xdem_amd.synth import fbm_torch

        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        depth = xd.halo_depth(ATTRS, "Florinsky", 3)
        block = xd.RowBlock(n, n, depth, rank, world, dev)
        block.buf.fill_(float("nan"))
        block.interior.copy_(fbm_torch(block.rows, n, dev, seed=42, row0=block.r0, total_rows=n))
        ctx = _lib.default_context(0)
        res = {}
        for overlap in (True, False):
            out = xd.terrain_row_block(block, ATTRS, overlap=overlap, resolution=10.0, surface_fit="Florinsky",
                                       curv_method="geometric", ctx=ctx)
            torch.cuda.synchronize()
            res[overlap] = out.cpu().numpy()
        assert np.array_equal(res[True].view(np.int32), res[False].view(np.int32))
        np.save(os.path.join(outdir, f"rank{rank}.npy"), res[True])
        # second configuration: 7 x 7 windowed indexes (generic window kernel, halo depth 3 > the 3 x 3 fit's 1) + ZT fit
        attrs2 = ["slope", "max_curvature", "topographic_position_index", "terrain_ruggedness_index", "roughness"]
        depth2 = xd.halo_depth(attrs2, "ZevenbergThorne", 7)
        assert depth2 == 3
        block2 = xd.RowBlock(n, n, depth2, rank, world, dev)
        block2.buf.fill_(float("nan"))
        block2.interior.copy_(block.interior)
        for overlap in (True, False):
            out2 = xd.terrain_row_block(block2, attrs2, overlap=overlap, resolution=10.0, surface_fit="ZevenbergThorne",
                                        window_size=7, ctx=ctx)
            torch.cuda.synchronize()
            res[overlap] = out2.cpu().numpy()
        assert np.array_equal(res[True].view(np.int32), res[False].view(np.int32))
        np.save(os.path.join(outdir, f"rank{rank}_w7.npy"), res[True])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_blocks_on_real_kernels_equal_full_raster(world, tmp_path):
    from xdem_amd.synth import fbm_torch
    from xdem_amd.terrain import terrain_attributes_device

    n = 1537  # not a multiple of the world sizes or of the tile height
    ctx = mp.get_context("spawn")
    port = 29700 + (os.getpid() % 1000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    dev = torch.device("cuda", 0)
    full = terrain_attributes_device(fbm_torch(n, n, dev, seed=42), ATTRS, resolution=10.0, surface_fit="Florinsky",
                                     curv_method="geometric")
    torch.cuda.synchronize()
    full = full.cpu().numpy()
    got = np.concatenate([np.load(os.path.join(str(tmp_path), f"rank{r}.npy")) for r in range(world)], axis=1)
    assert got.shape == full.shape
    assert np.array_equal(got.view(np.int32), full.view(np.int32))
    attrs2 = ["slope", "max_curvature", "topographic_position_index", "terrain_ruggedness_index", "roughness"]
    full2 = terrain_attributes_device(fbm_torch(n, n, dev, seed=42), attrs2, resolution=10.0, surface_fit="ZevenbergThorne",
                                      window_size=7)
    torch.cuda.synchronize()
    got2 = np.concatenate([np.load(os.path.join(str(tmp_path), f"rank{r}_w7.npy")) for r in range(world)], axis=1)
    assert np.array_equal(got2.view(np.int32), full2.cpu().numpy().view(np.int32))


def _worker_reductions(rank, world, port, outdir):
    """Nuth-Kaab step and Dowd variogram with the data sharded over `world` ranks (one GPU shared, gloo): every rank must
    end up with the single-process results (integer histograms / counters all-reduced through the library hook)."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xdem_amd import _lib, coreg
        from xdem_amd import spatialstats as ss
        from xdem_amd.synth import fbm_numpy

        torch.cuda.set_device(0)
        ctx = _lib.default_context(0)
        res = {}
        for m in (300, 3072):  # plain digit passes / bracketed selection (>= 4 M pixels per rank)
            ref = fbm_numpy((m, m), seed=5)
            rng = np.random.default_rng(6)
            tba = (np.roll(ref, (1, -1), (0, 1)) + 1.5 + rng.normal(0, 0.3, (m, m))).astype(np.float32)
            tba[rng.uniform(size=(m, m)) < 0.1] = np.nan
            ctx.set_option("selection", 3 if m > 1000 else 0)  # (3: bracketed for the 72 aspect bins as well at this size)
            plan = coreg.NKPlan(ref, tba, None, ctx, group="world")
            d = plan.step(2.0, -3.0, (10.0, 10.0), 72)
            if m == 300:  # bin_statistic = mean: per-bin float64 sums / counts all-reduced through the hook
                plan.set_statistic(np.nanmean)
                dm = plan.step(2.0, -3.0, (10.0, 10.0), 72)
                res["nkmean"] = np.concatenate([dm["counts"], dm["medians"]])
            plan.close()
            ctx.set_option("selection", 0)
            res[f"nk{m}"] = np.concatenate([[d["vshift"], d["n_valid"], d["y_mean"]], d["counts"], d["medians"], d["edges"]])
        rng = np.random.default_rng(9)
        blocks = []
        for _ in range(6):
            ax, ay = rng.uniform(0, 5000, 700), rng.uniform(0, 5000, 700)
            bx, by = rng.uniform(0, 5000, 3000), rng.uniform(0, 5000, 3000)
            av = np.round(np.sin(ax / 500) + 0.2 * rng.normal(size=700), 2).astype(np.float32)
            bv = np.round(np.sin(bx / 500) + 0.2 * rng.normal(size=3000), 2).astype(np.float32)
            blocks.append((ax, ay, av, bx, by, bv))
        edges = np.geomspace(np.sqrt(2), 7100.0, 30)
        mine = blocks[rank::world]
        e, c = ss.empirical_variogram_pairs(mine, edges, "dowd", ctx)
        res["dowd"] = np.concatenate([e, c.astype(np.float64)])
        e, c = ss.empirical_variogram_pairs(mine, edges, "matheron", ctx)
        res["matheron_count"] = c.astype(np.float64)
        res["matheron"] = e
        np.savez(os.path.join(outdir, f"red{rank}.npz"), **res)
    finally:
        dist.destroy_process_group()


def test_sharded_reductions_on_real_kernels(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    port = 29800 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker_reductions, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    # single-process references
    from xdem_amd import coreg
    from xdem_amd import spatialstats as ss
    from xdem_amd.synth import fbm_numpy

    got = [np.load(os.path.join(str(tmp_path), f"red{r}.npz")) for r in range(world)]
    for m in (300, 3072):
        ref = fbm_numpy((m, m), seed=5)
        rng = np.random.default_rng(6)
        tba = (np.roll(ref, (1, -1), (0, 1)) + 1.5 + rng.normal(0, 0.3, (m, m))).astype(np.float32)
        tba[rng.uniform(size=(m, m)) < 0.1] = np.nan
        plan = coreg.NKPlan(ref, tba, None)
        d = plan.step(2.0, -3.0, (10.0, 10.0), 72)
        if m == 300:
            plan.set_statistic(np.nanmean)
            dm = plan.step(2.0, -3.0, (10.0, 10.0), 72)
            for g in got:
                assert np.array_equal(g["nkmean"][:72], dm["counts"])
                assert np.allclose(g["nkmean"][72:], dm["medians"], rtol=1e-6, atol=0, equal_nan=True)  # (means are rounded to float32)
        plan.close()
        want = np.concatenate([[d["vshift"], d["n_valid"], d["y_mean"]], d["counts"], d["medians"], d["edges"]])
        for g in got:
            a = g[f"nk{m}"]
            assert np.array_equal(np.delete(a, 2), np.delete(want, 2), equal_nan=True), m   # exact: selections, counts, edges
            assert np.isclose(a[2], want[2], rtol=1e-12)                                      # float64 sum of y: order differs
    rng = np.random.default_rng(9)
    blocks = []
    for _ in range(6):
        ax, ay = rng.uniform(0, 5000, 700), rng.uniform(0, 5000, 700)
        bx, by = rng.uniform(0, 5000, 3000), rng.uniform(0, 5000, 3000)
        av = np.round(np.sin(ax / 500) + 0.2 * rng.normal(size=700), 2).astype(np.float32)
        bv = np.round(np.sin(bx / 500) + 0.2 * rng.normal(size=3000), 2).astype(np.float32)
        blocks.append((ax, ay, av, bx, by, bv))
    edges = np.geomspace(np.sqrt(2), 7100.0, 30)
    e, c = ss.empirical_variogram_pairs(blocks, edges, "dowd")
    em, cm = ss.empirical_variogram_pairs(blocks, edges, "matheron")
    for g in got:
        assert np.array_equal(g["dowd"], np.concatenate([e, c.astype(np.float64)]), equal_nan=True)
        assert np.array_equal(g["matheron_count"], cm.astype(np.float64))
        assert np.allclose(g["matheron"], em, rtol=1e-12, atol=0, equal_nan=True)


def _worker_nk_blocks(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xdem_amd import _lib, coreg
        from xdem_amd import dist as xd

        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        ctx = _lib.default_context(0)
        ref, tba, inl = _nk_pair()
        H = ref.shape[0]
        r0, r1 = xd.row_block(H, world, rank)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a[r0:r1])).to(dev)
        # (1) one step on a partitioned plan == the same step on the whole rasters
        rb = xd.RowBlock(H, ref.shape[1], 4, rank, world, dev)
        tb = xd.RowBlock(H, ref.shape[1], 4, rank, world, dev)
        ib = xd.RowBlock(H, ref.shape[1], 4, rank, world, dev, dtype=torch.uint8)
        for b, a in ((rb, ref), (tb, tba), (ib, inl.astype(np.uint8))):
            b.interior.copy_(to(a))
            xd.RowBlock.wait_all(b.exchange())
        plan = coreg.NKPlan(rb.buf, tb.buf, ib.buf, ctx, "world", block=(H, r0, r1, rb.halo_top, rb.halo_bottom))
        d = plan.step(12.0, -17.0, (10.0, 10.0), 72)   # 1.7 rows down: inside the 4-row halo
        nv = plan.n_valid
        try:
            plan.step(0.0, 55.0, (10.0, 10.0), 72)      # 5.5 rows: outside
            raised = False
        except coreg.HaloTooSmall:
            raised = True
        plan.close()
        # (2) the whole fit through the helper (halo 2 -> has to grow for this pair's 4-row shift)
        off, n_final = xd.nuth_kaab_row_blocks(to(ref), to(tba), H, (10.0, 10.0), inlier_rows=to(inl.astype(np.uint8)), halo=2, ctx=ctx)
        np.savez(os.path.join(outdir, f"nkb{rank}.npz"), step=np.concatenate([[d["vshift"], d["n_valid"], nv, float(raised)], d["counts"],
                                                                              d["medians"], d["edges"]]),
                 fit=np.array([*off, n_final], dtype=np.float64))
    finally:
        dist.destroy_process_group()


def _nk_pair():
    from xdem_amd.synth import fbm_numpy

    m = 3000  # 4.5e6 pixels per rank: from 2^22 on the queued route with the lean kernels runs (block geometry: roff > 0, halo rows)
    ref = fbm_numpy((m, m), seed=15, std=200.0)
    rng = np.random.default_rng(16)
    tba = (np.roll(ref, (4, -2), (0, 1)) + 1.5 + rng.normal(0, 0.2, (m, m))).astype(np.float32)
    tba[rng.uniform(size=(m, m)) < 0.05] = np.nan
    inl = np.ones((m, m), dtype=bool)
    inl[1490:1510, 300:900] = False   # straddles the 2-rank block boundary
    return ref, tba, inl


@pytest.mark.parametrize("rule", [None, 3])
def test_nuth_kaab_partitioned_row_blocks(tmp_path, monkeypatch, rule):
    """the pair is PARTITIONED -- every rank holds only its row block + halo rows (RowBlock exchange) --
    not replicated; steps and the whole fit are identical to the single-process results on the full rasters (exact
    selections, integer histograms all-reduced through the hook).
    rule = 3: the same under the dilating nodata rule, handed to the ranks the way a decision would reach them (a decision file):
    row blocks then run the streaming kernels through a bad-bit mask of their own buffer (round 5; the generic kernel before)."""
    from conftest import decided
    from xdem_amd import _lib

    prev = decided("nk_nan_rule")
    if rule is not None:
        import json

        path = os.path.join(str(tmp_path), "decision.json")
        json.dump({"nk_nan_rule": rule}, open(path, "w"))
        monkeypatch.setenv("XDEM_THIRDPARTY_DECISION", path)
        _lib.default_context().set_option("nk_nan_rule", rule)
    try:
        _partitioned_row_blocks(tmp_path)
    finally:
        if rule is not None:
            _lib.default_context().set_option("nk_nan_rule", prev)   # (monkeypatch puts the environment back)


def _partitioned_row_blocks(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_worker_nk_blocks, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    hung = [p for p in procs if p.exitcode is None]
    for p in hung:
        p.kill()
    assert not hung and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    from xdem_amd import coreg

    ref, tba, inl = _nk_pair()
    plan = coreg.NKPlan(ref, tba, inl)
    d = plan.step(12.0, -17.0, (10.0, 10.0), 72)
    nv = plan.n_valid
    plan.close()
    want_step = np.concatenate([[d["vshift"], d["n_valid"], nv, 1.0], d["counts"], d["medians"], d["edges"]])
    off, n_final = coreg.nuth_kaab(ref, tba, inl, (10.0, 10.0))
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"nkb{r}.npz"))
        assert np.array_equal(g["step"], want_step, equal_nan=True), r
        assert np.allclose(g["fit"][:3], off, rtol=1e-9, atol=1e-9) and g["fit"][3] == n_final, (g["fit"], off)
    assert abs(off[1] - (-40.0)) < 2.0 or abs(off[1] - 40.0) < 2.0   # the 4-row shift is found (|northing| ~ 40 m)


def test_bench_two_ranks_reports_c4(tmp_path):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank) -- here with
    both ranks on GPU 0 over gloo (XDEM_BENCH_SHARE_GPU) and small rasters: one JSON line, n_gpus 2, and the C4 row-block
    measurement under "secondary" (65536^2 in production; XDEM_BENCH_C4_SIZE shrinks it for this test)."""
    import json
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XDEM_BENCH_SHARE_GPU="1", XDEM_BENCH_C4_SIZE="4096", XDEM_BENCH_C3_SIZE="3000", XDEM_BENCH_C5_RUNS="6",
               MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "3000"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["scaling"] == "strong" and res["value"] > 0
    c4 = res["secondary"]["c4_terrain_row_blocks"]
    assert c4["n_gpus"] == 2 and c4["value"] > 0 and "4096x4096" in c4["workload"]
    # the other two paths report with more than one rank too: pair blocks dealt to the ranks (histograms / counters all-reduced),
    # the Nuth-Kaab pair partitioned by row block -- both validated inside bench.py (equal class counts; recovered shift)
    assert "error" not in res["secondary"], res["secondary"]
    v, nk = res["secondary"]["variogram"], res["secondary"]["nuthkaab"]
    assert v["n_gpus"] == 2 and v["matheron_pass_Gpairs_s"] > 0 and v["dowd_exact_median_Gpairs_s"] > 0 and "validated" in v
    assert nk["n_gpus"] == 2 and "row blocks of 2 ranks" in nk["partition"] and abs(nk["fitted_shift_px"][0] - 1.7) < 0.05


def test_bench_two_ranks_over_rccl_one_gpu_each(tmp_path):
    """The run the driver makes on its 8-GPU node, in small: `bench.py --gpus 2` under torch.distributed.run with ONE RANK PER GPU --
    `init_process_group("nccl")`, `batch_isend_irecv` halo rows between two devices, the device-side reduction hook with a real peer,
    the partitioned Nuth-Kaab fit and the sharded variogram -- so that the first multi-GPU run is not also the first execution of
    those code paths.  Needs two GPUs: on the one-GPU test boxes of this pool it is an expected failure (the reason says so)."""
    import json
    import socket
    import subprocess

    if torch.cuda.device_count() < 2:
        pytest.xfail(f"needs >= 2 GPUs for one RCCL rank per GPU; this box shows {torch.cuda.device_count()} "
                     "(the shared-GPU gloo form of the same run is test_bench_two_ranks_reports_c4)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XDEM_BENCH_C4_SIZE="8192", XDEM_BENCH_C3_SIZE="6000", XDEM_BENCH_C5_RUNS="6", MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("XDEM_BENCH_SHARE_GPU", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "8192"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=560)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["rccl_ranks"] == 2 and "RCCL send/recv" in res["config"]["partition"]
    assert res["config"]["halo_bytes_per_step_and_rank"] > 0 and res["value"] > 0
    sec = res["secondary"]
    assert "error" not in sec, sec
    nk = sec["nuthkaab"]
    assert nk["n_gpus"] == 2 and abs(nk["fitted_shift_px"][0] - 1.7) < 0.05 and nk["reductions"]["through_the_host"] == 0, nk
    assert sec["variogram"]["n_gpus"] == 2 and sec["variogram"]["reductions"]["device"] > 0
    assert sec["c4_terrain_row_blocks"]["n_gpus"] == 2


def test_rccl_branch_of_the_halo_exchange_on_a_one_rank_group():
    """The branch of ``RowBlock.exchange`` the 8-GPU run takes -- ``batch_isend_irecv`` of DEVICE row slices under the nccl (=
    RCCL) backend, grouped ncclSend / ncclRecv on the communicator's stream -- driven on the one GPU of the test box: a block
    with the geometry of a MIDDLE rank (halo rows on both sides) whose two neighbours are this very process in a 1-rank RCCL
    group.  Self-addressed send / recv pairs of one group call match in issue order, so the halo above must receive the
    block's own first rows and the halo below its last ones.  Checked: the halo contents, the stream ordering of the overlapped
    form (interior launch, exchange, boundary launches behind it -- on a side stream, with the interior rewritten just before),
    and the planes against the raster the exchange is equivalent to."""
    import torch.distributed as dist

    from xdem_amd import _lib
    from xdem_amd import dist as xd
    from xdem_amd.synth import fbm_torch
    from xdem_amd.terrain import terrain_attributes_device

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        assert dist.get_backend() == "nccl"
        dev = torch.device("cuda", 0)
        ctx = _lib.default_context(0)
        n, W, depth = 1537, 2100, xd.halo_depth(ATTRS, "Florinsky", 3)
        block = xd.RowBlock(n, W, depth, 1, 3, dev)            # rows [512, 1024) of a 3-rank partition
        assert block.halo_top == depth and block.halo_bottom == depth
        block.peer_up = block.peer_down = 0                     # both neighbours: this process
        block.buf.fill_(float("nan"))
        rows = fbm_torch(block.rows, W, dev, seed=42, row0=block.r0, total_rows=n)
        block.interior.copy_(rows)
        works = block.exchange()
        assert len(works) > 0   # (the gloo branch stages through the host and returns a landing object: not this one)
        xd.RowBlock.wait_all(works)
        torch.cuda.synchronize()
        assert torch.equal(block.buf[:depth].view(torch.int32), rows[:depth].view(torch.int32))
        assert torch.equal(block.buf[-depth:].view(torch.int32), rows[-depth:].view(torch.int32))
        # overlapped form on a side stream: the interior is rewritten right before, so a boundary launch that ran ahead of the
        # exchange (or an exchange that ran ahead of the copy) would read stale / NaN halo rows
        want_buf = torch.cat([rows[:depth] + 1.0, rows + 1.0, rows[-depth:] + 1.0])
        want = terrain_attributes_device(want_buf, ATTRS, resolution=10.0, halo_top=depth, halo_bottom=depth, ctx=ctx)
        torch.cuda.synchronize()
        s = torch.cuda.Stream(dev)
        for overlap in (True, False):
            block.buf.fill_(float("nan"))
            torch.cuda.synchronize()
            with torch.cuda.stream(s):
                block.interior.copy_(rows + 1.0)
                out = xd.terrain_row_block(block, ATTRS, overlap=overlap, resolution=10.0, surface_fit="Florinsky",
                                           curv_method="geometric", ctx=ctx)
            s.synchronize()
            assert torch.equal(out.view(torch.int32), want.view(torch.int32)), overlap
        # the partitioned Nuth-Kaab plan (xdemhip_nk_create_block: row block + halo rows) with the DEVICE-side reduction hook on
        # the same 1-rank RCCL group: every reduction of the fit enqueued through RCCL on the library's stream, nothing staged
        from xdem_amd import coreg
        import scipy.optimize

        m = 2304
        ref = fbm_torch(m, m, dev, seed=21)
        tba = torch.roll(ref, shifts=(1, -2), dims=(0, 1)) + 1.0 + 0.3 * torch.randn((m, m), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
        tba[torch.rand((m, m), device=dev, generator=torch.Generator(device=dev).manual_seed(6)) < 0.1] = float("nan")
        torch.cuda.synchronize()
        ctx.set_option("selection", 3)
        try:
            base = coreg.NKPlan(ref, tba, None, ctx)
            want_off = coreg._iterate(base, (10.0, 10.0), 0.0, 3, 72, scipy.optimize.curve_fit, True)
            base.close()
            h0, d0 = ctx.reduction_calls()
            plan = coreg.NKPlan(ref, tba, None, ctx, "world", block=(m, 0, m, 0, 0))
            got_off = coreg._iterate(plan, (10.0, 10.0), 0.0, 3, 72, scipy.optimize.curve_fit, True)
            plan.close()
            h1, d1 = ctx.reduction_calls()
        finally:
            ctx.set_option("selection", 0)
        assert d1 - d0 >= 15 and h1 == h0, (h0, h1, d0, d1)
        # (the single-rank reference runs the one-pass step, whose nanmean / nanstd -- the p0 of the curve fit -- carry float32
        # partial sums: the fitted offsets agree to the optimiser's tolerance, not to the last bits)
        assert np.allclose(got_off, want_off, rtol=1e-6, atol=1e-6), (got_off, want_off)
    finally:
        if created:
            dist.destroy_process_group()


NB1P = 18   # aspect bins of the one-pass test: 6000^2 / 18 = the pixels per bin of a 12000^2 pair with 72 (narrower samples give brackets too wide for the route)


def _worker_nk_onepass(rank, world, port, outdir, rule):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xdem_amd import _lib, coreg
        from xdem_amd import dist as xd

        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        ctx = _lib.Context(0)
        if rule is not None:
            ctx.set_option("nk_nan_rule", rule)
        ref = np.load(os.path.join(outdir, "ref.npy"), mmap_mode="r")
        tba = np.load(os.path.join(outdir, "tba.npy"), mmap_mode="r")
        H, W = ref.shape
        r0, r1 = xd.row_block(H, world, rank)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a[r0:r1])).to(dev)
        rb = xd.RowBlock(H, W, 6, rank, world, dev)
        tb = xd.RowBlock(H, W, 6, rank, world, dev)
        for b, a in ((rb, ref), (tb, tba)):
            b.interior.copy_(to(a))
            xd.RowBlock.wait_all(b.exchange())
        plan = coreg.NKPlan(rb.buf, tb.buf, None, ctx, "world", block=(H, r0, r1, rb.halo_top, rb.halo_bottom))
        steps = [tuple(s) for s in np.load(os.path.join(outdir, "steps.npy"))]
        plan.step(0.3, 0.1, (10.0, 10.0), NB1P)   # (route agreement, buffers, bin cache)
        h0, d0 = ctx.reduction_calls()
        c0 = plan.route_counts()
        out = [plan.step(sx, sy, (10.0, 10.0), NB1P) for (sx, sy) in steps]
        h1, d1 = ctx.reduction_calls()
        c1 = plan.route_counts()
        plan.close()
        info = {}
        off, n_final = xd.nuth_kaab_row_blocks(to(ref), to(tba), H, (10.0, 10.0), halo=6, ctx=ctx, tolerance=0.0, max_iterations=6, bin_sizes=NB1P, info=info)
        np.savez(os.path.join(outdir, f"nk1p{rank}.npz"),
                 steps=np.array([np.concatenate([[d["vshift"], d["n_valid"], d["y_mean"], d["y_std"]], d["counts"], d["medians"], d["edges"]]) for d in out]),
                 routes=np.array([c1[k] - c0[k] for k in ("onepass", "plain")]), reductions=np.array([h1 - h0, d1 - d0]),
                 predicted=np.array([c1["predicted"] - c0["predicted"], c1["predict_missed"] - c0["predict_missed"], c1["predicted_dh_only"] - c0["predicted_dh_only"]]),
                 fit=np.array([*off, n_final], dtype=np.float64), fit_routes=np.array([info["routes"][k] for k in ("onepass", "plain")]))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rule", [None, 3])
def test_nuth_kaab_partitioned_one_pass_step(tmp_path, rule):
    """Round 5 (the review's task 4): PARTITIONED plans take the one-pass step.  Two ranks hold a row block + halo rows
    each of a 6000^2 pair built like bench.py's C3 (18 aspect bins); every step is answered by the one-pass route with exactly TEN reductions
    through the hook, and vshift / counts / medians / edges are the single-process plan's bit for bit -- fractional steps and the
    aligned pair (dh collapses onto a few float32 values) included; so is the whole fit through `nuth_kaab_row_blocks`.
    rule = 3: the same under the dilating nodata rule (bad-bit mask instantiation of the pass)."""
    sys.path.insert(0, ROOT)
    import scipy.optimize

    import bench
    from xdem_amd import _lib, coreg

    dev = torch.device("cuda", 0)
    m = 6000
    ref, tba = bench._c3_pair(dev, m)
    np.save(os.path.join(str(tmp_path), "ref.npy"), ref.cpu().numpy())
    np.save(os.path.join(str(tmp_path), "tba.npy"), tba.cpu().numpy())
    steps = np.array([(0.0, 0.0), (1.7, 0.6), (1.2, 0.9), (-17.0, -6.0), (-16.9998, -5.9996)])
    np.save(os.path.join(str(tmp_path), "steps.npy"), steps)
    world = 2
    mpc = mp.get_context("spawn")
    port = 29800 + (os.getpid() % 90)
    procs = [mpc.Process(target=_worker_nk_onepass, args=(r, world, port, str(tmp_path), rule)) for r in range(world)]
    for p in procs:
        p.start()
    # the single-process answers meanwhile
    ctx = _lib.Context(0)
    try:
        if rule is not None:
            ctx.set_option("nk_nan_rule", rule)
        plan = coreg.NKPlan(ref, tba, None, ctx)
        plan.step(0.3, 0.1, (10.0, 10.0), NB1P)
        want = [plan.step(float(sx), float(sy), (10.0, 10.0), NB1P) for (sx, sy) in steps]
        assert plan.route_counts()["onepass"] == 1 + len(steps), plan.route_counts()
        plan.close()
        plan = coreg.NKPlan(ref, tba, None, ctx)
        off = coreg._iterate(plan, (10.0, 10.0), 0.0, 6, NB1P, scipy.optimize.curve_fit, True)
        n_final = plan.n_valid
        plan.close()
    finally:
        ctx.close()
    for p in procs:
        p.join(timeout=200)
    hung = [p for p in procs if p.exitcode is None]
    for p in hung:
        p.kill()
    assert not hung and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), f"nk1p{r}.npz"))
        assert tuple(g["routes"]) == (len(steps), 0), (r, g["routes"])
        # (gloo group: every reduction staged through the host hook.)  Ten per step with sampled brackets, five with predicted ones -- round 6:
        # the last step moves the aligned pair by 4e-5 px and takes its brackets from the step before it, on every rank alike
        n_pred, n_miss, n_pd = (int(v) for v in g["predicted"])
        assert n_pred >= 1 and n_miss == 0, (r, g["predicted"])
        assert tuple(g["reductions"]) == (10 * (len(steps) - n_pred - n_pd) + 8 * n_pd + 5 * n_pred, 0), (r, g["reductions"], n_pred, n_pd)
        assert g["fit_routes"][1] == 0 and g["fit_routes"][0] == 6, (r, g["fit_routes"])
        for row, d in zip(g["steps"], want):
            exact = np.concatenate([[d["vshift"], d["n_valid"]], d["counts"], d["medians"], d["edges"]])
            assert np.array_equal(np.concatenate([row[:2], row[4:]]), exact, equal_nan=True), r
            # nanmean / nanstd of y: float32 partial sums per chunk of rows -- the chunks of two row blocks are not those of the whole raster
            assert abs(row[2] - d["y_mean"]) <= 2e-6 * abs(d["y_std"]) and abs(row[3] - d["y_std"]) <= 2e-6 * abs(d["y_std"]), (r, row[2:4], d["y_mean"], d["y_std"])
        assert np.allclose(g["fit"][:3], off, rtol=0, atol=1e-6) and g["fit"][3] == n_final, (g["fit"], off)
    g0, g1 = (np.load(os.path.join(str(tmp_path), f"nk1p{r}.npz")) for r in range(world))
    assert np.array_equal(g0["steps"], g1["steps"], equal_nan=True) and np.array_equal(g0["fit"], g1["fit"])   # both ranks: the same bits, moments included
