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
