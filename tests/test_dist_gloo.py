"""Multi-process CPU tests (gloo, world_size 2) of the N > 1 path: halo exchange of the row-block partition and the
all-reduce helpers of the variogram path.  The HIP kernels themselves need a GPU; what is checked here is that every
rank ends up with exactly the neighbour rows / combined accumulators the single-process computation would have."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from xdem_amd import dist as xd
        from xdem_amd import spatialstats as ss

        total, width, depth = 37, 11, 2
        full = torch.arange(total * width, dtype=torch.float32).reshape(total, width)
        blk = xd.RowBlock(total, width, depth, rank, world, "cpu")
        blk.buf.fill_(float("nan"))
        blk.interior.copy_(full[blk.r0:blk.r1])
        xd.RowBlock.wait_all(blk.exchange())
        want = full[blk.r0 - blk.halo_top: blk.r1 + blk.halo_bottom]
        ok_halo = bool(torch.equal(blk.buf, want))
        # second exchange after the interior changed (bench.py refreshes halos every step)
        blk.interior.mul_(2.0)
        xd.RowBlock.wait_all(blk.exchange())
        ok_halo2 = bool(torch.equal(blk.buf, 2.0 * want))
        # uint8 blocks with a deeper halo: the inlier-mask blocks of the partitioned Nuth-Kaab fit (dist.nuth_kaab_row_blocks)
        full8 = (torch.arange(total * width) % 251).to(torch.uint8).reshape(total, width)
        b8 = xd.RowBlock(total, width, 5, rank, world, "cpu", dtype=torch.uint8)
        b8.buf.fill_(255)
        b8.interior.copy_(full8[b8.r0:b8.r1])
        xd.RowBlock.wait_all(b8.exchange())
        ok_halo2 = ok_halo2 and b8.buf.dtype == torch.uint8 and bool(torch.equal(b8.buf, full8[b8.r0 - b8.halo_top: b8.r1 + b8.halo_bottom]))
        # accumulator all-reduces used by the variogram / selection passes
        h = np.full((3, 256), rank + 1, dtype=np.uint64)
        s = ss._allreduce(h)
        c = ss._allreduce(np.array([1.5 * (rank + 1)]))
        m = ss._allreduce_min(np.array([10 + rank, 0xFFFFFFFFFFFFFFFF if rank == 0 else 7], dtype=np.uint64))
        ok_red = bool((s == sum(range(1, world + 1))).all()) and abs(c[0] - 1.5 * sum(range(1, world + 1))) < 1e-12 \
            and m.tolist() == [10, 7]
        # the library's reduction hook (xdemhip_set_allreduce) on raw 8-byte host arrays, full unsigned 64-bit key range:
        # keys of positive doubles have the top bit set, the all-ones marker means "none"
        import ctypes

        from xdem_amd import _lib

        hook = _lib.make_reduce_hook("world", None)

        def red(vals, kind, dtype=np.uint64):
            a = (ctypes.c_uint64 * len(vals))()
            np.frombuffer(a, dtype=dtype)[:] = np.array(vals, dtype=dtype)
            assert hook(ctypes.addressof(a), len(vals), kind, None) == 0
            return np.frombuffer(a, dtype=dtype).tolist()

        big = 0x8000000000000000
        ok_hook = red([5 + rank, big + 3], 0) == [sum(5 + r for r in range(world)), (world * (big + 3)) % 2**64]
        ok_hook &= red([0.25 * (rank + 1)], 1, np.float64) == [0.25 * sum(range(1, world + 1))]
        ok_hook &= red([big + 10 - rank, 7 + rank, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF if rank else big + 1], 2) == \
            [big + 10 - (world - 1), 7, 0xFFFFFFFFFFFFFFFF, big + 1]
        ok_hook &= red([big + rank, 3 + rank, (3 if rank else big + 2)], 3) == [big + world - 1, 3 + world - 1, big + 2]
        # a SUM all-reduce used as an ALL-GATHER (the one-pass Nuth-Kaab step on partitioned plans, xdemhip_set_rank): every rank fills
        # its own slot with arbitrary 8-byte patterns -- packed uint32 pairs, float64 bits (-0.0, NaN payloads, denormals, -Inf),
        # all-ones words -- and zeros elsewhere; what comes back must be every rank's slot, bit for bit
        def slot_words(r):
            f = np.array([-0.0, np.nan, 5e-324, -np.inf, 1.0 + r, -1.75e300 * (r + 1)], dtype=np.float64).view(np.uint64)
            u = np.array([0xFFFFFFFFFFFFFFFF, (0xFFFFFFFF << 32) | r, (r + 1) << 32, 0x80000000_80000000, 0x7FF8_0000_DEAD_0000 + r], dtype=np.uint64)
            return np.concatenate([f, u])
        width_s = len(slot_words(0))
        mine = np.zeros(world * width_s, dtype=np.uint64)
        mine[rank * width_s:(rank + 1) * width_s] = slot_words(rank)
        got = np.array(red(mine.tolist(), 0), dtype=np.uint64)
        ok_hook &= bool(np.array_equal(got, np.concatenate([slot_words(r) for r in range(world)])))
        ok_red = ok_red and ok_hook
        t = torch.tensor([float(rank)])
        xd.allreduce_sum_(t)
        q.put((rank, ok_halo, ok_halo2, ok_red, float(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_block_halo_exchange_and_allreduce(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_halo, ok_halo2, ok_red, tsum in res:
        assert ok_halo and ok_halo2, f"rank {rank}: halo rows differ from the full raster"
        assert ok_red, f"rank {rank}: all-reduce helpers wrong"
        assert tsum == sum(range(world))
