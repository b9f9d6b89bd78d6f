"""Partitioned Nuth-Kaab step, one-pass step vs plain route, on ranks that SHARE one GPU (gloo group, reductions staged through the
host hook -- an upper bound of what RCCL ranks on their own GPUs pay per reduction).  Every rank builds bench.py's C3 pair, keeps its
row block + halo, and times steps of a partitioned plan under option "nk_fused_dist" = 1 (one data pass, 5-10 all-reduces) and 0 (the plain
route: stored dh, a reduction per key digit); rank 0 also times the hook-less plan on the whole pair while the others wait.

    python tools/nk_dist_probe.py [size=20000] [world=2] [steps=5]
"""
import os
import sys
import time

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, m, k):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from xdem_amd import _lib, coreg
        from xdem_amd import dist as xd

        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        ctx = _lib.Context(0)
        ref, tba = bench._c3_pair(dev, m)
        steps = [(3.0 + i, -4.0) for i in range(k)]
        single = None
        if rank == 0:
            plan = coreg.NKPlan(ref, tba, None, ctx)
            plan.step(0.0, 0.0, (10.0, 10.0), 72)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            single = [plan.step(sx, sy, (10.0, 10.0), 72) for (sx, sy) in steps]
            t_single = (time.perf_counter() - t0) / k
            print(f"[{m}^2] one process, whole pair: {t_single * 1e3:.2f} ms per step, routes {plan.route_counts()}", flush=True)
            plan.close()
        dist.barrier()
        halo = 8
        r0, r1 = xd.row_block(m, world, rank)
        ht, hb = (halo if rank > 0 else 0), (halo if rank < world - 1 else 0)
        rbuf = ref[r0 - ht:r1 + hb].contiguous()
        tbuf = tba[r0 - ht:r1 + hb].contiguous()
        del ref, tba
        torch.cuda.empty_cache()
        for fused_dist in (1, 0):
            ctx.set_option("nk_fused_dist", fused_dist)
            plan = coreg.NKPlan(rbuf, tbuf, None, ctx, "world", block=(m, r0, r1, ht, hb))
            plan.step(0.0, 0.0, (10.0, 10.0), 72)
            c0 = plan.route_counts()
            h0, d0 = ctx.reduction_calls()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            out = [plan.step(sx, sy, (10.0, 10.0), 72) for (sx, sy) in steps]
            dist.barrier()
            dt = (time.perf_counter() - t0) / k
            h1, d1 = ctx.reduction_calls()
            c1 = plan.route_counts()
            plan.close()
            if rank == 0:
                same = all(a["vshift"] == b["vshift"] and np.array_equal(a["medians"], b["medians"], equal_nan=True) and np.array_equal(a["counts"], b["counts"])
                           for a, b in zip(out, single))
                print(f"[{m}^2] {world} ranks on one GPU, nk_fused_dist={fused_dist}: {dt * 1e3:.2f} ms per step, "
                      f"{(h1 - h0 + d1 - d0) / k:.1f} all-reduces per step, routes {({q: c1[q] - c0[q] for q in c1})}, identical to one process: {same}", flush=True)
        ctx.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    port = 29700 + (os.getpid() % 90)
    ctxm = mp.get_context("spawn")
    procs = [ctxm.Process(target=worker, args=(r, world, port, m, k)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    bad = [p for p in procs if p.exitcode != 0]
    for p in procs:
        if p.exitcode is None:
            p.kill()
    sys.exit(1 if bad else 0)
