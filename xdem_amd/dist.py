"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

Terrain / Nuth-Kaab rasters are split into contiguous ROW BLOCKS -- the GPU analogue of the reference's
tile-with-overlap multiprocessing (``map_overlap_multiproc_save(depth=...)``, xdem/terrain/terrain.py:412-462).
Each rank holds its block plus ``depth`` halo rows per neighbour, refreshed by one grouped
send/recv pair per neighbour (point-to-point over xGMI; <= 2 rows x 65536 px x 4 B = 512 KiB, latency-bound).
Accumulator-style paths (variogram bins, Nuth-Kaab histograms) shard their work list and all-reduce small
integer / float64 arrays.  Everything here also runs on CPU tensors with the "gloo" backend (tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def halo_depth(attribute: list[str], surface_fit: str = "Florinsky", window_size: int = 3) -> int:
    """Overlap rows needed per neighbour: same rule as xdem/terrain/terrain.py:417-432."""
    from .terrain import list_requiring_surface_fit, list_requiring_windowed_index

    window_depth = window_size // 2 if set(attribute) & set(list_requiring_windowed_index) else 0
    if any(a in list_requiring_surface_fit for a in attribute):
        surface_fit_depth = 2 if surface_fit.lower() == "florinsky" else 1
    else:
        surface_fit_depth = 0
    return max(window_depth, surface_fit_depth)


def row_block(total_rows: int, world: int, rank: int) -> tuple[int, int]:
    """[r0, r1) of `rank` in a balanced contiguous partition of `total_rows` rows over `world` ranks."""
    base, rem = divmod(total_rows, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


class RowBlock:
    """A rank's row block with halo storage: ``buf`` has halo_top + rows + halo_bottom rows, where the halo is
    `depth` rows towards each existing neighbour (none at the raster's first / last block)."""

    def __init__(self, total_rows: int, width: int, depth: int, rank: int, world: int, device, dtype=torch.float32):
        self.rank, self.world, self.depth = rank, world, depth
        self.r0, self.r1 = row_block(total_rows, world, rank)
        self.rows = self.r1 - self.r0
        if world > 1 and self.rows < depth:
            raise ValueError(f"row block of {self.rows} rows is thinner than the halo depth {depth}")
        self.halo_top = depth if rank > 0 else 0
        self.halo_bottom = depth if rank < world - 1 else 0
        # ranks (of the exchange's group) that hold the rows above / below; a test may point both at this very process to drive
        # the RCCL branch of `exchange` on a one-GPU box (self-addressed ncclSend / ncclRecv pairs match in issue order)
        self.peer_up, self.peer_down = rank - 1, rank + 1
        self.buf = torch.empty((self.halo_top + self.rows + self.halo_bottom, width), device=device, dtype=dtype)

    @property
    def interior(self) -> torch.Tensor:
        return self.buf[self.halo_top : self.halo_top + self.rows]

    def exchange(self, group=None) -> list:
        """Refresh the halo rows from the neighbours: one batched isend/irecv group (ncclSend/ncclRecv grouped
        under RCCL).  Returns the work handles; call ``wait_all`` before launching kernels that read the halo.

        Stream ordering under the nccl (= RCCL) backend, by torch's ProcessGroupNCCL rules: the grouped send / recv is
        enqueued on the communicator's own stream after an event of the CURRENT torch stream (so it reads interior rows
        that earlier work on the current stream -- the synthesis copy, the previous step's kernels -- has finished with,
        and never overwrites halo rows a previous boundary launch still reads); ``work.wait()`` makes the current
        stream wait for the exchange without blocking the host.  The library launches every device-resident call on the
        current torch stream (``ctx.set_stream(torch.cuda.current_stream(...))`` in terrain_attributes_device), so the
        interior launch issued between ``exchange`` and ``wait_all`` overlaps the transfer -- it reads only this rank's own
        rows and writes only ``out`` -- and the boundary launches issued after ``wait_all`` are ordered behind it.  The send
        slices are row ranges of the contiguous block buffer (``contiguous()`` returns views, nothing is staged)."""
        if self.world == 1:
            return []
        if self.buf.is_cuda and dist.get_backend(group) == "gloo":
            return self._exchange_staged(group)
        d, ops = self.depth, []
        if self.halo_top:
            ops.append(dist.P2POp(dist.isend, self.interior[:d].contiguous(), self.peer_up, group))
            ops.append(dist.P2POp(dist.irecv, self.buf[: self.halo_top], self.peer_up, group))
        if self.halo_bottom:
            ops.append(dist.P2POp(dist.isend, self.interior[-d:].contiguous(), self.peer_down, group))
            ops.append(dist.P2POp(dist.irecv, self.buf[self.halo_top + self.rows :], self.peer_down, group))
        return dist.batch_isend_irecv(ops)

    def _exchange_staged(self, group=None) -> list:
        """Same exchange for device buffers under the gloo backend (no device-side point-to-point there): halo rows are
        staged through host tensors.  Lets several ranks share one GPU, which RCCL refuses -- used to test the multi-rank
        path on a single-GPU box; production runs use RCCL."""
        d = self.depth
        up, down = self.peer_up, self.peer_down
        ops, landings = [], []
        if self.halo_top:
            ops.append(dist.P2POp(dist.isend, self.interior[:d].cpu(), up, group))
            r = torch.empty((self.halo_top, self.buf.shape[1]), dtype=self.buf.dtype)
            ops.append(dist.P2POp(dist.irecv, r, up, group))
            landings.append((r, self.buf[: self.halo_top]))
        if self.halo_bottom:
            ops.append(dist.P2POp(dist.isend, self.interior[-d:].cpu(), down, group))
            r = torch.empty((self.halo_bottom, self.buf.shape[1]), dtype=self.buf.dtype)
            ops.append(dist.P2POp(dist.irecv, r, down, group))
            landings.append((r, self.buf[self.halo_top + self.rows :]))
        works = dist.batch_isend_irecv(ops)

        class _Landing:
            def wait(self_inner):
                for w in works:
                    w.wait()
                for src, dst in landings:
                    dst.copy_(src)

        return [_Landing()]

    @staticmethod
    def wait_all(works: list) -> None:
        for w in works:
            w.wait()


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over ranks (bin accumulators: int64 counts / float64 sums / uint64-as-int64 histograms)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def terrain_row_block(block: RowBlock, attribute: list[str], out: torch.Tensor | None = None, overlap: bool = True,
                      group=None, **kw) -> torch.Tensor:
    """All requested attributes for this rank's rows: halo exchange + fused kernel.

    With ``overlap`` the interior rows (which need no neighbour data) are launched first, the halo exchange
    proceeds on RCCL's stream meanwhile, and the two `depth`-row boundary strips are launched once it lands.
    """
    from .terrain import terrain_attributes_device

    n, d = len(attribute), block.depth
    if out is None:
        out = torch.empty((n, block.rows, block.buf.shape[1]), device=block.buf.device, dtype=block.buf.dtype)
    works = block.exchange(group)
    ht, hb, rows = block.halo_top, block.halo_bottom, block.rows
    if not works or not overlap or rows <= 4 * d:
        RowBlock.wait_all(works)
        terrain_attributes_device(block.buf, attribute, out=out, halo_top=ht, halo_bottom=hb, **kw)
        return out
    # interior: output rows [top_n, rows - bot_n) only read this rank's own rows
    top_n = d if ht else 0
    bot_n = d if hb else 0
    b0 = ht  # the interior launch starts at this rank's first own row: output row top_n reads own rows from top_n - d = 0 on
    inner_rows = rows - top_n - bot_n
    inner = block.buf[b0 : ht + rows - bot_n + (d if bot_n else 0)]
    terrain_attributes_device(inner, attribute, out=_rows(out, top_n, inner_rows), halo_top=(d if top_n else 0),
                              halo_bottom=(d if bot_n else 0), **kw)
    RowBlock.wait_all(works)
    if top_n:
        terrain_attributes_device(block.buf[: ht + top_n + d], attribute, out=_rows(out, 0, top_n), halo_top=ht,
                                  halo_bottom=d, **kw)
    if bot_n:
        terrain_attributes_device(block.buf[ht + rows - bot_n - d :], attribute, out=_rows(out, rows - bot_n, bot_n),
                                  halo_top=d, halo_bottom=hb, **kw)
    return out


def _rows(out: torch.Tensor, r0: int, n: int):
    return out[:, r0 : r0 + n, :]


def nuth_kaab_row_blocks(ref_rows: torch.Tensor, tba_rows: torch.Tensor, total_rows: int, res: tuple[float, float],
                         inlier_rows: torch.Tensor | None = None, group=None, halo: int = 8, ctx=None, tolerance: float = 0.001,
                         max_iterations: int = 10, bin_sizes=72, fit_optimizer=None, bin_statistic=None, bin_before_fit: bool = True,
                         initial_offsets: tuple[float, float] = (0.0, 0.0), info: dict | None = None):
    """Nuth & Kaab fit of a raster pair PARTITIONED by row block over the ranks of ``group``: every rank passes only ITS
    rows -- ``row_block(total_rows, world, rank)`` -- of the reference / to-be-aligned DEM (device tensors, same dtype) and
    of the optional inlier mask (uint8).  Layout and halo exchange as for the terrain path (structural
    ancestor: xdem/coreg/blockwise.py:174): each rank's buffers hold ``halo`` rows of both neighbours, copied once per fit
    with one grouped send / recv pair per neighbour (``RowBlock.exchange``: ncclSend / ncclRecv under RCCL).  One halo row
    serves ``np.gradient``, the rest bounds the vertical shift the iteration may reach (``floor(|shift_y| / res_y) + 2``
    rows); if a step leaves it (``HaloTooSmall``, raised identically on every rank) the halo is doubled and the fit
    restarts.  Every reduction of a step -- integer histograms, counters, min / max keys -- is all-reduced through the
    library hook, so all ranks obtain the same, exact result as a single-GPU fit of the whole rasters.  Large blocks take the
    one-pass step (one data pass over the rank's rows, ten all-reduces per sampled step, five per predicted one: include/xdemhip.h, xdemhip_nk_route_counts).

    ``info`` (optional dict) receives ``routes`` -- how the steps of the fit were answered on this rank (one-pass /
    plain; identical on every rank) -- and ``reductions``, the (host-staged, device-side) reductions the fit made.

    Returns ((easting, northing, vertical) offsets, number of valid pixels of the whole pair)."""
    import numpy as np
    import scipy.optimize

    from . import _lib, coreg

    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    W = ref_rows.shape[1]
    r0, r1 = row_block(total_rows, world, rank)
    if tuple(ref_rows.shape) != (r1 - r0, W) or tba_rows.shape != ref_rows.shape:
        raise ValueError(f"rank {rank} must pass rows [{r0}, {r1}) of both rasters")
    ctx = ctx or _lib.default_context(ref_rows.device.index or 0)
    fit_optimizer = fit_optimizer or scipy.optimize.curve_fit
    bin_statistic = bin_statistic if bin_statistic is not None else np.nanmedian
    while True:
        depth = min(halo, (total_rows // world) if world > 1 else halo)
        blocks = []
        for t, dt in ((ref_rows, ref_rows.dtype), (tba_rows, tba_rows.dtype), (inlier_rows, torch.uint8)):
            if t is None:
                blocks.append(None)
                continue
            b = RowBlock(total_rows, W, depth, rank, world, ref_rows.device, dtype=dt)
            b.interior.copy_(t)
            RowBlock.wait_all(b.exchange(group))
            blocks.append(b)
        rb, tb, ib = blocks
        plan = coreg.NKPlan(rb.buf, tb.buf, ib.buf if ib is not None else None, ctx,
                            (group if group is not None else "world") if world > 1 else None,
                            block=(total_rows, r0, r1, rb.halo_top, rb.halo_bottom))
        try:
            if plan.n_valid == 0:
                raise ValueError(
                    "There is no valid points common to the input and auxiliary data (bias variables, or "
                    "derivatives required for this method, for example slope, aspect, etc)."
                )
            plan.set_statistic(bin_statistic)
            if not isinstance(bin_sizes, (int, np.integer)):
                plan.set_bin_edges(bin_sizes)
            red0 = ctx.reduction_calls()
            offsets = coreg._iterate(plan, res, tolerance, max_iterations, bin_sizes, fit_optimizer, bin_before_fit, initial_offsets)
            if info is not None:
                red1 = ctx.reduction_calls()
                info["routes"] = plan.route_counts()
                info["reductions"] = (red1[0] - red0[0], red1[1] - red0[1])
            return offsets, plan.n_valid
        except coreg.HaloTooSmall:
            if world == 1 or depth >= total_rows // world:
                raise
            halo = 2 * depth
        finally:
            plan.close()
