"""REHEARSAL transport of the final gather: the plan's parts / the rows as HOST tensors through `torch.distributed`
point-to-point messages (gloo), in the layout of the RCCL path (`uav_ac.comm`, which shares `sharding.gather_layout` with it
and nothing else).  It exists so that the N > 1 control flow -- uneven blocks, the root re-sampling its peers' plans -- can be
walked where no second GPU exists: the multi-process CPU tests (tests/test_distributed_gloo.py) and `UAVAC_BENCH_REHEARSAL=1`.
It validates layout logic, NOT the RCCL code; the product never imports this module."""
from __future__ import annotations

import numpy as np

from .sharding import gather_layout


def _torch():
    import torch
    return torch


def gather_plan(plan, dst: int = 0, group=None, comm: "RcclComm" = None, engine: "Engine" = None, parts=None):
    """`RcclComm.gather_plan` with the host rehearsal path beside it (like `gather_rows`).

    A device-resident Plan + `comm`: RCCL behind the C ABI.  Otherwise the plan's parts travel as HOST tensors through
    `torch.distributed` point-to-point messages (gloo) in the same layout -- the rehearsal path of the multi-process CPU
    tests and of `UAVAC_BENCH_REHEARSAL`; on dst the result is a Plan re-sampled on `engine`'s GPU when one is given, else
    the gathered parts `{"coeffs", "times", "seg_rows", "m"}` as host tensors.  Returns (result | None, row counts).
    `parts` (True or cumulative shares, with `engine`): re-sample in the ORDER of the pipelined RCCL gather -- part p of every
    rank's block, part after part (`Engine.sample_range`) -- nothing is pipelined over gloo, the rows must not care."""
    torch = _torch()
    if comm is not None and getattr(plan.coeffs, "is_cuda", False):
        return comm.gather_plan(plan, dst, parts=parts)
    import torch.distributed as dist
    if not hasattr(plan, "m"):
        raise ValueError("the host (gloo) path of gather_plan takes a Plan with one segment count for the batch; a ragged "
                         "batch travels over RCCL only (RcclComm.gather_plan)")
    m = int(plan.m)
    host = lambda t, dt_: torch.as_tensor(np.asarray(t.cpu() if hasattr(t, "cpu") else t)).to(dt_).contiguous()   # noqa: E731
    co, seg_counts = gather_rows(host(plan.coeffs, torch.float64).reshape(-1, 24), dst, group)
    tm = None
    if plan.times is not None:
        tm, _ = gather_rows(host(plan.times, torch.float64).reshape(-1, 1), dst, group)
    sr, _ = gather_rows(host(plan.seg_rows, torch.int32).reshape(-1, 1), dst, group)
    n = torch.tensor([int(plan.total_rows)], dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(dist.get_world_size(group))]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    if dist.get_rank(group) != dst:
        return None, counts
    if engine is not None and parts:
        from .sharding import PIPELINE_SHARES, part_bounds
        shares = PIPELINE_SHARES if parts is True else tuple(parts)
        missions = [c // m for c in seg_counts]              # (no further collective here: the peers have left already)
        got = engine.plan_from_parts(co.reshape(-1, 8 * m, 3), None if tm is None else tm.reshape(-1, m), sr.reshape(-1, m), m,
                                     plan.velocity, plan.dt, total_rows=sum(counts), sample=False)
        base = np.concatenate([[0], np.cumsum(missions)])
        bounds = [part_bounds(b, shares) for b in missions]
        for p in range(len(shares)):
            for r in range(len(missions)):
                engine.sample_range(got, base[r] + bounds[r][p], base[r] + bounds[r][p + 1])
        return got, counts
    if engine is not None:
        return engine.plan_from_parts(co.reshape(-1, 8 * m, 3), None if tm is None else tm.reshape(-1, m), sr.reshape(-1, m), m,
                                      plan.velocity, plan.dt, total_rows=sum(counts)), counts
    return {"coeffs": co.reshape(-1, 8 * m, 3), "times": None if tm is None else tm.reshape(-1, m), "seg_rows": sr.reshape(-1, m),
            "m": m}, counts


def gather_rows(rows, dst: int = 0, group=None, max_message_bytes: int = 1 << 30, comm: "RcclComm" = None):
    """Gather ragged (n_r, C) row blocks to `dst` with point-to-point transfers (one direct xGMI link per
    peer into the root; a ring all-gather would push 7/8 of the total through every link).

    GPU tensors travel over RCCL behind the C ABI (`comm`: an RcclComm; `uavac_gather_rows_dev` =
    ncclGroupStart + ncclSend / ncclRecv + ncclGroupEnd).  HOST tensors take the same layout through
    `torch.distributed` point-to-point messages (gloo): the rehearsal path of the multi-process CPU tests, where no
    GPU exists.  Returns (all_rows, counts) on dst and (None, counts) elsewhere.
    """
    torch = _torch()
    if rows.is_cuda:
        if comm is None:
            raise ValueError("GPU rows are gathered over RCCL: pass comm=RcclComm(engine)")
        return comm.gather_rows(rows, dst)
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    if world == 1:
        return rows, counts
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    row_bytes = max(1, rows.element_size() * int(np.prod(rows.shape[1:], dtype=np.int64)))
    step = max(1, int(max_message_bytes) // row_bytes)             # rows per message, same on both ends

    def pieces(count):
        return [(a, min(a + step, count)) for a in range(0, count, step)]

    offs, senders = gather_layout(counts, dst)
    if rank == dst:
        out = torch.empty((int(offs[-1]),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        out[offs[dst]:offs[dst + 1]].copy_(rows)
        ops = [dist.P2POp(dist.irecv, out[offs[r] + a:offs[r] + b], peer(r), group)
               for r in senders for a, b in pieces(counts[r])]
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        return out, counts
    if rank in senders:
        mine = rows.contiguous()
        ops = [dist.P2POp(dist.isend, mine[a:b], peer(dst), group) for a, b in pieces(counts[rank])]
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return None, counts
