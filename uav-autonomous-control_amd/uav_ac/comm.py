"""The final gather of the trajectories over RCCL, behind the C ABI (`uavac_comm_*`, `uavac_gather_*_dev`:
csrc/uavac_comm.hip).  The ONE exchange of the multi-GPU path; no reference counterpart."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat

_P = C.c_void_p


def _torch():
    import torch
    return torch


def _ptr(t) -> _P:
    return _P(0 if t is None else t.data_ptr())
from .engine import Engine, Plan


class RcclComm:
    """The communicator of the final gather: an ncclComm_t owned through the C ABI (`uavac_comm_*`, include/uavac.h).

    Bootstrap needs one side channel for the 128-byte unique id; here it is the already initialised
    `torch.distributed` process group (any backend), nothing else of torch takes part in the exchange."""

    def __init__(self, engine: Engine, group=None, unique_id: bytes = None, world: int = None, rank: int = None):
        torch = engine._torch
        self.engine = engine
        if unique_id is None:
            import torch.distributed as dist
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            box = [None]
            if rank == 0:
                buf = C.create_string_buffer(nat.COMM_ID_BYTES)
                engine.ctx.call("uavac_comm_unique_id", buf)
                box[0] = bytes(buf.raw)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            unique_id = box[0]
        if len(unique_id) != nat.COMM_ID_BYTES:
            raise ValueError("the RCCL unique id has 128 bytes")
        self.world, self.rank = int(world), int(rank)
        self._h = _P()
        engine._bind_stream()
        engine.ctx.call("uavac_comm_init_rank", C.create_string_buffer(unique_id, nat.COMM_ID_BYTES), self.world, self.rank,
                        C.byref(self._h))

    def shape(self):
        """(ranks, this rank) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        w, r = C.c_int(-1), C.c_int(-1)
        self.engine.ctx.call("uavac_comm_shape", self._h, C.byref(w), C.byref(r))
        return int(w.value), int(r.value)

    def counts(self, n_rows: int):
        out = (C.c_int64 * self.world)()
        self.engine._bind_stream()
        self.engine.ctx.call("uavac_gather_counts", self._h, int(n_rows), out)
        return [int(v) for v in out]

    def gather_rows(self, rows, dst: int = 0):
        """Ragged (n_r, C) f64 row blocks of all ranks -> (all_rows on dst | None, counts).  Synchronous."""
        return self.gather_finish(self.gather_rows_begin(rows, dst))

    def gather_rows_begin(self, rows, dst: int = 0, stream=None):
        """Enqueue the gather and return at once: (the trajectories are final when planning ends, so their gather can run
        beside the rollout instead of after it).  `stream`: a torch.cuda.Stream for the transfers; it first waits for
        what the current stream has enqueued so far (the kernels that produce `rows`).  Default: the current stream.
        Returns a ticket for `gather_finish`; `rows` must not be written before that.  The Engine's ctx is bound to the
        caller's stream again on return.  Between `*_begin` and `gather_finish` fly and log on this Engine as you like, but do
        NOT plan on it (`plan`, `replan`, `sample`, obstacle rounds): the root's re-sampling on the side stream and a planning
        call on yours would share the ctx's scratch arrays."""
        e, torch = self.engine, self.engine._torch
        if not rows.is_cuda or rows.dtype != torch.float64 or rows.dim() != 2:
            raise ValueError("rows must be a 2-D float64 GPU tensor")
        rows = rows.contiguous()
        here = torch.cuda.current_stream(e.device)
        stream = here if stream is None else stream
        if stream is not here:
            stream.wait_stream(here)
        with torch.cuda.stream(stream):
            counts = self.counts(rows.shape[0])              # a tiny synchronous all-gather on that stream
            out = None
            if self.rank == dst:
                out = torch.empty((sum(counts), rows.shape[1]), dtype=torch.float64, device=e.device)
            e._bind_stream()
            e.ctx.call("uavac_gather_rows_dev", self._h, _ptr(rows), int(rows.shape[0]), int(rows.shape[1]),
                       (C.c_int64 * self.world)(*counts), int(dst), _ptr(out))
        e._bind_stream()                                      # back on the caller's stream
        return (stream, out, counts, rows)

    def gather_plan(self, plan: Plan, dst: int = 0, traj=None, parts=None):
        """The final gather as a gather of the PLAN: every rank sends the coefficients, durations and per-spline row
        counts of its missions (204 B per spline; ~10 KB of rows per spline stay where they are -- or were never sampled: a
        rows-free plan, `Engine.plan(rows=False)`, travels just the same), and `dst` re-samples
        them with the very kernel the peers ran -> (Plan of all missions on dst | None, row counts per rank).  A ragged
        batch (`RaggedBatch`, or the `RaggedPlan` of `plan_collision_free`) travels the same way, with the number of splines
        of every mission as one more column, and comes back as a RaggedBatch.
        `gathered.traj` equals what `gather_rows(plan.traj)` delivers, bit for bit.  Synchronous."""
        return self.gather_finish(self.gather_plan_begin(plan, dst, traj=traj, parts=parts))

    def plan_counts(self, plan):
        """(segments per rank, rows per rank) of a plan gather: two tiny synchronous all-gathers.  `gather_plan_begin(...,
        known_counts=...)` takes them from a previous call for the same job instead of asking again."""
        ragged = not hasattr(plan, "m")
        n_seg = int(plan.seg_offsets_host[-1]) if ragged else plan.B * int(plan.m)
        self.engine._bind_stream()
        return self.counts(n_seg), self.counts(plan.total_rows)

    def _transfer_stream(self):
        if getattr(self, "_xfer", None) is None:
            self._xfer = self.engine._torch.cuda.Stream(device=self.engine.device)
        return self._xfer

    def gather_plan_begin(self, plan: Plan, dst: int = 0, stream=None, traj=None, parts=None, known_counts=None):
        """Enqueue `gather_plan` and return at once (`stream`, ticket, and what may run meanwhile: as for `gather_rows_begin`;
        the root's re-sampling is enqueued on that stream too, behind the receives).  `traj`: a preallocated row buffer for
        the root.  A ragged batch makes the root wait inside this call for the splines-per-mission column (it sizes the
        segment table on the host); a uniform batch returns at once on every rank.
        `parts` (uniform batches): None = one transfer, then one re-sampling.  True or a tuple of cumulative shares
        (`sharding.PIPELINE_SHARES`) = PIPELINED: durations and row counts of the whole blocks travel first (12 B per spline: the
        root can lay out every mission's rows), then the coefficients in parts (`uavac_gather_plan_part_dev`) on a transfer
        stream of their own; the root samples part p of every rank's block (`Engine.sample_range`) as soon as it has arrived,
        while part p + 1 is on its links.  The rows are the same bits in the same places; the root's critical path loses the
        transfer time of all but the first part.
        `known_counts` = `plan_counts(plan)` of an earlier gather of the same job: skips the two synchronous all-gathers."""
        e, torch = self.engine, self.engine._torch
        if getattr(plan, "coeffs", None) is None or not plan.coeffs.is_cuda:
            raise ValueError("gather_plan takes a device-resident Plan, RaggedBatch or RaggedPlan with its batch")
        ragged = not hasattr(plan, "m")
        if ragged and getattr(plan, "seg_offsets", None) is None:
            raise ValueError("this plan has neither one segment count for the batch nor seg_offsets")
        if parts is not None and parts is not False:
            if ragged:
                raise ValueError("the pipelined gather cuts blocks at mission boundaries every rank can compute: uniform batches only")
            from .sharding import PIPELINE_SHARES
            return self._gather_plan_pipelined(plan, dst, stream, traj, PIPELINE_SHARES if parts is True else tuple(parts), known_counts)
        m = 0 if ragged else int(plan.m)
        n_seg = int(plan.seg_offsets_host[-1]) if ragged else plan.B * m
        here = torch.cuda.current_stream(e.device)
        stream = here if stream is None else stream
        if stream is not here:
            stream.wait_stream(here)
        with torch.cuda.stream(stream):
            seg_counts, row_counts = known_counts if known_counts is not None else self.plan_counts(plan)
            kw = dict(device=e.device)
            per_mission = b_counts = None
            if ragged:
                # splines per mission travel as one more (exact) f64 column through the row gather
                b_counts = self.counts(plan.B)
                mine = (plan.seg_offsets[1:] - plan.seg_offsets[:-1]).to(torch.float64).reshape(-1, 1).contiguous()
                per_mission = torch.empty((sum(b_counts), 1), dtype=torch.float64, **kw) if self.rank == dst else None
                e._bind_stream()
                e.ctx.call("uavac_gather_rows_dev", self._h, _ptr(mine), plan.B, 1, (C.c_int64 * self.world)(*b_counts), int(dst),
                           _ptr(per_mission))
            elif any(c % m for c in seg_counts):
                raise ValueError(f"every rank must plan with the same segment count (m = {m} here)")
            gathered = keep = None
            S = sum(seg_counts)
            co = tm = sr = None
            times = getattr(plan, "times", None)
            if self.rank == dst:
                co = torch.empty((S, 8, 3), dtype=torch.float64, **kw)
                tm = torch.empty((S,), dtype=torch.float64, **kw) if times is not None else None
                sr = torch.empty((S,), dtype=torch.int32, **kw)
            e._bind_stream()
            e.ctx.call("uavac_gather_plan_dev", self._h, _ptr(plan.coeffs), _ptr(times), _ptr(plan.seg_rows), n_seg,
                       (C.c_int64 * self.world)(*seg_counts), int(dst), _ptr(co), _ptr(tm), _ptr(sr))
            if self.rank == dst:
                if ragged:
                    gathered = e.ragged_from_parts(co, tm, sr, per_mission.reshape(-1).round().to(torch.int64), plan.velocity, plan.dt,
                                                   total_rows=sum(row_counts), traj=traj)
                else:
                    gathered = e.plan_from_parts(co, tm, sr, m, plan.velocity, plan.dt, total_rows=sum(row_counts), traj=traj)
            keep = (plan, per_mission)
        e._bind_stream()                                      # back on the caller's stream
        return (stream, gathered, row_counts, keep)

    def _gather_plan_pipelined(self, plan, dst, stream, traj, shares, known_counts):
        e, torch = self.engine, self.engine._torch
        from .sharding import part_bounds
        m = int(plan.m)
        here = torch.cuda.current_stream(e.device)
        stream = here if stream is None else stream
        xfer = self._transfer_stream()                        # every RCCL operation of this gather is issued on it
        xfer.wait_stream(here)
        if stream is not here:
            stream.wait_stream(here)
        root = self.rank == dst
        kw = dict(device=e.device)
        times = getattr(plan, "times", None)
        arr = lambda v: (C.c_int64 * self.world)(*[int(x) for x in v])      # noqa: E731
        events, co = [], None
        with torch.cuda.stream(xfer):
            seg_counts, row_counts = known_counts if known_counts is not None else self.plan_counts(plan)
            if any(c % m for c in seg_counts):
                raise ValueError(f"every rank must plan with the same segment count (m = {m} here)")
            missions = [c // m for c in seg_counts]
            bounds = [part_bounds(b, shares) for b in missions]             # the same arithmetic on every rank
            S = sum(seg_counts)
            tm = sr = None
            if root:
                co = torch.empty((S, 8, 3), dtype=torch.float64, **kw)
                tm = torch.empty((S,), dtype=torch.float64, **kw) if times is not None else None
                sr = torch.empty((S,), dtype=torch.int32, **kw)
            e._bind_stream()
            # durations + rows per spline of the WHOLE blocks first: the root lays out every mission's rows from them
            e.ctx.call("uavac_gather_plan_part_dev", self._h, None, _ptr(times), _ptr(plan.seg_rows), arr(seg_counts),
                       arr([0] * self.world), arr(seg_counts), int(dst), None, _ptr(tm), _ptr(sr))
            for p in range(len(shares)):
                first = [bounds[r][p] * m for r in range(self.world)]
                count = [(bounds[r][p + 1] - bounds[r][p]) * m for r in range(self.world)]
                e.ctx.call("uavac_gather_plan_part_dev", self._h, _ptr(plan.coeffs), None, None, arr(seg_counts), arr(first), arr(count),
                           int(dst), _ptr(co), None, None)
                if root:
                    ev = torch.cuda.Event()
                    ev.record(xfer)
                    events.append(ev)
        gathered = None
        if root:
            with torch.cuda.stream(stream):
                stream.wait_event(events[0])
                gathered = e.plan_from_parts(co, tm, sr, m, plan.velocity, plan.dt, total_rows=sum(row_counts), traj=traj, sample=False)
            base = np.concatenate([[0], np.cumsum(missions)])
            self._sample_parts(gathered, [[(base[r] + bounds[r][p], base[r] + bounds[r][p + 1]) for r in range(self.world)]
                                          for p in range(len(shares))], events, stream, (co, tm, sr))
        stream.wait_stream(xfer)                              # whoever waits for `stream` (gather_finish) has waited for the transfers
        e._bind_stream()                                      # back on the caller's stream
        return (stream, gathered, row_counts, (plan, None))

    def _sample_parts(self, gathered, ranges_by_part, events, stream, inputs):
        """The root's sampling of a pipelined gather: part p's ranges (one per rank) behind events[p].  The ranges of one part go
        to `stream` and to two helper streams in turn: launches on ONE stream run one after the other, and every launch ends with
        a tail in which the chip drains -- 32 launches in a row take 4.2 ms for config 4's 20.85 GB against 3.9 ms for one launch;
        spread over three streams the tails of one launch lie under the body of the next.  `stream` waits for the helpers at
        the end: whoever waits for `stream` has waited for every row."""
        e, torch = self.engine, self.engine._torch
        if getattr(self, "_helpers", None) is None:
            self._helpers = [torch.cuda.Stream(device=e.device) for _ in range(2)]
        lanes = [stream] + self._helpers
        ready = torch.cuda.Event()
        ready.record(stream)                                  # row offsets laid out (plan_from_parts ran on `stream`)
        for h in self._helpers:
            h.wait_event(ready)
        k = 0
        for p, ranges in enumerate(ranges_by_part):
            for s_ in lanes:
                s_.wait_event(events[p])
            for b0, b1 in ranges:
                if b1 > b0:
                    with torch.cuda.stream(lanes[k % len(lanes)]):
                        e.sample_range(gathered, b0, b1)
                    k += 1
        for h in self._helpers:
            stream.wait_stream(h)
        for t in list(inputs) + [gathered.row_offsets, gathered.first_yaw, gathered.traj]:
            if t is not None:                                 # allocated under one stream, used by the sampler on all three
                for s_ in lanes:
                    t.record_stream(s_)
        e._bind_stream()

    def gather_finish(self, ticket):
        """Wait for a gather started with `gather_rows_begin` / `gather_plan_begin` -> (result on dst | None, counts)."""
        e, torch = self.engine, self.engine._torch
        stream, out, counts, _rows = ticket
        with torch.cuda.stream(stream):
            e._bind_stream()
            e.ctx.call("uavac_comm_finish", self._h)
        e._bind_stream()                                      # back on the caller's stream
        here = torch.cuda.current_stream(e.device)
        if stream is not here and out is not None:
            # the result was allocated under the side stream and is consumed on the caller's: tell the caching allocator
            for t in ([out] if torch.is_tensor(out) else
                      [getattr(out, k, None) for k in ("traj", "coeffs", "times", "seg_rows", "row_offsets", "first_yaw", "status",
                                                       "seg_offsets")]):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(here)
        return out, counts

    def loopback(self, src):
        """Self-test of the transport on one GPU: src -> copy through ncclSend + ncclRecv to this very rank."""
        e, torch = self.engine, self.engine._torch
        src = src.contiguous()
        dst = torch.empty_like(src)
        e._bind_stream()
        e.ctx.call("uavac_comm_loopback_dev", self._h, _ptr(src), _ptr(dst), int(src.numel()))
        e.ctx.call("uavac_comm_finish", self._h)
        return dst

    def close(self, abort: bool = False):
        h = getattr(self, "_h", None)
        if h is not None and h.value and self.engine.ctx._h.value:
            try:
                self.engine.ctx.call("uavac_comm_abort" if abort else "uavac_comm_destroy", h)
            finally:
                self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:                # pragma: no cover - interpreter shutdown
            pass
