"""Batched planning over libuavac.so: `Engine` (one GPU, one `uavac_ctx`) and the device-resident plans it returns.

The batched form of `MinimumSnap(...).get_trajectory()` (uav_ac/planning/minimum_snap.py:59-124, upstream path) and of
`RRTStar.run()` (planning/rrt.py).  PyTorch is plumbing only: it owns device memory and the HIP stream; all arithmetic
happens in the hand-written HIP kernels behind the C ABI (include/uavac.h).  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _native as nat

_P = C.c_void_p


def _torch():
    import torch
    return torch


def _ptr(t) -> _P:
    return _P(0 if t is None else t.data_ptr())


@dataclass
class Plan:
    """Device-resident result of planning B missions of m segments."""
    B: int
    m: int
    velocity: float
    dt: float
    waypoints: "object"      # (B, m+1, 3) f64
    times: "object"          # (B, m) f64
    seg_rows: "object"       # (B, m) i32
    row_offsets: "object"    # (B+1,) i64
    coeffs: "object"         # (B, 8m, 3) f64
    status: "object"         # (B,) i32: 0 ok, 1 singular
    traj: "object"           # (N, 11) f64, missions back to back; None for a rows-free plan (Engine.plan(..., rows=False))
    total_rows: int          # N (known for a rows-free plan too: what the root of a plan gather will sample)
    yaw: "object" = None     # (N,) f64 or None: the yaw column on its own (== traj[:, 9]); one way to feed the plan-fed rollout
    first_yaw: "object" = None   # (B,) f64: heading of each mission's first row that has one; lets the rollout scan the yaw itself
    placement_ms: "object" = None    # sampler times of the candidate row buffers when plan(..., placement_trials > 1) chose one
    pooled: bool = False             # the rows live in the Engine's pooled buffer (shared with every other pooled plan of that Engine)
    epoch: int = 0                   # bumped whenever the plan is re-solved / re-sampled in place (Engine.replan / solve / sample):
                                     # an attached Fleet then rebuilds the yaw scan it carries instead of trusting a stale one

    def mission(self, b: int) -> np.ndarray:
        """Rows of mission b as a fresh host array (N_b, 11) -- the reference's `full_trajectory`."""
        if self.traj is None:
            raise ValueError("a rows-free plan holds no rows: Engine.sample_rows(plan) samples them")
        ro = self.row_offsets[b:b + 2].cpu().numpy()
        return self.traj[int(ro[0]):int(ro[1])].cpu().numpy().copy()

    @property
    def algorithmic_bytes(self) -> int:
        """24(m+1) in + 192 m coefficients out + 88 N rows out, per mission, summed (a rows-free plan: the
        first heading, 8 B per mission, instead of the rows)."""
        rows = 88 * self.total_rows if self.traj is not None else 8 * self.B
        return self.B * (24 * (self.m + 1) + 192 * self.m) + rows


@dataclass
class RaggedPlan:
    """Trajectories of B missions whose segment counts differ (after obstacle-driven midpoint insertion).
    Has what `Fleet` needs from a Plan: traj, row_offsets, start positions."""
    B: int
    velocity: float
    dt: float
    final_waypoints: list            # B host arrays (m_b + 1, 3): the waypoint lists after insertion
    row_offsets: "object"            # (B+1,) i64
    traj: "object"                   # (N, 11) f64
    total_rows: int
    start_positions: "object"        # (B, 3) f64
    converged: "object" = None       # (B,) bool, host: False where the bounded obstacle loop gave up
    batch: "object" = None           # the RaggedBatch the rows were sampled from (coefficients, rows per spline, first headings):
                                     # lets a Fleet fly the plan from its coefficients and RcclComm.gather_plan ship it

    def mission(self, b: int) -> np.ndarray:
        ro = self.row_offsets[b:b + 2].cpu().numpy()
        return self.traj[int(ro[0]):int(ro[1])].cpu().numpy().copy()

    def __getattr__(self, name):     # coeffs, seg_rows, seg_offsets, first_yaw, max_m, times ...: the batch's, when there is one
        batch = self.__dict__.get("batch")
        if batch is not None and name in ("coeffs", "seg_rows", "seg_offsets", "seg_offsets_host", "first_yaw", "max_m", "times",
                                          "waypoints", "status"):
            return getattr(batch, name)
        raise AttributeError(name)


@dataclass
class RaggedBatch:
    """One ragged planning call (`Engine.plan_ragged`): B missions with m_b segments each, everything per-segment back to
    back in mission order (include/uavac.h, "Ragged batches")."""
    B: int
    max_m: int
    velocity: float
    dt: float
    seg_offsets: "object"            # (B+1,) i64, device
    seg_offsets_host: np.ndarray     # the same on the host
    waypoints: "object"              # (S + B, 3) f64
    times: "object"                  # (S,) f64
    seg_rows: "object"               # (S,) i32
    row_offsets: "object"            # (B+1,) i64
    coeffs: "object"                 # (S, 8, 3) f64
    status: "object"                 # (B,) i32, 0 = ok
    traj: "object"                   # (N, 11) f64; None for a rows-free batch
    total_rows: int
    first_yaw: "object"              # (B,) f64
    hit: "object" = None             # (S,) i32 when a cuboid was given

    def mission(self, b: int) -> np.ndarray:
        if self.traj is None:
            raise ValueError("a rows-free batch (plan_ragged(..., rows=False)) holds no rows")
        ro = self.row_offsets[b:b + 2].cpu().numpy()
        return self.traj[int(ro[0]):int(ro[1])].cpu().numpy().copy()

    @property
    def start_positions(self):
        """(B, 3): first waypoint of every mission (what `Fleet` starts its vehicles from)."""
        import torch
        if self.waypoints is None:                  # assembled from gathered parts: c0 of a mission's first spline IS its first waypoint
            return self.coeffs[self.seg_offsets[:-1], 0, :]
        first = self.seg_offsets[:-1] + torch.arange(self.B, dtype=self.seg_offsets.dtype, device=self.seg_offsets.device)
        return self.waypoints[first]

    def mission_coeffs(self, b: int) -> np.ndarray:
        s0, s1 = int(self.seg_offsets_host[b]), int(self.seg_offsets_host[b + 1])
        return self.coeffs[s0:s1].reshape(-1, 3).cpu().numpy().copy()


@dataclass
class RRTDeviceBatch:
    """Device-resident results of `Engine.rrt_star` (torch tensors; layouts of include/uavac.h).
    counts[:, k]: 0 n_nodes, 1 iterations begun, 2 status, 3 entries when best_tree was stored, 4 best_path rows,
    5 dynamic_it_counter."""
    nodes: "object"
    canon: "object"
    parent: "object"
    best_parent: "object"
    best_path: "object"
    counts: "object"
    best_cost: "object"

    def to_host(self):
        """-> uav_ac.planning.rrt.RRTBatch (NumPy)."""
        from .planning.rrt import RRTBatch
        c = self.counts.cpu().numpy()
        return RRTBatch(self.nodes.cpu().numpy(), self.canon.cpu().numpy(), self.parent.cpu().numpy(),
                        self.best_parent.cpu().numpy(), self.best_path.cpu().numpy(), c[:, 0].copy(), c[:, 1].copy(),
                        c[:, 2].copy(), c[:, 3].copy(), c[:, 4].copy(), c[:, 5].copy(), self.best_cost.cpu().numpy())


class Engine:
    """One GPU, one `uavac_ctx`.  Kernels are enqueued on torch's current stream for that device."""
    FAST_ROW_BUFFER_FRACTION_OF_PEAK = 0.70    # `place_rows`: a row buffer the sampler fills at this share of the device's HBM peak is of the fast kind

    def __init__(self, device=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise nat.UavacError(nat.EHIP, "no GPU visible: the uavac engine has no CPU fallback")
        dev = torch.device("cuda") if device is None else torch.device(device)
        if dev.type != "cuda":
            raise nat.UavacError(nat.EHIP, f"device {dev} is not a GPU: the uavac engine has no CPU fallback")
        if dev.index is None:                                   # "cuda": the thread's current device
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        # the ctx remembers its device; every C entry point makes it current for its own duration (and restores the
        # caller's), so an Engine for cuda:1 works while cuda:0 is torch's current device
        self.ctx = nat.Context(self.device.index)
        self._torch = torch
        self._comm = None
        self._row_pool = None            # plan(..., pool=True): the one pooled row buffer (rows x 11, float64)

    # -- plumbing ---------------------------------------------------------------
    def _bind_stream(self):
        self.ctx.set_stream(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, dtype):
        torch = self._torch
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    def clock_probe_begin(self, window_us: int, stream=None):
        """Start ONE wavefront on `stream` (a side stream: it then runs BESIDE whatever the current stream executes) that stamps
        shader cycles and real time `window_us` apart (`uavac_clock_probe_dev`).  Returns the ticket for `clock_probe_ghz`."""
        torch = self._torch
        with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream(self.device)):
            stamps = torch.empty((4,), dtype=torch.int64, device=self.device)      # (the kernel writes all four; no fill on another stream)
            self._bind_stream()
            self.ctx.call("uavac_clock_probe_dev", int(window_us), _ptr(stamps))
        self._bind_stream()                                   # back on the caller's stream
        return stamps

    @staticmethod
    def clock_probe_ghz(stamps) -> float:
        """Shader clock over a finished probe's window: (cycles1 - cycles0) / (real1 - real0) x 100 MHz.  (Synchronises.)"""
        c0, r0, c1, r1 = (int(v) for v in stamps.cpu().tolist())
        return (c1 - c0) / max(1, r1 - r0) * 0.1

    # -- planning ---------------------------------------------------------------
    def plan(self, waypoints, velocity: float = 1.0, dt: float = 0.01, strict: bool = True, dense_yaw: bool = False,
             placement_trials: int = 1, pool: bool = False, rows: bool = True) -> Plan:
        """Batched `MinimumSnap(path, None, velocity, dt).get_trajectory()` (minimum_snap.py:59-61,97-124).
        `strict`: raise UavacError(ESINGULAR) when a mission's knot system is singular (a repeated waypoint) instead of
        returning NaN coefficients for it; with strict=False inspect `plan.status`.
        `dense_yaw`: also keep the yaw column on its own (`plan.yaw`, 8 B per row).  Not needed to fly the plan: the
        plan-fed rollout scans the yaw itself from `plan.first_yaw` (8 B per mission).
        `placement_trials` > 1 (opt-in; default 1 = take the first allocation): draw up to that many row buffers one after the
        other and keep the first of the fast kind, else the fastest seen (`place_rows`: at most TWO alive at any time).  Row
        buffers come in three kinds (DESIGN K2, NOTES R4-6): the bench's 7.5 GB of rows take the default chunk-streaming sampler
        1.24-1.26 ms into a fast one, 1.38-1.43 into a slow one, and a process's first large allocation is usually a slow one.
        `pool=True`: the row buffer comes from / goes to the Engine's pool -- ONE buffer, found once (with `placement_trials`), handed
        to every later pooled plan of at most that many rows, so that the search is paid once per process.  Pooled plans share
        their rows' storage: one of them is current at a time (the use it is meant for: the same fleet planned again and again).
        `rows=False`: the ROWS-FREE chain (`uavac_minsnap_plan_dev` with traj = NULL): durations, row counts, offsets, coefficients
        and the missions' first headings (`uavac_minsnap_first_yaw_dev`), not one sampled row -- `plan.traj` is None.  Everything a
        plan-fed `Fleet` and `RcclComm.gather_plan` need; for the ranks of a multi-GPU job whose trajectories are sampled where they
        are wanted (the gather's root re-samples them from the gathered plan, bit-identical).  `Engine.sample_rows(plan)` adds the
        rows later.
        """
        torch = self._torch
        wp = self._dev(waypoints, torch.float64)
        if wp.dim() != 3 or wp.shape[2] != 3 or wp.shape[1] < 2:
            raise ValueError(f"waypoints must have shape (B, m+1, 3), got {tuple(wp.shape)}")
        if not bool(torch.isfinite(wp).all()):
            raise ValueError("waypoints must be finite")
        B, m = int(wp.shape[0]), int(wp.shape[1]) - 1
        kw = dict(device=self.device)
        times = torch.empty((B, m), dtype=torch.float64, **kw)
        seg_rows = torch.empty((B, m), dtype=torch.int32, **kw)
        row_offsets = torch.empty((B + 1,), dtype=torch.int64, **kw)
        coeffs = torch.empty((B, 8 * m, 3), dtype=torch.float64, **kw)
        status = torch.zeros((B,), dtype=torch.int32, **kw)
        self._bind_stream()
        if not rows:
            if dense_yaw or pool or int(placement_trials) > 1:
                raise ValueError("dense_yaw, pool and placement_trials are about the rows: not with rows=False")
            first_yaw = torch.empty((B,), dtype=torch.float64, **kw)
            plan = Plan(B, m, float(velocity), float(dt), wp, times, seg_rows, row_offsets, coeffs, status, None, 0, None, first_yaw)
            self.replan(plan)
            plan.epoch = 0
            plan.total_rows = int(row_offsets[-1].item())       # (the one host sync; the rows would have needed it to be allocated)
            if strict:
                self.check(plan)
            return plan
        self.ctx.call("uavac_minsnap_row_counts_dev", _ptr(wp), B, m, float(velocity), float(dt), _ptr(times),
                      _ptr(seg_rows), _ptr(row_offsets))
        self.ctx.call("uavac_minsnap_solve_dev", _ptr(wp), _ptr(times), B, m, _ptr(coeffs), _ptr(status))
        total = int(row_offsets[-1].item())                 # the one host sync: sizes the trajectory buffer
        pooled = pool and self._row_pool is not None and self._row_pool.shape[0] >= total
        traj = self._row_pool[:total] if pooled else torch.empty((total, nat.TRAJ_COLS), dtype=torch.float64, **kw)
        yaw = torch.empty((total,), dtype=torch.float64, **kw) if dense_yaw else None
        first_yaw = torch.empty((B,), dtype=torch.float64, **kw)
        plan = Plan(B, m, float(velocity), float(dt), wp, times, seg_rows, row_offsets, coeffs, status, traj, total, yaw, first_yaw)
        del traj                                                 # (place_rows may release the first draw: no second reference to it)
        self.sample(plan)
        if int(placement_trials) > 1 and total > 0 and not pooled:
            self.place_rows(plan, int(placement_trials))
        if pool and not pooled:
            self._row_pool = plan.traj                           # (a larger pooled plan later replaces it)
        plan.pooled = bool(pool)
        if strict:
            self.check(plan)
        return plan

    def empty_plan(self, m: int, velocity: float = 1.0, dt: float = 0.01, rows: bool = True) -> Plan:
        """A Plan of NO missions: what a rank holds that plans and flies nothing -- the root of a final gather that only assembles
        the trajectories -- so that it takes part in the gathers (`RcclComm.gather_rows` / `gather_plan`) with empty blocks."""
        torch = self._torch
        kw = dict(device=self.device)
        m = int(m)
        z = lambda shape, dtype: torch.zeros(shape, dtype=dtype, **kw)      # noqa: E731
        return Plan(0, m, float(velocity), float(dt), z((0, m + 1, 3), torch.float64), z((0, m), torch.float64), z((0, m), torch.int32),
                    z((1,), torch.int64), z((0, 8 * m, 3), torch.float64), z((0,), torch.int32),
                    z((0, nat.TRAJ_COLS), torch.float64) if rows else None, 0, None, z((0,), torch.float64))

    def hbm_peak_bytes_per_s(self) -> float:
        """The device's HBM peak from its own properties (memory clock x bus width x 2, DDR): 8.0e12 on MI355X."""
        p = self._torch.cuda.get_device_properties(self.device)
        clock_khz = getattr(p, "memory_clock_rate", 0) or 0
        width_bits = getattr(p, "memory_bus_width", 0) or 0
        peak = 2.0 * clock_khz * 1e3 * width_bits / 8.0
        return peak if peak > 1e11 else 8.0e12

    def place_rows(self, plan: Plan, trials: int):
        """Optional: choose `plan.traj` among up to `trials` candidate allocations by timing the sampler on each (see `plan`)."""
        torch = self._torch

        def timed(buf):
            # The chip's clock sags within milliseconds of idling (an allocation, a device query) and takes ~30 ms of work to come
            # back: blocks of three sampler runs are timed until two blocks in a row agree to 2 % (ten at most), the last one counts.
            plan.traj = buf
            self.sample(plan)                                    # first touch of fresh pages is not what is compared
            prev = None
            for _ in range(10):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    self.sample(plan)
                b.record()
                b.synchronize()
                t = a.elapsed_time(b) / 3
                if prev is not None and abs(t - prev) <= 0.02 * prev:
                    break
                prev = t
            return t

        # Draws come one after the other and at most two buffers are alive: the best so far and the candidate.  A released
        # buffer goes back to the DRIVER (torch.cuda.empty_cache(): torch's cache would hand the very same block to the next
        # request) and the next allocation is other physical memory -- consecutive draws walk through the device's memory, of
        # which stretches are fast and stretches are slow (twelve draws on one box: 4 slow, 5 fast, 3 slow; NOTES R4-6).
        # Stop at the first buffer the rows stream into at >= 0.70 of the HBM peak -- the fast kind -- else keep the fastest.
        # (Round 3 kept every candidate alive side by side: 4x the row memory; round 2 stopped at "7 % below the slowest
        # seen", which a still slower outlier satisfied for a slow buffer.)
        row_bytes = float(plan.total_rows) * nat.TRAJ_COLS * 8.0
        fast_ms = row_bytes / (self.FAST_ROW_BUFFER_FRACTION_OF_PEAK * self.hbm_peak_bytes_per_s()) * 1e3
        best, times = plan.traj, [timed(plan.traj)]
        best_t = times[0]
        while len(times) < trials and best_t > fast_ms:
            try:
                cand = torch.empty_like(best)
            except RuntimeError:                                 # out of memory: keep what there is
                break
            times.append(timed(cand))                            # (every candidate holds the same rows afterwards)
            if times[-1] < best_t:
                best, best_t = cand, times[-1]
            del cand
            plan.traj = best
            torch.cuda.empty_cache()
        plan.traj = best
        plan.placement_ms = times

    def replan(self, plan: Plan):
        """The whole chain again into plan's buffers -- times + row counts, offsets, solve, sampler (+ yaw column) --
        enqueued by ONE call into the C ABI (`uavac_minsnap_plan_dev`): no allocation, no sync, no Python between
        the four launches.  The buffers keep their size: a plan that would need more rows than `plan.traj` holds is
        refused on the device AS A WHOLE (flag 2, see `take_flags`): every array of the plan keeps what it held, so the
        previous plan stays consistent and flyable.  A rows-free plan (`plan.traj` is None) runs the rows-free chain: times + row
        counts, offsets, solve, first headings -- nothing to refuse."""
        if plan.B == 0:
            return                                               # (`empty_plan`: nothing to plan)
        self._bind_stream()
        cap = 0 if plan.traj is None else int(plan.traj.shape[0])
        self.ctx.call("uavac_minsnap_plan_dev", _ptr(plan.waypoints), plan.B, plan.m, plan.velocity, plan.dt,
                      _ptr(plan.times), _ptr(plan.seg_rows), _ptr(plan.row_offsets), _ptr(plan.coeffs), _ptr(plan.status),
                      _ptr(plan.traj), cap, _ptr(plan.yaw), _ptr(plan.first_yaw))
        plan.epoch += 1

    def sample_rows(self, plan: Plan, traj=None):
        """Give a rows-free plan its rows: allocate (or take `traj`, >= plan.total_rows rows) and sample -- the same rows, bit for
        bit, as `plan(..., rows=True)` would have written."""
        torch = self._torch
        total = int(plan.row_offsets[-1].item())
        if traj is None:
            traj = torch.empty((total, nat.TRAJ_COLS), dtype=torch.float64, device=self.device)
        elif traj.shape[0] < total or traj.dtype != torch.float64 or not traj.is_contiguous():
            raise ValueError("traj must be a contiguous float64 tensor with at least total_rows rows")
        plan.traj, plan.total_rows = traj[:total], total
        self.sample(plan)
        return plan

    def first_yaw(self, plan):
        """The missions' first headings from coefficients and row counts alone (`uavac_minsnap_first_yaw_dev`; a Plan or a
        RaggedBatch) -> (B,) f64: bit for bit what the sampler writes into `plan.first_yaw`."""
        torch = self._torch
        out = torch.empty((plan.B,), dtype=torch.float64, device=self.device)
        ragged = hasattr(plan, "seg_offsets")
        self._bind_stream()
        self.ctx.call("uavac_minsnap_first_yaw_dev", _ptr(plan.coeffs), _ptr(plan.seg_rows), _ptr(plan.seg_offsets) if ragged else None,
                      plan.B, plan.max_m if ragged else plan.m, float(plan.dt), _ptr(out))
        return out

    def sample_range(self, plan: Plan, b0: int, b1: int):
        """The rows (and first headings) of missions [b0, b1) of a uniform plan, written where `sample(plan)` writes them: the
        row offsets are absolute, so a sub-range is the same kernel on offset pointers.  Any cover of [0, B) by ranges, in any
        order, leaves the rows of one `sample(plan)` bit for bit -- what lets the root of a pipelined plan gather sample a part
        of every peer's block while the next part arrives.  No allocation, no sync."""
        b0, b1 = int(b0), int(b1)
        if not (0 <= b0 <= b1 <= plan.B):
            raise ValueError("need 0 <= b0 <= b1 <= plan.B")
        if plan.traj is None:
            raise ValueError("a rows-free plan has no row buffer: Engine.sample_rows(plan) allocates one and samples")
        if b1 == b0:
            return
        self._bind_stream()
        fy = getattr(plan, "first_yaw", None)
        self.ctx.call("uavac_minsnap_sample_derivs_dev", _ptr(plan.coeffs[b0:b1]), _ptr(plan.seg_rows[b0:b1]), _ptr(plan.row_offsets[b0:]),
                      b1 - b0, plan.m, plan.dt, _ptr(plan.traj), _ptr(plan.yaw), _ptr(None if fy is None else fy[b0:b1]), None, None)

    def plan_from_parts(self, coeffs, times, seg_rows, m: int, velocity: float, dt: float, total_rows: int = None,
                        traj=None, sample: bool = True) -> Plan:
        """A Plan from its solved parts -- coefficients (B, 8m, 3), durations (B, m) or None, rows per spline (B, m) -- e.g.
        the peers' plans after `RcclComm.gather_plan`: row offsets from the row counts (`uavac_minsnap_row_offsets_dev`),
        then the sampler writes the rows (and the first headings).  The rows are a deterministic function of coefficients,
        row counts and dt: bit-identical to the rows of the plan the parts came from.  `total_rows` (when the caller knows
        it) avoids the one host synchronisation that sizes the row buffer; `traj`: a preallocated (>= total, 11) buffer.
        `sample=False`: lay the rows out (offsets, buffers) but leave the sampling to the caller's `sample_range` calls."""
        torch = self._torch
        co = self._dev(coeffs, torch.float64).reshape(-1, 8 * int(m), 3)
        sr = self._dev(seg_rows, torch.int32).reshape(-1, int(m))
        B = int(co.shape[0])
        if sr.shape[0] != B or B < 1:
            raise ValueError("coeffs and seg_rows disagree on the number of missions")
        tm = None if times is None else self._dev(times, torch.float64).reshape(B, int(m))
        kw = dict(device=self.device)
        row_offsets = torch.empty((B + 1,), dtype=torch.int64, **kw)
        self._bind_stream()
        self.ctx.call("uavac_minsnap_row_offsets_dev", _ptr(sr), B, int(m), _ptr(row_offsets))
        total = int(row_offsets[-1].item()) if total_rows is None else int(total_rows)
        if traj is None:
            traj = torch.empty((total, nat.TRAJ_COLS), dtype=torch.float64, **kw)
        elif traj.shape[0] < total or traj.dtype != torch.float64 or not traj.is_contiguous():
            raise ValueError("traj must be a contiguous float64 tensor with at least total_rows rows")
        first_yaw = torch.empty((B,), dtype=torch.float64, **kw)
        status = torch.zeros((B,), dtype=torch.int32, **kw)
        plan = Plan(B, int(m), float(velocity), float(dt), None, tm, sr, row_offsets, co, status, traj[:total], total, None, first_yaw)
        if sample:
            self.sample(plan)
        return plan

    def ragged_from_parts(self, coeffs, times, seg_rows, seg_counts, velocity: float, dt: float, total_rows: int = None,
                          traj=None) -> RaggedBatch:
        """`plan_from_parts` for a ragged batch: coefficients (S, 8, 3), durations (S,) or None, rows per spline (S,) back to back
        and the number of splines of every mission (B,) -> RaggedBatch with the rows re-sampled (bit-identical)."""
        torch = self._torch
        co = self._dev(coeffs, torch.float64).reshape(-1, 8, 3)
        sr = self._dev(seg_rows, torch.int32).reshape(-1)
        cnt = np.asarray(seg_counts.cpu() if hasattr(seg_counts, "cpu") else seg_counts, dtype=np.int64).reshape(-1)
        B, S = len(cnt), int(cnt.sum())
        if S != co.shape[0] or S != sr.shape[0] or B < 1 or cnt.min() < 1 or cnt.max() > nat.MAX_SEGMENTS:
            raise ValueError("segment counts, coefficients and row counts disagree")
        so_host = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(cnt, out=so_host[1:])
        so = self._dev(so_host, torch.int64)
        max_m = int(cnt.max())
        kw = dict(device=self.device)
        row_offsets = torch.empty((B + 1,), dtype=torch.int64, **kw)
        self._bind_stream()
        self.ctx.call("uavac_minsnap_row_offsets_ragged_dev", _ptr(sr), _ptr(so), B, max_m, _ptr(row_offsets))
        total = int(row_offsets[-1].item()) if total_rows is None else int(total_rows)
        if traj is None:
            traj = torch.empty((total, nat.TRAJ_COLS), dtype=torch.float64, **kw)
        first_yaw = torch.empty((B,), dtype=torch.float64, **kw)
        self.ctx.call("uavac_minsnap_sample_ragged_dev", _ptr(co), _ptr(sr), _ptr(so), _ptr(row_offsets), B, max_m, S, float(dt),
                      _ptr(traj), int(traj.shape[0]), None, None, _ptr(first_yaw))
        tm = None if times is None else self._dev(times, torch.float64).reshape(-1)
        # the first waypoint of every mission is c0 of its first spline; the others are not needed to fly or to ship the plan
        return RaggedBatch(B, max_m, float(velocity), float(dt), so, so_host, None, tm, sr, row_offsets, co,
                           torch.zeros((B,), dtype=torch.int32, **kw), traj[:total], total, first_yaw, None)

    def take_flags(self):
        """Synchronise and return-and-clear the sticky device-side flags of the `_dev` planning entry points:
        [non-finite duration, singular system, trajectory buffer too small, mission longer than 2^31-1 rows]."""
        fl = (C.c_int32 * 4)()
        self._bind_stream()
        self.ctx.call("uavac_take_flags", fl)
        return [int(v) for v in fl]

    def sample_derivatives(self, plan: Plan):
        """Jerk and snap along the plan's rows: (N, 3) each -- `polynom(8, 3, t) @ coeffs` and `polynom(8, 4, t) @
        coeffs`, the samples minimum_snap.py:111-112 holds in comments.  Separate arrays; plan.traj keeps its 11
        columns (and is rewritten with the same values)."""
        torch = self._torch
        jerk = torch.empty((plan.total_rows, 3), dtype=torch.float64, device=self.device)
        snap = torch.empty((plan.total_rows, 3), dtype=torch.float64, device=self.device)
        self._bind_stream()
        self.ctx.call("uavac_minsnap_sample_derivs_dev", _ptr(plan.coeffs), _ptr(plan.seg_rows), _ptr(plan.row_offsets),
                      plan.B, plan.m, plan.dt, _ptr(plan.traj), _ptr(plan.yaw), _ptr(plan.first_yaw), _ptr(jerk), _ptr(snap))
        return jerk, snap

    def yaw_scan(self, velocities, offsets=None):
        """Batched `MinimumSnap._calculate_yaws` (minimum_snap.py:126-136): velocities (N, 3) rows of B sequences back
        to back, sequence b = rows offsets[b]:offsets[b+1] (default: one sequence).  -> yaws (N,) on the GPU."""
        torch = self._torch
        v = self._dev(velocities, torch.float64)
        if v.dim() != 2 or v.shape[1] != 3:
            raise ValueError(f"velocities must have shape (N, 3), got {tuple(v.shape)}")
        n = int(v.shape[0])
        off = self._dev([0, n] if offsets is None else offsets, torch.int64)
        if off.dim() != 1 or off.numel() < 2:
            raise ValueError("offsets must be a 1-D array of B+1 row indices")
        yaws = torch.empty((n,), dtype=torch.float64, device=self.device)
        if n:
            self._bind_stream()
            self.ctx.call("uavac_yaw_scan_dev", _ptr(v), _ptr(off), int(off.numel()) - 1, _ptr(yaws))
        return yaws

    def plan_collision_free(self, waypoints, obstacles, velocity: float = 1.0, dt: float = 0.01,
                            max_iterations: int = 64, strict: bool = True, recheck_passes: int = 0,
                            device_loop: bool = True) -> RaggedPlan:
        """Batched `MinimumSnap(path, obstacles, velocity, dt).get_trajectory()` with obstacles
        (minimum_snap.py:63-95) for B missions at once.

        Per mission the reference's semantics are kept: obstacles are visited in order; for each one the mission is
        planned, every spline with a sample inside the cuboid gets a midpoint inserted before its end waypoint,
        and it is re-planned until clean; earlier obstacles are not re-checked.  Here all missions advance
        together, and a round is ONE call into the C ABI (`uavac_minsnap_obstacle_round_dev`): the still-active missions
        are planned as a ragged batch, their splines scanned for samples inside the cuboid (no rows are stored inside the
        loop), and the midpoints inserted into the next round's waypoint arrays by a kernel; the host reads back four
        counters per round.  The trajectories are sampled once, from the final waypoints.  `device_loop=False` runs round
        2's loop instead (rows sampled in every round, hit flags to the host, midpoints inserted with NumPy): same
        waypoints, kept for comparison.
        `waypoints`: (B, m+1, 3) array or a list of (m_b+1, 3) arrays.  The loop is bounded (the reference's is
        not: it cannot end when a waypoint lies inside a cuboid, or when a leg crosses one squarely).  A mission
        that exhausts `max_iterations` or UAVAC_MAX_SEGMENTS raises RuntimeError when `strict`; otherwise it is
        reported in `RaggedPlan.converged` (False) with its last (still colliding) trajectory and the batch goes on.
        `recheck_passes` > 0 goes beyond the reference: missions that received midpoints are swept over the whole
        obstacle list again (up to that many extra passes, until a pass inserts nothing), which removes the
        conflicts a late midpoint can create with an earlier obstacle.
        """
        if device_loop:
            return self._plan_collision_free_device(waypoints, obstacles, velocity, dt, max_iterations, strict, recheck_passes)
        return self._plan_collision_free_host(waypoints, obstacles, velocity, dt, max_iterations, strict, recheck_passes)

    def _plan_collision_free_device(self, waypoints, obstacles, velocity, dt, max_iterations, strict, recheck_passes) -> RaggedPlan:
        torch = self._torch
        wps = [np.ascontiguousarray(w, dtype=np.float64) for w in waypoints]
        B = len(wps)
        if B == 0 or any(w.ndim != 2 or w.shape[1] != 3 or w.shape[0] < 2 for w in wps):
            raise ValueError("waypoints must be B arrays of shape (m+1, 3)")
        M = nat.MAX_SEGMENTS
        counts = np.array([w.shape[0] - 1 for w in wps], dtype=np.int64)
        if counts.max() > M:
            raise ValueError(f"a mission has {int(counts.max())} segments; at most {M}")
        cuboids = np.zeros((0, 6)) if obstacles is None else np.asarray(obstacles, dtype=np.float64).reshape(-1, 6)
        so_host = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(counts, out=so_host[1:])
        kw = dict(device=self.device)
        S_cap = B * M                                           # no mission ever has more than M segments
        wp_a = torch.empty((S_cap + B, 3), dtype=torch.float64, **kw)
        wp_b = torch.empty_like(wp_a)
        wp_a[:int(so_host[-1]) + B] = self._dev(np.concatenate(wps, axis=0), torch.float64)
        so_a, so_b = self._dev(so_host, torch.int64), torch.empty((B + 1,), dtype=torch.int64, **kw)
        failed = torch.zeros((B,), dtype=torch.int32, **kw)
        max_m = int(counts.max())
        if len(cuboids):
            times = torch.empty((S_cap,), dtype=torch.float64, **kw)
            seg_rows = torch.empty((S_cap,), dtype=torch.int32, **kw)
            row_offsets = torch.empty((B + 1,), dtype=torch.int64, **kw)
            coeffs = torch.empty((S_cap, 8, 3), dtype=torch.float64, **kw)
            hit = torch.empty((S_cap,), dtype=torch.int32, **kw)
            active = torch.empty((B,), dtype=torch.int32, **kw)
            overflow = torch.zeros((B,), dtype=torch.int32, **kw)
            touched = torch.zeros((B,), dtype=torch.int32, **kw)
            counters = torch.zeros((4,), dtype=torch.int32, **kw)
            cub_dev = self._dev(cuboids, torch.float64)
            todo = torch.ones((B,), dtype=torch.int32, **kw)
            self._bind_stream()
            for sweep in range(1 + max(0, int(recheck_passes))):
                touched.zero_()
                for ci in range(len(cuboids)):
                    torch.mul(todo, 1 - failed, out=active)
                    n_active = int(active.sum().item())
                    for it in range(max_iterations + 1):
                        if n_active == 0:
                            break
                        self.ctx.call("uavac_minsnap_obstacle_round_dev", _ptr(wp_a), _ptr(so_a), B, max_m, float(velocity), float(dt),
                                      _ptr(cub_dev[ci]), _ptr(active), _ptr(overflow), _ptr(touched), _ptr(wp_b), _ptr(so_b),
                                      _ptr(counters), _ptr(times), _ptr(seg_rows), _ptr(row_offsets), _ptr(coeffs), _ptr(hit))
                        wp_a, wp_b, so_a, so_b = wp_b, wp_a, so_b, so_a
                        n_active, n_over, max_m, _total = (int(v) for v in counters.tolist())      # the round's one read-back
                        if n_over:
                            if strict:
                                raise RuntimeError(f"obstacle correction needs more than {M} splines")
                            failed.logical_or_(overflow)                 # stays as it is, reported in `converged`
                    else:
                        if n_active:
                            if strict:
                                raise RuntimeError("obstacle correction did not converge (a waypoint inside an obstacle?)")
                            failed.logical_or_(active)
                todo = touched * (1 - failed)                          # only missions that changed can have new conflicts
                if int(todo.sum().item()) == 0:
                    break
            flags = self.take_flags()
            if flags[0]:
                raise ValueError("non-finite waypoint or segment duration")
        # the trajectories, once, from the final waypoints
        so_final = so_a.cpu().numpy()
        S = int(so_final[-1])
        wp_final = wp_a[:S + B]
        batch = self._plan_ragged_tensors(wp_final, so_a, so_final, int((so_final[1:] - so_final[:-1]).max()), velocity, dt, None)
        if strict:
            self.check(batch)
        wp_host = wp_final.cpu().numpy()
        final_wps = [wp_host[so_final[b] + b:so_final[b + 1] + b + 1].copy() for b in range(B)]
        converged = ~failed.cpu().numpy().astype(bool)
        return RaggedPlan(B, float(velocity), float(dt), final_wps, batch.row_offsets, batch.traj, batch.total_rows,
                          batch.start_positions.contiguous(), converged, batch)

    def _plan_collision_free_host(self, waypoints, obstacles, velocity, dt, max_iterations, strict, recheck_passes) -> RaggedPlan:
        """Round 2's loop: every round a ragged planning batch with rows, hit flags to the host, NumPy midpoint insertion."""
        torch = self._torch
        wps = [np.ascontiguousarray(w, dtype=np.float64) for w in waypoints]
        B = len(wps)
        if B == 0 or any(w.ndim != 2 or w.shape[1] != 3 or w.shape[0] < 2 for w in wps):
            raise ValueError("waypoints must be B arrays of shape (m+1, 3)")
        cuboids = np.zeros((0, 6)) if obstacles is None else np.asarray(obstacles, dtype=np.float64).reshape(-1, 6)
        source = [None] * B                                    # mission -> (group Plan, index inside it)
        failed = set()

        def run_round(ids, cub):
            """One planning call for every mission of `ids` (their segment counts differ: a ragged batch)."""
            members = []
            for b in ids:
                if wps[b].shape[0] - 1 > nat.MAX_SEGMENTS:
                    if strict:
                        raise RuntimeError(f"obstacle correction needs more than {nat.MAX_SEGMENTS} splines")
                    failed.add(b)                                     # keeps the plan of the previous round
                else:
                    members.append(b)
            if not members:
                return []
            batch = self.plan_ragged([wps[b] for b in members], velocity, dt, cuboid=cub)
            for j, b in enumerate(members):
                source[b] = (batch, j)
            again = []
            if batch.hit is not None:
                hit = batch.hit.cpu().numpy().astype(bool)
                so = batch.seg_offsets_host
                hit_missions = np.flatnonzero(np.add.reduceat(hit, so[:-1]) > 0) if len(hit) else []
                for j in hit_missions:
                    b = members[j]
                    idx = np.flatnonzero(hit[so[j]:so[j + 1]]) + 1    # spline s -> insert before waypoint s+1
                    if wps[b].shape[0] - 1 + len(idx) > nat.MAX_SEGMENTS and not strict:
                        failed.add(b)                                 # would outgrow the kernels: stop here
                        continue
                    mids = (wps[b][idx - 1] + wps[b][idx]) / 2
                    wps[b] = np.insert(wps[b], idx, mids, axis=0)
                    again.append(b)
            return again

        if len(cuboids) == 0:
            run_round(list(range(B)), None)
        todo = list(range(B))                                  # missions the next pass over the obstacles looks at
        for sweep in range(1 + max(0, int(recheck_passes))):
            touched = set()
            for cub in cuboids:
                active = [b for b in todo if b not in failed]
                for it in range(max_iterations + 1):
                    if not active:
                        break
                    active = run_round(active, cub)
                    touched.update(active)
                else:
                    if strict:
                        raise RuntimeError("obstacle correction did not converge (a waypoint inside an obstacle?)")
                    failed.update(active)
            todo = sorted(touched - failed)                     # only missions that changed can have new conflicts
            if not todo:
                break

        # stitch the final trajectories together in mission order
        nrows = torch.zeros((B,), dtype=torch.int64, device=self.device)
        by_plan = {}
        for b, (plan, j) in enumerate(source):
            by_plan.setdefault(id(plan), (plan, [], []))
            by_plan[id(plan)][1].append(b)
            by_plan[id(plan)][2].append(j)
        parts = []
        for plan, ids, js in by_plan.values():
            ids_t = torch.as_tensor(ids, device=self.device)
            js_t = torch.as_tensor(js, device=self.device)
            length = (plan.row_offsets[1:] - plan.row_offsets[:-1])[js_t]
            nrows[ids_t] = length
            parts.append((plan, ids_t, js_t, length))
        offsets = torch.zeros((B + 1,), dtype=torch.int64, device=self.device)
        offsets[1:] = torch.cumsum(nrows, 0)
        total = int(offsets[-1].item())
        traj = torch.empty((total, nat.TRAJ_COLS), dtype=torch.float64, device=self.device)
        for plan, ids_t, js_t, length in parts:
            rep = torch.repeat_interleave(torch.arange(len(js_t), device=self.device), length)
            within = torch.arange(int(length.sum().item()), device=self.device) - (torch.cumsum(length, 0) - length)[rep]
            traj[offsets[ids_t][rep] + within] = plan.traj[plan.row_offsets[js_t][rep] + within]
        starts = torch.as_tensor(np.stack([w[0] for w in wps]), dtype=torch.float64, device=self.device)
        converged = np.ones(B, dtype=bool)
        converged[sorted(failed)] = False
        return RaggedPlan(B, float(velocity), float(dt), wps, offsets, traj, total, starts, converged)

    def plan_ragged(self, waypoints, velocity: float = 1.0, dt: float = 0.01, cuboid=None, strict: bool = True,
                    rows: bool = True) -> RaggedBatch:
        """`MinimumSnap(path_b, None, velocity, dt).get_trajectory()` for B paths of DIFFERENT lengths in one batch
        (minimum_snap.py:13-57 takes any path; `Engine.plan` wants equal lengths).  `waypoints`: B arrays (m_b + 1, 3),
        1 <= m_b <= UAVAC_MAX_SEGMENTS.  Mission b's rows and coefficients equal those of `plan` on it alone, bit for
        bit.  `cuboid` (6,): also return per-spline hit flags (the collision scan of minimum_snap.py:81-87).
        `rows=False`: no rows (`batch.traj` is None), the first headings from `uavac_minsnap_first_yaw_dev`; not with `cuboid`."""
        torch = self._torch
        wps = [np.ascontiguousarray(w, dtype=np.float64) for w in waypoints]
        B = len(wps)
        if B == 0 or any(w.ndim != 2 or w.shape[1] != 3 or w.shape[0] < 2 for w in wps):
            raise ValueError("waypoints must be B arrays of shape (m_b + 1, 3)")
        counts = np.array([w.shape[0] - 1 for w in wps], dtype=np.int64)
        max_m = int(counts.max())
        if max_m > nat.MAX_SEGMENTS:
            raise ValueError(f"a mission has {max_m} segments; at most {nat.MAX_SEGMENTS}")
        so_host = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(counts, out=so_host[1:])
        S = int(so_host[-1])
        wp = self._dev(np.concatenate(wps, axis=0), torch.float64)
        so = self._dev(so_host, torch.int64)
        batch = self._plan_ragged_tensors(wp, so, so_host, max_m, velocity, dt, cuboid, rows)
        if strict:
            self.check(batch)
        return batch

    def _plan_ragged_tensors(self, wp, so, so_host, max_m: int, velocity: float, dt: float, cuboid, rows: bool = True) -> RaggedBatch:
        """`plan_ragged` on device-resident waypoints wp (S + B, 3) / seg_offsets so (B + 1,) (so_host: the same on the host)."""
        torch = self._torch
        B, S = len(so_host) - 1, int(so_host[-1])
        kw = dict(device=self.device)
        times = torch.empty((S,), dtype=torch.float64, **kw)
        seg_rows = torch.empty((S,), dtype=torch.int32, **kw)
        row_offsets = torch.empty((B + 1,), dtype=torch.int64, **kw)
        coeffs = torch.empty((S, 8, 3), dtype=torch.float64, **kw)
        status = torch.zeros((B,), dtype=torch.int32, **kw)
        first_yaw = torch.empty((B,), dtype=torch.float64, **kw)
        hit = aabb = None
        if cuboid is not None:
            hit = torch.empty((S,), dtype=torch.int32, **kw)
            aabb = self._dev(np.asarray(cuboid, dtype=np.float64).reshape(6), torch.float64)
        self._bind_stream()
        self.ctx.call("uavac_minsnap_row_counts_ragged_dev", _ptr(wp), _ptr(so), B, max_m, float(velocity), float(dt),
                      _ptr(times), _ptr(seg_rows), _ptr(row_offsets))
        self.ctx.call("uavac_minsnap_solve_ragged_dev", _ptr(wp), _ptr(times), _ptr(so), B, max_m, _ptr(coeffs), _ptr(status))
        if not rows:
            if cuboid is not None:
                raise ValueError("the collision scan walks the rows' positions: not with rows=False")
            self.ctx.call("uavac_minsnap_first_yaw_dev", _ptr(coeffs), _ptr(seg_rows), _ptr(so), B, max_m, float(dt), _ptr(first_yaw))
            return RaggedBatch(B, max_m, float(velocity), float(dt), so, so_host, wp, times, seg_rows, row_offsets, coeffs, status,
                               None, int(row_offsets[-1].item()), first_yaw, None)
        total = int(row_offsets[-1].item())
        traj = torch.empty((total, nat.TRAJ_COLS), dtype=torch.float64, **kw)
        self.ctx.call("uavac_minsnap_sample_ragged_dev", _ptr(coeffs), _ptr(seg_rows), _ptr(so), _ptr(row_offsets), B, max_m,
                      S, float(dt), _ptr(traj), total, _ptr(aabb), _ptr(hit), _ptr(first_yaw))
        return RaggedBatch(B, max_m, float(velocity), float(dt), so, so_host, wp, times, seg_rows, row_offsets, coeffs, status,
                           traj, total, first_yaw, hit)

    def solve(self, plan: Plan):
        """Re-run times/row counts + coefficient solve into plan's buffers (no allocation, no sync)."""
        self._bind_stream()
        self.ctx.call("uavac_minsnap_row_counts_dev", _ptr(plan.waypoints), plan.B, plan.m, plan.velocity, plan.dt,
                      _ptr(plan.times), _ptr(plan.seg_rows), _ptr(plan.row_offsets))
        self.ctx.call("uavac_minsnap_solve_dev", _ptr(plan.waypoints), _ptr(plan.times), plan.B, plan.m,
                      _ptr(plan.coeffs), _ptr(plan.status))
        plan.epoch += 1

    def sample(self, plan: Plan):
        """Re-run the sampler + yaw scan into plan.traj (and plan.yaw / plan.first_yaw when the plan has them); no
        allocation, no sync."""
        if plan.traj is None:
            raise ValueError("a rows-free plan has no row buffer: Engine.sample_rows(plan) allocates one and samples")
        self._bind_stream()
        if plan.yaw is None and getattr(plan, "first_yaw", None) is None:
            self.ctx.call("uavac_minsnap_sample_dev", _ptr(plan.coeffs), _ptr(plan.times), _ptr(plan.seg_rows),
                          _ptr(plan.row_offsets), plan.B, plan.m, plan.dt, _ptr(plan.traj))
        else:
            self.ctx.call("uavac_minsnap_sample_derivs_dev", _ptr(plan.coeffs), _ptr(plan.seg_rows), _ptr(plan.row_offsets),
                          plan.B, plan.m, plan.dt, _ptr(plan.traj), _ptr(plan.yaw), _ptr(getattr(plan, "first_yaw", None)),
                          None, None)

    def check(self, plan: Plan):
        """Raise like the C ABI's host twins would: singular knot systems (repeated waypoints)."""
        if bool((plan.status != 0).any()):
            bad = int((plan.status != 0).nonzero()[0])
            raise nat.UavacError(nat.ESINGULAR, f"mission {bad}: singular knot system (repeated waypoint?)")

    # -- RRT* ---------------------------------------------------------------------
    def rrt_star(self, starts, goals, max_distance: float, samples, obstacles=None) -> "RRTDeviceBatch":
        """B independent RRT* runs (uav_ac/planning/rrt.py `RRTStar.run`), one wavefront each, inputs and results
        resident on the GPU.  `samples` (B, max_iterations, 3): the nodes `_generate_random_node` returns, e.g. from
        `uav_ac.planning.rrt.draw_random_nodes_batch`.  Layouts as documented in include/uavac.h."""
        torch = self._torch
        s = self._dev(starts, torch.float64)
        g = self._dev(goals, torch.float64)
        smp = self._dev(samples, torch.float64)
        if s.dim() != 2 or s.shape[1] != 3 or g.shape != s.shape:
            raise ValueError("starts and goals must both have shape (B, 3)")
        B = int(s.shape[0])
        if smp.dim() != 3 or smp.shape[0] != B or smp.shape[2] != 3 or smp.shape[1] < 1:
            raise ValueError("samples must have shape (B, max_iterations, 3)")
        if not bool(torch.isfinite(s).all() and torch.isfinite(g).all() and torch.isfinite(smp).all()):
            raise ValueError("starts, goals and samples must be finite")
        cub = None if obstacles is None else self._dev(np.asarray(obstacles, dtype=np.float64).reshape(-1, 6), torch.float64)
        n_obs = 0 if cub is None else int(cub.shape[0])
        max_iter = int(smp.shape[1])
        cap = max_iter + 1
        kw = dict(device=self.device)
        nodes = torch.empty((B, cap, 3), dtype=torch.float64, **kw)
        path = torch.empty((B, cap, 3), dtype=torch.float64, **kw)
        canon = torch.empty((B, cap), dtype=torch.int32, **kw)
        parent = torch.empty((B, cap), dtype=torch.int32, **kw)
        best_parent = torch.empty((B, cap), dtype=torch.int32, **kw)
        counts = torch.empty((B, 6), dtype=torch.int32, **kw)
        cost = torch.empty((B,), dtype=torch.float64, **kw)
        self._bind_stream()
        self.ctx.call("uavac_rrt_star_dev", _ptr(s), _ptr(g), B, float(max_distance), max_iter, _ptr(smp),
                      _ptr(cub) if n_obs else None, n_obs, _ptr(nodes), _ptr(canon), _ptr(parent), _ptr(best_parent),
                      _ptr(path), _ptr(counts), _ptr(cost))
        return RRTDeviceBatch(nodes, canon, parent, best_parent, path, counts, cost)

    def rrt_draw_nodes(self, seeds, goals, limits_lw, limits_up, n: int, epsilon: float = 0.15, with_consumed: bool = False):
        """What `RRTStar._generate_random_node` returns in `n` calls after `np.random.seed(seeds[b])`, for B problems,
        generated on the GPU (NumPy's legacy MT19937 stream, bit for bit) -> samples (B, n, 3) [, consumed (B, n)]."""
        torch = self._torch
        g = self._dev(np.round(np.asarray(goals.cpu() if isinstance(goals, torch.Tensor) else goals, dtype=np.float64), 2),
                      torch.float64)
        B = int(g.shape[0])
        sd = np.asarray(seeds, dtype=np.int64).reshape(-1)
        if len(sd) != B or sd.min() < 0 or sd.max() > 0xffffffff:
            raise ValueError("one seed in [0, 2**32) per problem")
        sd_t = torch.as_tensor(sd.astype(np.uint32).view(np.int32), device=self.device)
        lw = np.ascontiguousarray(limits_lw, dtype=np.float64)[:3].copy()
        up = np.ascontiguousarray(limits_up, dtype=np.float64)[:3].copy()
        samples = torch.empty((B, int(n), 3), dtype=torch.float64, device=self.device)
        consumed = torch.empty((B, int(n)), dtype=torch.int64, device=self.device) if with_consumed else None
        self._bind_stream()
        self.ctx.call("uavac_rrt_draw_nodes_dev", _ptr(sd_t), _ptr(g), B, int(n), nat.np_ptr(lw), nat.np_ptr(up), float(epsilon),
                      _ptr(samples), _ptr(consumed))
        return (samples, consumed) if with_consumed else samples

    def rrt_star_seeded(self, starts, goals, space_limits, seeds, max_distance: float, max_iterations: int, obstacles=None,
                        epsilon: float = 0.15) -> "RRTDeviceBatch":
        """B runs of `np.random.seed(seeds[b]); RRTStar(space_limits, starts[b], goals[b], max_distance, max_iterations,
        obstacles).run()` entirely on the GPU: the node draws (`rrt_draw_nodes`) and the planner (`rrt_star`)."""
        goals = np.round(np.asarray(goals, dtype=np.float64), 2)
        samples = self.rrt_draw_nodes(seeds, goals, space_limits[0], space_limits[1], max_iterations, epsilon)
        return self.rrt_star(starts, goals, max_distance, samples, obstacles)

    def rrt_simplify(self, batch: "RRTDeviceBatch", obstacles=None):
        """`RRTStar.simplify_path` (rrt.py:93-116) of every best path of `batch` in one launch.
        -> (paths (B, cap, 3), lengths (B,)) on the GPU; rows past a path's length are zero."""
        torch = self._torch
        B, cap = int(batch.best_path.shape[0]), int(batch.best_path.shape[1])
        cub = None if obstacles is None else self._dev(np.asarray(obstacles, dtype=np.float64).reshape(-1, 6), torch.float64)
        n_obs = 0 if cub is None else int(cub.shape[0])
        lens = batch.counts[:, 4].contiguous()
        out = torch.empty_like(batch.best_path)
        out_lens = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._bind_stream()
        self.ctx.call("uavac_rrt_simplify_dev", _ptr(batch.best_path), _ptr(lens), B, cap, _ptr(cub) if n_obs else None, n_obs,
                      _ptr(out), _ptr(out_lens))
        return out, out_lens

    # -- control ----------------------------------------------------------------
    def fleet(self, plan: Plan, vehicle: Optional[nat.Vehicle] = None, hover: bool = True,
              positions=None, from_plan=None, yaw_from: str = "scan") -> "Fleet":
        from .fleet import Fleet
        return Fleet(self, plan, vehicle, hover, positions, from_plan, yaw_from)
