"""Batched host API over libuavac.so: plan B missions, fly B UAVs.

This is the batched form of the reference's single-UAV flow in uav_ac/main.py:87-120
(`MinimumSnap(...).get_trajectory()` once per mission, then `TrajectoryController.step()` +
`simulation.step()` per tick).  PyTorch is plumbing only: it owns device memory and the HIP
stream, and `torch.distributed` carries the final gather; all arithmetic happens in the
hand-written HIP kernels behind the C ABI (include/uavac.h).  No CPU fallback exists.
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
    traj: "object"           # (N, 11) f64, missions back to back
    total_rows: int
    yaw: "object" = None     # (N,) f64 or None: the yaw column on its own (== traj[:, 9]); one way to feed the plan-fed rollout
    first_yaw: "object" = None   # (B,) f64: heading of each mission's first row that has one; lets the rollout scan the yaw itself
    placement_ms: "object" = None    # sampler times of the candidate row buffers when plan(..., placement_trials > 1) chose one
    pooled: bool = False             # the rows live in the Engine's pooled buffer (shared with every other pooled plan of that Engine)
    epoch: int = 0                   # bumped whenever the plan is re-solved / re-sampled in place (Engine.replan / solve / sample):
                                     # an attached Fleet then rebuilds the yaw scan it carries instead of trusting a stale one

    def mission(self, b: int) -> np.ndarray:
        """Rows of mission b as a fresh host array (N_b, 11) -- the reference's `full_trajectory`."""
        ro = self.row_offsets[b:b + 2].cpu().numpy()
        return self.traj[int(ro[0]):int(ro[1])].cpu().numpy().copy()

    @property
    def algorithmic_bytes(self) -> int:
        """24(m+1) in + 192 m coefficients out + 88 N rows out, per mission, summed."""
        return self.B * (24 * (self.m + 1) + 192 * self.m) + 88 * self.total_rows


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
    traj: "object"                   # (N, 11) f64
    total_rows: int
    first_yaw: "object"              # (B,) f64
    hit: "object" = None             # (S,) i32 when a cuboid was given

    def mission(self, b: int) -> np.ndarray:
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
             placement_trials: int = 1, pool: bool = False) -> Plan:
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
        previous plan stays consistent and flyable."""
        self._bind_stream()
        self.ctx.call("uavac_minsnap_plan_dev", _ptr(plan.waypoints), plan.B, plan.m, plan.velocity, plan.dt,
                      _ptr(plan.times), _ptr(plan.seg_rows), _ptr(plan.row_offsets), _ptr(plan.coeffs), _ptr(plan.status),
                      _ptr(plan.traj), int(plan.traj.shape[0]), _ptr(plan.yaw), _ptr(plan.first_yaw))
        plan.epoch += 1

    def plan_from_parts(self, coeffs, times, seg_rows, m: int, velocity: float, dt: float, total_rows: int = None,
                        traj=None) -> Plan:
        """A Plan from its solved parts -- coefficients (B, 8m, 3), durations (B, m) or None, rows per spline (B, m) -- e.g.
        the peers' plans after `RcclComm.gather_plan`: row offsets from the row counts (`uavac_minsnap_row_offsets_dev`),
        then the sampler writes the rows (and the first headings).  The rows are a deterministic function of coefficients,
        row counts and dt: bit-identical to the rows of the plan the parts came from.  `total_rows` (when the caller knows
        it) avoids the one host synchronisation that sizes the row buffer; `traj`: a preallocated (>= total, 11) buffer."""
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

    def plan_ragged(self, waypoints, velocity: float = 1.0, dt: float = 0.01, cuboid=None, strict: bool = True) -> RaggedBatch:
        """`MinimumSnap(path_b, None, velocity, dt).get_trajectory()` for B paths of DIFFERENT lengths in one batch
        (minimum_snap.py:13-57 takes any path; `Engine.plan` wants equal lengths).  `waypoints`: B arrays (m_b + 1, 3),
        1 <= m_b <= UAVAC_MAX_SEGMENTS.  Mission b's rows and coefficients equal those of `plan` on it alone, bit for
        bit.  `cuboid` (6,): also return per-spline hit flags (the collision scan of minimum_snap.py:81-87)."""
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
        batch = self._plan_ragged_tensors(wp, so, so_host, max_m, velocity, dt, cuboid)
        if strict:
            self.check(batch)
        return batch

    def _plan_ragged_tensors(self, wp, so, so_host, max_m: int, velocity: float, dt: float, cuboid) -> RaggedBatch:
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
        return Fleet(self, plan, vehicle, hover, positions, from_plan, yaw_from)


class Fleet:
    """B UAVs tracking the B missions of a Plan: batched TrajectoryController + free-flight simulation."""
    PLAN_FED_MIN_BATCH = 18432

    def __init__(self, engine: Engine, plan: Plan, vehicle=None, hover=True, positions=None, from_plan=None,
                 yaw_from: str = "scan"):
        torch = engine._torch
        self.engine, self.plan = engine, plan
        if yaw_from not in ("scan", "column"):
            raise ValueError("yaw_from is 'scan' (the rollout carries the yaw scan) or 'column' (dense plan.yaw)")
        self.yaw_from = yaw_from
        # from_plan: feed the rollout with the plan's coefficients (rows evaluated in the kernel; the yaw scanned by the
        # kernel from plan.first_yaw, or read from the dense column plan.yaw when only that exists)
        # instead of the sampled rows.  Same bits either way.  Default: when the plan carries them (a RaggedPlan does
        # not) and a CU holds more than one workgroup -- measured per 1 000 logged ticks on an MI355X, plan-fed / row-fed
        # (round 4: target rows by the second wave up to two workgroups per CU, coefficients by LDS-DMA above, through
        # registers on a full chip): B = 16 384 0.791 / 0.759 ms, 20 480 0.823 / 0.852, 24 576 0.864 / 0.877, 32 768
        # 0.883 / 0.921, 40 960 0.921 / 1.102, 65 536 1.247 / 1.475, 131 072 2.48 / 2.82 (tools/plan_vs_rows.py,
        # profiles/r04_plan_vs_rows.txt).  With one workgroup per CU reading rows is still a little cheaper than having them
        # evaluated; from two up, the row-fed kernel's 64 scattered row loads per wave and outer tick share the CU's address
        # path with the store waves' log stream (profiles/r04_tick_stamps_*.jsonl) and reading loses.
        # a Plan (one segment count for the whole batch) or a RaggedBatch (seg_offsets); a RaggedPlan has rows only
        can = (hasattr(plan, "coeffs") and (hasattr(plan, "m") or hasattr(plan, "seg_offsets")) and
               (getattr(plan, "first_yaw", None) is not None or getattr(plan, "yaw", None) is not None))
        self.from_plan = (can and plan.B >= self.PLAN_FED_MIN_BATCH) if from_plan is None else bool(from_plan)
        if self.from_plan and not can:
            raise ValueError("this plan has no coefficients / first headings / yaw column to fly from")
        self.vehicle = vehicle if vehicle is not None else nat.Vehicle.default()
        self.B = plan.B
        self.state = torch.empty((nat.STATE_ROWS, self.B), dtype=torch.float64, device=engine.device)
        self.istate = torch.empty((nat.ISTATE_ROWS, self.B), dtype=torch.int32, device=engine.device)
        self._hover = bool(hover)
        if positions is not None:
            self._positions = engine._dev(positions, torch.float64)
        elif hasattr(plan, "start_positions"):
            self._positions = plan.start_positions.contiguous()
        elif getattr(plan, "waypoints", None) is not None:
            self._positions = plan.waypoints[:, 0, :].contiguous()
        else:                                    # a plan assembled from gathered parts: c0 of the first spline IS the first waypoint
            self._positions = plan.coeffs[:, 0, :].contiguous()
        self._plan_epoch = getattr(plan, "epoch", 0)
        self.reset()

    def reset(self):
        """`TrajectoryController.reset` (main.py:29-35) + vehicle back at its first waypoint, at rest."""
        e = self.engine
        e._bind_stream()
        e.ctx.call("uavac_state_init_dev", C.byref(self.vehicle), _ptr(self._positions), self.B,
                   int(self._hover), _ptr(self.state), _ptr(self.istate))
        self._plan_epoch = getattr(self.plan, "epoch", 0)        # the carried yaw scan starts afresh

    def rollout(self, K: int, state_log=None, cmd_log=None, aabbs=None, log_pitch: int = None):
        """K fused ticks.  state_log / cmd_log: None, True (allocate) or a preallocated tensor.

        Layout of the logs: rows are `pitch` doubles apart, [K][13 | 12][pitch], columns B .. pitch-1 never touched.
        * A caller's tensor is written DENSELY ([K][rows][B] in its first K*rows*B elements) unless `log_pitch` says
          otherwise -- then it must hold K*rows*log_pitch elements (log_pitch >= B).  The pitch is never inferred from a
          tensor's shape, and a 3-D tensor whose last dimension is not the pitch that will be written is REFUSED (ValueError):
          indexing it as (K, rows, P) afterwards would read scrambled data.
        * A log allocated here (True) takes the pitch of the caller's other log, else `log_pitch`, else B rounded up to a
          multiple of 16 (rows on 128-byte lines: B = 65 534 at pitch B streams at half the rate of 65 536).
        Returns (state_log, cmd_log) as (K, rows, B) views of pitched buffers (the tensor itself for a caller's dense one), or
        None for a log that was not asked for.
        """
        e, torch = self.engine, self.engine._torch
        B = self.B
        callers = any(t is not None and t is not True for t in (state_log, cmd_log))
        if log_pitch is not None:
            pitch = int(log_pitch)
            if pitch < B:
                raise ValueError(f"log_pitch must be >= B = {B}")
        else:
            pitch = B if callers else -(-B // 16) * 16
        bufs, views = {}, {}
        for name, t, rows in (("state_log", state_log, 13), ("cmd_log", cmd_log, nat.CMD_COLS)):
            if t is None:
                continue
            if t is True:
                t = torch.empty((K, rows, pitch), dtype=torch.float64, device=e.device)
            elif t.dtype != torch.float64 or not t.is_contiguous() or t.numel() < K * rows * pitch:
                raise ValueError(f"{name} must be a contiguous float64 tensor with >= K*{rows}*{pitch} elements")
            elif t.dim() == 3 and t.shape[2] != pitch:
                raise ValueError(f"{name} has rows of {t.shape[2]} doubles but would be written with rows of {pitch}: pass "
                                 f"log_pitch={t.shape[2]} (or a tensor whose last dimension is {pitch})")
            bufs[name] = t
            dense_callers = pitch == B and (state_log if name == "state_log" else cmd_log) is not True
            views[name] = t if dense_callers else t.reshape(-1)[:K * rows * pitch].view(K, rows, pitch)[:, :, :B]
        pitched = pitch != B and bool(bufs)                  # (an unlogged rollout has no pitch to set)
        if pitched:
            e.ctx.set_option("log_pitch", pitch)
        try:
            self._launch_rollout(K, bufs.get("state_log"), bufs.get("cmd_log"), aabbs)
        finally:
            if pitched:                       # the option belongs to this call: other users of the ctx get pitch = B
                e.ctx.set_option("log_pitch", 0)
        return views.get("state_log"), views.get("cmd_log")

    def _launch_rollout(self, K, state_log, cmd_log, aabbs):
        e, torch = self.engine, self.engine._torch
        ab, n_obs = None, 0
        if aabbs is not None:
            ab = e._dev(aabbs, torch.float64).reshape(-1, 6)
            n_obs = int(ab.shape[0])
        e._bind_stream()
        p = self.plan
        if self.from_plan and getattr(p, "epoch", 0) != self._plan_epoch:
            # the plan was re-solved under a flying fleet (Engine.replan / solve without reset()): the yaw scan the vehicles
            # carry (state rows 26-29) belongs to the old coefficients.  Row -1 matches no cursor, so the kernel rebuilds
            # the scan from the mission's first row with the new ones -- what reading a dense yaw column would give.
            self.state[26].fill_(-1.0)
            self._plan_epoch = p.epoch
        if self.from_plan and hasattr(p, "seg_offsets"):
            if self.yaw_from == "column":
                raise ValueError("a ragged batch has no dense yaw column: yaw_from='scan'")
            e.ctx.call("uavac_control_rollout_plan_ragged_dev", C.byref(self.vehicle), _ptr(p.coeffs), _ptr(p.seg_rows),
                       _ptr(p.seg_offsets), _ptr(p.row_offsets), _ptr(p.first_yaw), p.max_m, float(p.dt), _ptr(self.state),
                       _ptr(self.istate), self.B, int(K), _ptr(state_log), _ptr(cmd_log), _ptr(ab), n_obs)
        elif self.from_plan:
            # target rows are evaluated inside the kernel from the plan's coefficients; the yaw is scanned by the kernel
            # (plan.first_yaw) unless only the dense column exists or `yaw_from="column"` was asked for
            first = getattr(p, "first_yaw", None)
            yaw_col = p.yaw if (self.yaw_from == "column" or first is None) else None
            if yaw_col is None and first is None:
                raise ValueError("this plan has neither first headings nor a dense yaw column")
            if self.yaw_from == "column" and yaw_col is None:
                raise ValueError("yaw_from='column' needs a plan made with dense_yaw=True")
            e.ctx.call("uavac_control_rollout_plan_dev", C.byref(self.vehicle), _ptr(p.coeffs), _ptr(p.seg_rows),
                       _ptr(p.row_offsets), _ptr(yaw_col), _ptr(first), p.m, float(p.dt), _ptr(self.state), _ptr(self.istate), self.B,
                       int(K), _ptr(state_log), _ptr(cmd_log), _ptr(ab), n_obs)
        else:
            e.ctx.call("uavac_control_rollout_dev", C.byref(self.vehicle), _ptr(p.traj), _ptr(p.row_offsets),
                       _ptr(self.state), _ptr(self.istate), self.B, int(K), _ptr(state_log), _ptr(cmd_log), _ptr(ab), n_obs)

    def step(self):
        """One tick: `tc.step()` + `simulation.step()` for every UAV (main.py:37-45, mujoco_sim.py:144-151)."""
        e = self.engine
        e._bind_stream()
        e.ctx.call("uavac_control_step_dev", C.byref(self.vehicle), _ptr(self.plan.traj),
                   _ptr(self.plan.row_offsets), _ptr(self.state), _ptr(self.istate), self.B)

    # views -------------------------------------------------------------------------
    @property
    def X(self):
        return self.state[0:13]

    @property
    def trajectory_index(self):
        return self.istate[0]

    @property
    def collided(self):
        return self.istate[2]

    @staticmethod
    def algorithmic_bytes(B: int, K: int, F: int = 10) -> float:
        """104 B state log per UAV tick + one 88 B trajectory row per outer tick."""
        return float(B) * K * (104.0 + 88.0 / F)


# ---------------------------------------------------------------------- multi-GPU
def shard_sizes(B: int, world: int, root_share: float = None, root: int = 0):
    """Missions per rank: contiguous blocks in rank order.  Equal blocks (sizes differ by at most one) unless
    `root_share` is given: then rank `root` -- the rank the trajectories are gathered to -- takes round(root_share * B)
    missions (at least 1 when B >= world) and the other ranks share the rest equally.  The root of the final gather has
    extra work (it re-samples or receives everybody's rows while it flies), so its block is made smaller:
    `balanced_root_share` says by how much."""
    B, world = int(B), int(world)
    if world < 1 or B < 0 or not (0 <= root < world):
        raise ValueError("need world >= 1, B >= 0, 0 <= root < world")
    if root_share is None or world == 1:
        base, rem = divmod(B, world)
        return [base + (1 if r < rem else 0) for r in range(world)]
    if not (0.0 <= root_share <= 1.0):
        raise ValueError("root_share is a fraction of the batch")
    n_root = int(round(root_share * B))
    n_root = max(min(n_root, B), 1 if B >= world else 0)
    n_root = min(n_root, B - (world - 1) if B >= world else n_root)     # every peer keeps at least one mission
    base, rem = divmod(B - n_root, world - 1)
    peers = [base + (1 if i < rem else 0) for i in range(world - 1)]
    return peers[:root] + [n_root] + peers[root:]


def shard_bounds(B: int, rank: int, world: int, root_share: float = None, root: int = 0):
    """Contiguous mission-index block [lo, hi) of this rank: no data-path collective needed.  Sizes: `shard_sizes`."""
    sizes = shard_sizes(B, world, root_share, root)
    lo = sum(sizes[:rank])
    return lo, lo + sizes[rank]


# One MI355X, measured (round 4: profiles/r04_config_sweep.jsonl, tools/rollout_ab.py, m = 8 .. 12): UAVs in flight on the GPU,
# us per logged tick, ms of the planning chain per 1 000 missions.  A FALLBACK: `measure_tick_table` measures the same three
# columns on the GPU at hand in a few tens of milliseconds, and `bench.py --gpus N` does so before it cuts the shards.
DEFAULT_TICK_TABLE = ((4096, 0.787, 0.027), (16384, 0.792, 0.0180), (24576, 0.843, 0.0170), (32768, 0.857, 0.0163), (35237, 0.876, 0.0163),
                      (49152, 1.019, 0.0163), (65536, 1.262, 0.0163))


def measure_tick_table(engine: "Engine", segments: int, sizes, velocity: float = 3.0, dt: float = 0.01, ticks: int = 2500,
                       launches: int = 2, seed: int = 7):
    """What a shard of n missions costs on THIS GPU, for every n in `sizes`: [(n, us per logged tick, ms of the planning chain
    per 1 000 missions)].  Synthetic missions of the SURVEY 8(d) shape, planned once more after a warm-up, then `launches`
    logged launches of `ticks` ticks (the first is thrown away; long launches, as the job itself flies them: a launch boundary
    costs 50-80 us below a full chip).  A few tens of milliseconds per size."""
    torch = engine._torch
    rng = np.random.default_rng(seed)
    table = []
    for n in sorted({int(x) for x in sizes if int(x) > 0}):
        d = rng.standard_normal((n, segments, 3)) * np.array([1, 1, 0.25])
        d /= np.linalg.norm(d, axis=2, keepdims=True)
        w0 = np.concatenate([rng.uniform(0, 24, (n, 1, 1)), rng.uniform(0, 14, (n, 1, 1)), np.full((n, 1, 1), -3.0)], axis=2)
        wps = np.concatenate([w0, w0 + np.cumsum(rng.uniform(2.5, 3.5, (n, segments, 1)) * d, axis=1)], axis=1)
        plan = engine.plan(wps, velocity, dt)
        fleet = engine.fleet(plan)
        pitch = -(-n // 16) * 16
        log = torch.empty((ticks, 13, pitch), dtype=torch.float64, device=engine.device)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        engine.replan(plan)
        ev[0].record()
        engine.replan(plan)
        ev[1].record()
        fleet.reset()
        fleet.rollout(ticks, state_log=log, log_pitch=pitch)
        ev[2].record()
        for _ in range(launches - 1):
            fleet.rollout(ticks, state_log=log, log_pitch=pitch)
        ev[3].record()
        torch.cuda.synchronize(engine.device)
        table.append((n, ev[2].elapsed_time(ev[3]) * 1e3 / ((launches - 1) * ticks), ev[0].elapsed_time(ev[1]) / (n / 1000.0)))
        del plan, fleet, log
    return table


def candidate_shard_sizes(B: int, world: int):
    """The shard sizes worth measuring before `balanced_root_share` cuts a B-mission job over `world` ranks: half an equal block
    (about what the root ends up with), an equal block, and a peer's block when the root takes next to nothing."""
    eq = max(1, B // world)
    return sorted({max(1, eq // 2), eq, min(B, -(-B // max(1, world - 1)))})


def balanced_root_share(B: int, world: int, ticks: int, segments: int, rows_per_segment: float = 112.9,
                        plan_gather: bool = True, hbm_write_bytes_per_s: float = 5.8e12, tick_table=None) -> float:
    """The share of a B-mission job the gather's root should take so that it finishes with its peers (BASELINE configs[3]).

    A PROJECTION from one-GPU measurements, not a measurement of N GPUs: a peer with n missions plans them and flies `ticks`
    logged ticks -- both read off `tick_table` = [(n, us per logged tick, ms of planning per 1 000 missions)], as
    `measure_tick_table` returns it for the GPU at hand (default: `DEFAULT_TICK_TABLE`, round-4 numbers of one MI355X), linear
    between its points, flat below the first, proportional to n above the last; the root does the same for its own block and,
    beside it, receives the peers' plans and re-samples their rows (plan gather) or receives the rows themselves -- either way
    its HBM takes the peers' rows on top of its own log (104 B per UAV tick), so its time is the larger of its flight and of
    (log + all rows) / the HBM write rate.  Bisection on the share."""
    if world <= 1:
        return 1.0
    table = sorted((float(n), float(t), float(p)) for n, t, p in (tick_table or DEFAULT_TICK_TABLE))
    if not table or any(t <= 0 or p <= 0 or n <= 0 for n, t, p in table):
        raise ValueError("tick_table: [(missions, us per tick, ms of planning per 1000 missions)], all positive")
    row_bytes = 88.0 * rows_per_segment * segments                       # per mission

    def lookup(n, col):
        if n <= table[0][0]:
            return table[0][col]
        for lo_, hi_ in zip(table, table[1:]):
            if n <= hi_[0]:
                return lo_[col] + (hi_[col] - lo_[col]) * (n - lo_[0]) / (hi_[0] - lo_[0])
        return table[-1][col] * (n / table[-1][0] if col == 1 else 1.0)   # a full chip walks its tiles pass after pass

    def own(n):                                                          # plan + flight of n missions, seconds
        return lookup(n, 2) * 1e-3 * n / 1000.0 + ticks * lookup(n, 1) * 1e-6

    def root_time(s):
        n = s * B
        stream = (n * ticks * 104.0 + B * row_bytes) / hbm_write_bytes_per_s
        return max(own(n), stream) if plan_gather else own(n) + (B - n) * row_bytes / (7 * 153e9 * min(1.0, (world - 1) / 7.0))

    def peer_time(s):
        return own((1.0 - s) * B / (world - 1))

    lo, hi = 0.0, 1.0 / world
    if root_time(hi) <= peer_time(hi):
        return hi                                                        # equal blocks already balance
    for _ in range(50):
        mid = 0.5 * (lo + hi)
        if root_time(mid) > peer_time(mid):
            hi = mid
        else:
            lo = mid
    return 0.5 * (lo + hi)


def gather_layout(counts, dst: int):
    """Where every rank's block lands in the root's buffer: row offsets (world + 1,) and the peers that send.
    Shared by the RCCL path (whose C side derives the same offsets from the same counts) and the host rehearsal."""
    offs = np.concatenate([[0], np.cumsum(np.asarray(counts, dtype=np.int64))])
    return offs, [r for r in range(len(counts)) if r != dst and counts[r] > 0]


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

    def gather_plan(self, plan: Plan, dst: int = 0, traj=None):
        """The final gather as a gather of the PLAN: every rank sends the coefficients, durations and per-spline row
        counts of its missions (204 B per spline; ~10 KB of rows per spline stay where they are), and `dst` re-samples
        them with the very kernel the peers ran -> (Plan of all missions on dst | None, row counts per rank).  A ragged
        batch (`RaggedBatch`, or the `RaggedPlan` of `plan_collision_free`) travels the same way, with the number of splines
        of every mission as one more column, and comes back as a RaggedBatch.
        `gathered.traj` equals what `gather_rows(plan.traj)` delivers, bit for bit.  Synchronous."""
        return self.gather_finish(self.gather_plan_begin(plan, dst, traj=traj))

    def gather_plan_begin(self, plan: Plan, dst: int = 0, stream=None, traj=None):
        """Enqueue `gather_plan` and return at once (`stream`, ticket, and what may run meanwhile: as for `gather_rows_begin`;
        the root's re-sampling is enqueued on that stream too, behind the receives).  `traj`: a preallocated row buffer for
        the root.  A ragged batch makes the root wait inside this call for the splines-per-mission column (it sizes the
        segment table on the host); a uniform batch returns at once on every rank."""
        e, torch = self.engine, self.engine._torch
        if getattr(plan, "coeffs", None) is None or not plan.coeffs.is_cuda:
            raise ValueError("gather_plan takes a device-resident Plan, RaggedBatch or RaggedPlan with its batch")
        ragged = not hasattr(plan, "m")
        if ragged and getattr(plan, "seg_offsets", None) is None:
            raise ValueError("this plan has neither one segment count for the batch nor seg_offsets")
        m = 0 if ragged else int(plan.m)
        n_seg = int(plan.seg_offsets_host[-1]) if ragged else plan.B * m
        here = torch.cuda.current_stream(e.device)
        stream = here if stream is None else stream
        if stream is not here:
            stream.wait_stream(here)
        with torch.cuda.stream(stream):
            seg_counts = self.counts(n_seg)                  # tiny synchronous all-gathers on that stream
            row_counts = self.counts(plan.total_rows)
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


def gather_plan(plan, dst: int = 0, group=None, comm: "RcclComm" = None, engine: "Engine" = None):
    """`RcclComm.gather_plan` with the host rehearsal path beside it (like `gather_rows`).

    A device-resident Plan + `comm`: RCCL behind the C ABI.  Otherwise the plan's parts travel as HOST tensors through
    `torch.distributed` point-to-point messages (gloo) in the same layout -- the rehearsal path of the multi-process CPU
    tests and of `UAVAC_BENCH_REHEARSAL`; on dst the result is a Plan re-sampled on `engine`'s GPU when one is given, else
    the gathered parts `{"coeffs", "times", "seg_rows", "m"}` as host tensors.  Returns (result | None, row counts)."""
    torch = _torch()
    if comm is not None and getattr(plan.coeffs, "is_cuda", False):
        return comm.gather_plan(plan, dst)
    import torch.distributed as dist
    if not hasattr(plan, "m"):
        raise ValueError("the host (gloo) path of gather_plan takes a Plan with one segment count for the batch; a ragged "
                         "batch travels over RCCL only (RcclComm.gather_plan)")
    m = int(plan.m)
    host = lambda t, dt_: torch.as_tensor(np.asarray(t.cpu() if hasattr(t, "cpu") else t)).to(dt_).contiguous()   # noqa: E731
    co, _ = gather_rows(host(plan.coeffs, torch.float64).reshape(-1, 24), dst, group)
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
