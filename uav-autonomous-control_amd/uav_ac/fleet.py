"""Batched host API over libuavac.so: plan B missions, fly B UAVs.

This is the batched form of the reference's single-UAV flow in uav_ac/main.py:87-120
(`MinimumSnap(...).get_trajectory()` once per mission, then `TrajectoryController.step()` +
`simulation.step()` per tick).  PyTorch is plumbing only: it owns device memory and the HIP
stream, and `torch.distributed` carries the final gather; all arithmetic happens in the
hand-written HIP kernels behind the C ABI (include/uavac.h).  No CPU fallback exists.

`Fleet` lives here; planning is `uav_ac.engine` (Engine, Plan, RaggedBatch ...), the cut of a job over ranks
`uav_ac.sharding`, the final gather over RCCL `uav_ac.comm`.  Their names are re-exported below for the callers that
knew this module when it held all of them.  The gloo rehearsal of the gather (`uav_ac.comm_host`: host tensors,
multi-process CPU tests) is NOT: the product never imports it.
"""
from __future__ import annotations

import ctypes as C

from . import _native as nat

_P = C.c_void_p


def _torch():
    import torch
    return torch


def _ptr(t) -> _P:
    return _P(0 if t is None else t.data_ptr())
from .engine import Engine, Plan, RaggedBatch, RaggedPlan, RRTDeviceBatch  # noqa: F401  (re-exported)
from .sharding import (DEFAULT_TICK_TABLE, balanced_root_share, candidate_shard_sizes, gather_layout,  # noqa: F401
                       measure_tick_table, shard_bounds, shard_sizes)
from .comm import RcclComm  # noqa: F401


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
        rows_free = getattr(plan, "traj", None) is None             # Engine.plan(..., rows=False): nothing to read rows from
        self.from_plan = (can and (rows_free or plan.B >= self.PLAN_FED_MIN_BATCH)) if from_plan is None else bool(from_plan)
        if self.from_plan and not can:
            raise ValueError("this plan has no coefficients / first headings / yaw column to fly from")
        if rows_free and not self.from_plan:
            raise ValueError("a rows-free plan can only be flown plan-fed (from_plan=True)")
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
        if getattr(self.plan, "traj", None) is None:
            raise ValueError("single ticks read the sampled rows: a rows-free plan flies through rollout()")
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
