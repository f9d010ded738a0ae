#!/usr/bin/env python3
"""Benchmark of the hot path on BASELINE.json's config 3: batch 65 536 UAVs per GPU, 12-segment
missions (start/end time factor 1.5), minimum-snap solve + sampling followed by 10 000 fused
controller + dynamics ticks, on synthetic missions (SURVEY.md 8(d) generator).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started WITHOUT a launcher (`python bench.py --gpus N`, N > 1, no WORLD_SIZE in the environment) it launches its N ranks
itself: the parent touches no GPU, checks that N GPUs are visible (exit 2 with a message otherwise), starts N children
with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, forwards rank 0's single JSON line and
exits with the worst exit code of the children.

A "step" is one whole job over one batch: the planning chain (times/row counts, offsets, coefficient solve,
sampler; ONE call into the C ABI), vehicle reset, then 10 000 control ticks as 10 launches of 1 000 ticks whose
6.8 GB state log buffer is reused.  Inputs (waypoints) are resident in HBM before the timed region.  The batch
shards by mission index over ranks with no data-path collective (weak scaling: 65 536 UAVs per GPU).

Prints ONE JSON line on rank 0.  `value` = UAV control ticks of all ranks / wall time of the K timed steps
(planning time included in the denominator).  `roofline` prices the dominant kernel (control_rollout) against
the HBM peak with the algorithmic 112.8 B per UAV tick of SURVEY.md 8(d), on the average launch duration
measured with HIP events on the launch stream inside the timed region; `cpu_baseline` is the CPU oracle timed
on this box's host cores on a bounded sample.  `config4` is BASELINE.json configs[3] run on the same N GPUs
(262 144 UAVs in total = strong scaling, 8-segment missions, 5 000 ticks, final gather of the trajectories to
rank 0 over RCCL timed and verified) with compute-only and compute+gather rates side by side, for the gather of the
ROWS (20.8 GB into the root's links) and for the gather of the PLAN (0.4 GB; the root re-samples the peers' rows from
it, bit-identical -- verified against the row gather in the same run).

Exit status: 0 when the line's `value` (configs[2], no collective in its data path) was measured in full AND nothing the
config-4 leg delivered was wrong.  Only a STALLED gather (no completion within 240 s) is soft: the line then carries `gather_error`,
`gather_error_kind: "stall"`, stderr says so, and the exit status is 0 (3 with UAVAC_BENCH_STRICT=1).  Gathered rows that do not match
their source, rows re-sampled from the gathered plan that differ from the gathered rows, an exception inside a collective, a rank
that died: exit status 3, always -- rc 0 means the trajectories on rank 0 were right.
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "uav-autonomous-control_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("UAVAC_AUTOBUILD", "1")       # a never-built checkout compiles the HIP library on first use (no fallback)
_LIB = os.path.join(PKG, "lib", "libuavac.so")
if not os.path.exists(_LIB) and int(os.environ.get("LOCAL_RANK", "0")) != 0:
    for _ in range(1200):                           # ... once: the other ranks of the node wait for local rank 0's build
        if os.path.exists(_LIB):                    # (the Makefile links to a temporary name and renames: it is complete)
            break
        time.sleep(0.5)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
B_PER_GPU = 65536
SEGMENTS = 12
TICKS = 10000
CHUNK = 1000
VELOCITY, DT, F = 3.0, 0.01, 10
FP64_WAVE_INSTR_PEAK = 39.3e12 / 64  # vector fp64 peak of MI355X_MICROARCH.md: 78.6 TFLOP/s = 39.3 T lane-FMA/s = 614 G wave-instr/s
GATHER_TIMEOUT_S = 240
GATHER_STALL_EXIT = 3 if os.environ.get("UAVAC_BENCH_STRICT") == "1" else 0      # a stalled exchange alone is soft (module docstring)
GATHER_FAILURE_EXIT = 3                                                          # wrong data, a failed collective: never rc 0
C4_TOTAL, C4_SEGMENTS, C4_TICKS = 262144, 8, 5000           # BASELINE.json configs[3]
ROLLOUT_SOURCES = ("csrc/control_rollout.hip", "csrc/control_law.h", "csrc/minsnap_eval.h", "csrc/minsnap_yaw.h", "csrc/uavac_internal.h")


def missions(B_total, m, lo, hi):
    """SURVEY.md 8(d) generator restated for the bench (same draw order as the oracle's copy)."""
    rng = np.random.default_rng(20260807 + m)
    d = rng.standard_normal((B_total, m, 3)) * np.array([1, 1, 0.25])
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    L = rng.uniform(2.5, 3.5, (B_total, m, 1))
    w0 = np.concatenate([rng.uniform(0, 24, (B_total, 1, 1)), rng.uniform(0, 14, (B_total, 1, 1)),
                         np.full((B_total, 1, 1), -3.0)], axis=2)
    return np.concatenate([w0, w0 + np.cumsum(L * d, axis=1)], axis=1)[lo:hi]


def rollout_source_sha():
    """Fingerprint of the sources the rollout kernel is compiled from.  Measurements kept under profiles/ (PMC
    traffic, VALU instruction counts) carry it; the bench only quotes them while it still matches."""
    h = hashlib.sha256()
    for rel in ROLLOUT_SOURCES:
        with open(os.path.join(PKG, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def kept_measurement(name, key, kernel=None):
    """A number measured by tools/pmc_traffic.py and committed under profiles/, or None when the
    kernel sources have changed since (or the file is for another kernel)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        rec = json.load(fh)
    if rec.get("rollout_source_sha") != rollout_source_sha():
        return None
    if kernel is not None and rec.get("kernel") not in (None, kernel):
        return None
    return rec.get(key)


PLANNING_SOURCES = ("csrc/minsnap_sample_stream.hip", "csrc/minsnap_solve_tw.hip", "csrc/minsnap_solve_bt.hip", "csrc/minsnap_solve.hip", "csrc/minsnap_eval.h",
                    "csrc/minsnap_yaw.h")


def planning_source_sha():
    import hashlib
    h = hashlib.sha256()
    for rel in PLANNING_SOURCES:
        with open(os.path.join(PKG, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def planning_traffic():
    """Counter bytes of one planning chain (K1 solve + K2 sampler) from profiles/hbm_traffic.json, or None when the planning
    kernels' sources have changed since they were collected."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        rec = json.load(fh)
    if rec.get("planning_source_sha") != planning_source_sha():
        return None
    a, b = rec.get("minsnap_sample_bytes_per_launch"), rec.get("minsnap_solve_bytes_per_launch")
    return (a + b) if (a is not None and b is not None) else None


def nat_build_info():
    from uav_ac import _native as nat
    return nat.lib().uavac_build_info().decode()


def cpu_baseline(eng=None, wps=None):
    """The CPU oracle (test infrastructure; here only as the timed baseline and as the checker, never as
    product) on a bounded sample of the same workload: plan + TICKS control ticks for a few missions.  When an
    engine is passed, the first missions of the bench batch are also planned and flown on the GPU and compared
    with the oracle (SURVEY.md 8(c) metric): the worst relative error goes into the record."""
    from oracle import cpu_baseline as cb
    out = cb.run(segments=SEGMENTS, ticks=TICKS, velocity=VELOCITY, dt=DT)
    if eng is not None:
        from oracle import c_oracle as co
        n, K = 8, 3000
        plan = eng.plan(wps[:n], VELOCITY, DT)
        fleet = eng.fleet(plan)
        slog, _ = fleet.rollout(K, state_log=True)
        slog = slog.cpu().numpy()

        def err(a, b):
            return float(np.max(np.max(np.abs(a - b), axis=0) / np.maximum(1.0, np.max(np.abs(b), axis=0))))
        e_plan = e_ctl = 0.0
        for b in range(n):
            traj, _, _ = co.plan(wps[b], VELOCITY, DT)
            got = plan.mission(b)
            assert got.shape == traj.shape
            e_plan = max(e_plan, err(got, traj))
            state, istate = co.initial_state(traj[0, 0:3])
            s_ref, _ = co.rollout(traj, state, istate, K, log_cmd=False)
            e_ctl = max(e_ctl, err(slog[:, :, b], s_ref))
        out["max_rel_err_vs_oracle"] = {"trajectory_rows": e_plan, "state_log": e_ctl, "missions": n, "ticks": K,
                                        "tolerance": 1e-5}
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks from here.  The parent never touches a GPU
    (counting devices does not initialise one); children are separate processes, nothing is exec'ed."""
    import socket
    import subprocess
    n = args.gpus
    if not args.launch_check:
        import torch
        have = torch.cuda.device_count()
        shared = os.environ.get("UAVAC_BENCH_REHEARSAL") == "1"
        if have < (1 if shared else n):
            print(f"bench.py: --gpus {n} but {have} GPU(s) visible; refusing to print a line for fewer GPUs than asked for",
                  file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # rank 0's stdout carries the JSON line; whatever the other ranks print goes to this process's stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(), text=(r == 0)))
    out0, _ = procs[0].communicate()                            # rank 0 prints the line last and leaves
    codes = [procs[0].returncode] + [None] * (n - 1)
    deadline = time.time() + 120.0                              # the other ranks get two minutes to follow it out
    while any(c is None for c in codes):
        for r in range(1, n):
            if codes[r] is None:
                codes[r] = procs[r].poll()
        if time.time() > deadline:
            for r in range(1, n):
                if codes[r] is None:
                    procs[r].kill()                             # exactly the children started above
                    codes[r] = procs[r].wait() or 9
        time.sleep(0.05)
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    worst = 0
    for c in codes:
        c = 128 - c if c < 0 else c                             # killed by a signal
        worst = max(worst, c)
    return worst


def launch_check(args):
    """`--launch-check`: the rank side of the self-launcher without any GPU work -- rendezvous over gloo, one barrier,
    one reduction, the single JSON line from rank 0.  What the CPU test suite runs through `self_launch`."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist.init_process_group("gloo")
    dist.barrier()
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    fail = os.environ.get("UAVAC_BENCH_FAIL_RANK")
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": int(t.item()),
                          "local_ranks_seen": world, "master": os.environ.get("MASTER_ADDR")}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if fail is not None and int(fail) == rank:
        sys.exit(3)


def yaw_fork_census(eng, n=1024):
    """How often does the yaw column fork from NumPy's by a multiple of 2 pi?  Where the horizontal velocity reverses through
    (almost) zero the heading steps by pi to the last bit, and whether np.unwrap then adds 2 pi hangs on the rounding of two
    atan2 results -- the GPU's here, NumPy's in the reference.  Counted on `n` missions of the 8(d) distribution and
    of the stress distribution U(1, 6) m: the NumPy oracle's yaw scan (np.arctan2 + np.unwrap, hold, back-fill) is run on the
    GPU rows' OWN velocities, so any difference is the yaw step's alone.  Part of the checker leg (oracle import)."""
    from oracle import minsnap_oracle as mo
    out = {}
    for name, lo, hi in (("8d_U(2.5,3.5)", 2.5, 3.5), ("stress_U(1,6)", 1.0, 6.0)):
        rng = np.random.default_rng(20260807 + SEGMENTS)
        d = rng.standard_normal((n, SEGMENTS, 3)) * np.array([1, 1, 0.25])
        d /= np.linalg.norm(d, axis=2, keepdims=True)
        L = rng.uniform(lo, hi, (n, SEGMENTS, 1))
        w0 = np.concatenate([rng.uniform(0, 24, (n, 1, 1)), rng.uniform(0, 14, (n, 1, 1)), np.full((n, 1, 1), -3.0)], axis=2)
        wps = np.concatenate([w0, w0 + np.cumsum(L * d, axis=1)], axis=1)
        plan = eng.plan(wps, VELOCITY, DT)
        rows = plan.traj.cpu().numpy()
        ro = plan.row_offsets.cpu().numpy()
        forks = worst_other = 0
        for b in range(n):
            r = rows[ro[b]:ro[b + 1]]
            ref = mo.yaws_from_velocity(r[:, 3:6])
            diff = np.abs(r[:, 9] - ref)
            turns = np.round(diff.max() / (2 * np.pi))
            if turns >= 1:
                forks += 1
            else:
                worst_other = max(worst_other, float(diff.max()))
        out[name] = {"missions": n, "missions_2pi_apart_from_numpy": int(forks), "max_abs_diff_elsewhere_rad": worst_other}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="UAVs per GPU (default: config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config4", action="store_true", help="skip the BASELINE configs[3] leg")
    ap.add_argument("--equal-shards", action="store_true", help="configs[3]: equal blocks per rank instead of a smaller block for the gather's root")
    ap.add_argument("--root-share", type=float, default=None, help="configs[3]: the share of the missions rank 0 (the gather's root) takes, "
                                                                  "instead of the balance measured at start-up; 0 = it flies nothing")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed diagnostic passes (per-launch table, "
                                                             "flyable-distribution rate)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous + one JSON line, no GPU work (tests of the "
                                                                "self-launcher)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 or os.environ.get("UAVAC_BENCH_FORCE_DIST") == "1":
        # HIP maps a process's streams round-robin onto a few hardware queues (default 4), and two streams that share a queue run
        # one after the other.  The config-4 leg of a multi-rank run uses a handful of them at once on the root -- its flight, the
        # transfers, the sampling launches -- and a collision there serialises flight and sampling (8.4 instead of 5.1 ms on one
        # MI355X, profiles/r06_hw_queue_collisions.txt).  Must be set before the HIP runtime comes up; ranks started by a launcher
        # set it themselves right here, children of the self-launcher inherit it.  (N = 1: the environment is left as it is.)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))          # no launcher around us: be the launcher (before any GPU call)
    if args.launch_check:
        return launch_check(args)

    import torch
    import torch.distributed as dist
    from uav_ac.fleet import Engine, Fleet, RcclComm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:                                 # never a line whose n_gpus differs from what was asked for
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # UAVAC_BENCH_REHEARSAL=1: several ranks share one GPU (RCCL refuses that), the process group is gloo and the gather
    # takes the host path of uav_ac.fleet.gather_rows on a slice of the rows.  It exists to walk the N > 1 control flow of
    # this file on a 1-GPU box; its numbers mean nothing and the line says so.  The driver's runs never set it.
    rehearsal = world > 1 and os.environ.get("UAVAC_BENCH_REHEARSAL") == "1"
    # UAVAC_BENCH_FORCE_DIST=1 at world 1: take the N > 1 path anyway -- the RCCL process group, the communicator behind
    # the C ABI and the gather (to this very rank) all run on one GPU.  tests/test_gpu_round2.py uses it; the line is
    # marked.  The driver's runs never set it.
    forced = world == 1 and os.environ.get("UAVAC_BENCH_FORCE_DIST") == "1"
    multi = world > 1 or forced
    if rehearsal:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if rehearsal else dev       # where the tensors of the (tiny) collectives live
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced:
            for k, v in (("MASTER_PORT", "29531"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # barrier + max-over-ranks only; the gather has its own comm

    B, m = args.batch, SEGMENTS
    eng = Engine(dev)
    wps = missions(B * world, m, rank * B, (rank + 1) * B)
    # The row buffer is the FIRST allocation: no placement search (Engine.plan's default; what kind of buffer that is, is the
    # box's choice: DESIGN K2, NOTES R4-6).
    plan = eng.plan(wps, VELOCITY, DT)                               # allocates; also the first warm-up
    fleet = eng.fleet(plan)
    log = torch.empty((CHUNK, 13, B), dtype=torch.float64, device=dev)
    n_chunks = TICKS // CHUNK
    ev = lambda: torch.cuda.Event(enable_timing=True)    # noqa: E731  (torch's current stream = the ctx's stream)

    def one_step(record=None):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        eng.replan(plan)                                 # 4 launches enqueued by one C call
        e1.record()
        fleet.reset()
        for _ in range(n_chunks):
            fleet.rollout(CHUNK, state_log=log)
        e2.record()
        if record is not None:
            record.append((e0, e1, e2))

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    rec = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(rec)
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    flags = eng.take_flags()
    assert flags == [0, 0, 0, 0], f"planning flags raised: {flags}"

    plan_ms = [a.elapsed_time(b) for a, b, _ in rec]
    # e1 -> e2 spans reset (a 5 us kernel) + the n_chunks rollout launches (each preceded by its ~2 us aligner launch);
    # their share is part of the average on purpose: it is what a launch costs in the job
    roll_avg_s = float(np.mean([b.elapsed_time(c) for _, b, c in rec])) * 1e-3 / n_chunks
    plan_avg_s = float(np.mean(plan_ms)) * 1e-3
    plan_med_s = float(np.median(plan_ms)) * 1e-3
    kernel_name = eng.ctx.last_rollout_kernel()

    # sanity inside the bench: after TICKS ticks every cursor must sit at min(TICKS/F, rows-1) exactly; the share of
    # UAVs within the reference's 0.5 m acceptance of their current target row is reported, and so is the state of the
    # others: the reference law itself loses about 1 in 7 of the 8(d) missions (3 m legs with random turns at 3 m/s;
    # the CPU oracle loses the same lanes, tests/test_gpu_control.py) -- do they go non-finite, do they leave the
    # small-angle branch of the attitude update (|w| dt / 2 > 0.0316, control_law.h), i.e. do they cost more per tick?
    nrows = plan.row_offsets[1:] - plan.row_offsets[:-1]
    idx = fleet.trajectory_index.long()
    cursor_ok = bool((idx == torch.clamp(torch.full_like(nrows, TICKS // F), max=nrows - 1)).all())
    target = plan.traj[plan.row_offsets[:-1] + idx, 0:3].T
    track = (fleet.X[0:3] - target).norm(dim=0)
    kept = track < 0.5
    frac_kept = float(kept.double().mean())
    lane_finite = torch.isfinite(fleet.state).all(dim=0)
    body_rate = fleet.X[10:13].norm(dim=0)
    large_angle = body_rate * (0.5 * fleet.vehicle.dt) > 0.0316
    checks = {"frac_uavs_within_0.5m_of_target_row": frac_kept,
              "tracking_lanes_finite": bool(lane_finite[kept].all()),
              "all_trajectory_cursors_exact": cursor_ok,
              "lanes_nonfinite_at_end": int((~lane_finite).sum()),
              "lanes_in_large_angle_branch_at_end": int((large_angle & lane_finite).sum()),
              "max_body_rate_rad_s_at_end": float(body_rate[lane_finite].max()),
              "planning_flags": flags}

    out = None
    if rank == 0:
        total_ticks = float(world) * B * TICKS * args.steps
        value = total_ticks / elapsed
        roll_bytes = Fleet.algorithmic_bytes(B, CHUNK, F)
        achieved = roll_bytes / roll_avg_s / 1e9
        same_size = B == B_PER_GPU
        traffic = kept_measurement("hbm_traffic.json", "control_rollout_bytes_per_launch", kernel_name) if same_size else None
        valu = kept_measurement("hbm_traffic.json", "control_rollout_valu_wave_insts_per_launch", kernel_name) if same_size else None
        plan_traffic = planning_traffic() if same_size else None
        out = {
            "metric": "UAV control-steps/sec at batch=65536",
            "value": value,
            "unit": "UAV control-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            **({"REHEARSAL": "UAVAC_BENCH_REHEARSAL=1: ranks share one GPU; not a measurement"} if rehearsal else {}),
            **({"FORCED_DIST": "UAVAC_BENCH_FORCE_DIST=1: the N > 1 path (RCCL process group, communicator, gather) at world 1"}
               if forced else {}),
            "config": {"workload": "BASELINE.json configs[2]: batch 65536 UAVs/GPU, 12-segment missions with start/end "
                                   "time factor 1.5, min-snap solve+sample then 10000 fused controller+dynamics ticks "
                                   "(10 launches x 1000 ticks, 13-f64 state logged every tick)",
                       "batch_per_gpu": B, "segments": m, "ticks": TICKS, "ticks_per_launch": CHUNK,
                       "velocity": VELOCITY, "dt": DT, "inner_per_outer": F, "rows": plan.total_rows,
                       "parallelism": f"missions sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("profiles/hbm_traffic.json (tools/pmc_traffic.py; same kernel sources)"
                                            if traffic is not None else None),
                         # the same launch priced by the bytes the counters saw it move (WRITE_SIZE + 2 x FETCH_SIZE): the plan-fed
                         # kernel evaluates its target rows instead of reading them, so it moves less than the algorithmic figure
                         "frac_counter_bytes": (traffic / roll_avg_s / 1e9 / HBM_PEAK_GBS) if traffic is not None else None,
                         "algorithmic_bytes_per_launch": roll_bytes, "avg_launch_ms": roll_avg_s * 1e3,
                         "kernel_vgprs": eng.ctx.last_rollout_vgprs(),
                         "rollout_source_sha": rollout_source_sha()},
            "build": nat_build_info(),
            # BASELINE's second metric.  `value`, `ms_solve_plus_sample` and `roofline.frac` are on the MEAN of the timed steps'
            # planning chains -- what rounds 1-4 reported and what `value` above paid; round 5 had silently moved these keys to the
            # median (round-5 advice).  The median -- one step in twenty that follows a host-side hiccup can take 4x as long and move a
            # mean by 14 % -- is beside them under its own names: `value_median`, `ms_median`, `frac_median`, `roofline.frac_median`.
            "minsnap": {"metric": "min-snap segments solved/sec", "value": B * m / plan_avg_s, "unit": "segments/s",
                        "statistic": "value / ms_solve_plus_sample / roofline.frac: MEAN over the timed steps (as in rounds 1-4); *_median: their median",
                        "ms_solve_plus_sample": plan_avg_s * 1e3, "ms_mean": plan_avg_s * 1e3,
                        "value_median": B * m / plan_med_s,
                        "ms_per_timed_step": [round(x, 4) for x in plan_ms],
                        # the planning chain of every timed step (HIP events round the one C call)
                        "ms_median": float(np.median(plan_ms)), "ms_min": float(np.min(plan_ms)), "ms_max": float(np.max(plan_ms)),
                        "frac_median": plan.algorithmic_bytes / plan_med_s / 1e9 / HBM_PEAK_GBS,
                        "row_buffer": "first allocation (Engine.plan's default: no placement search)",
                        "solve": "two-ended block-Thomas, two lanes per mission (csrc/minsnap_solve_tw.hip; option solve_order = 1, the default)",
                        "roofline": {"bound": "hbm", "achieved": plan.algorithmic_bytes / plan_avg_s / 1e9,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": plan.algorithmic_bytes / plan_avg_s / 1e9 / HBM_PEAK_GBS,
                                     "frac_median": plan.algorithmic_bytes / plan_med_s / 1e9 / HBM_PEAK_GBS,
                                     "algorithmic_bytes": plan.algorithmic_bytes,
                                     "traffic": plan_traffic,
                                     "frac_counter_bytes": (plan_traffic / plan_avg_s / 1e9 / HBM_PEAK_GBS)
                                     if plan_traffic is not None else None,
                                     "traffic_source": ("profiles/hbm_traffic.json: K1 + K2 counter bytes per planning chain "
                                                        "(same sampler / solver sources)") if plan_traffic is not None else None},
                        },
            "rollout_only": {"value": B * CHUNK / roll_avg_s, "unit": "UAV control-steps/s per GPU",
                             "note": "SURVEY 8(d)(i): B x K / time of the rollout launches alone (`value` above also "
                                     "carries the planning time of every step)"},
            "checks": checks,
        }
        if valu is not None:
            # every VALU instruction of the launch (SQ_INSTS_VALU; three quarters of them are fp64 in the ISA) priced at
            # the fp64 issue rate: an upper bound of the fp64 pipe's utilisation
            rate = valu / roll_avg_s
            out["roofline"]["valu"] = {"wave_instr_per_launch": valu, "per_uav_tick": valu / (B / 64.0) / CHUNK,
                                       "achieved": rate, "peak": FP64_WAVE_INSTR_PEAK, "unit": "VALU wave-instr/s",
                                       "frac": rate / FP64_WAVE_INSTR_PEAK,
                                       "source": "profiles/hbm_traffic.json (tools/pmc_traffic.py, SQ_INSTS_VALU)"}

    # The headline exists: say so on stderr at once, before the diagnostics and before any config-4 collective -- a crash inside
    # RCCL on first contact with real peers cannot lose it (stdout still carries exactly ONE line, at the end).
    if rank == 0:
        print(json.dumps({"early": True, "note": "headline measured; diagnostics and the config-4 leg follow; the full line goes to stdout",
                          **{k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype")},
                          "roofline_frac": out["roofline"]["frac"], "minsnap_segments_per_s": out["minsnap"]["value"]}),
              file=sys.stderr, flush=True)

    # ---- untimed diagnostics: per-launch durations of one more step, and the rollout on a flyable distribution -------
    if not args.no_extras:
        for attempt in range(2):                      # the first pass creates its 20 events (slow) and is thrown away
            eng.replan(plan)
            fleet.reset()
            pairs = []
            for _ in range(n_chunks):
                a, b = ev(), ev()
                a.record()
                fleet.rollout(CHUNK, state_log=log)
                b.record()
                pairs.append((a, b))
            torch.cuda.synchronize()
        per_launch = [a.elapsed_time(b) for a, b in pairs]
        # the sampler alone on the bench's row buffer, both kernels: the default (chunk-streaming, 4 waves per workgroup) and
        # the one-wave-per-mission kernel, whose time depends on where the buffer lies
        sampler_ms = {}
        for label, waves in (("chunk_streaming_w4", 4), ("one_wave_per_mission", 1)):
            eng.ctx.set_option("sampler_waves", waves)
            for _ in range(3):
                eng.sample(plan)
            a, b = ev(), ev()
            a.record()
            for _ in range(5):
                eng.sample(plan)
            b.record()
            torch.cuda.synchronize()
            sampler_ms[label] = round(a.elapsed_time(b) / 5, 4)
        eng.ctx.set_option("sampler_waves", 4)
        # The same steps once more with the shader clock beside them: one extra wavefront on a side stream stamps s_memtime /
        # s_memrealtime over a millisecond of every planning chain (uavac_clock_probe_dev; it starts a few tens of microseconds
        # into the chain, when the host gets to it).  Untimed: the steps `value` was measured on carry no probe.  Enqueued like
        # the timed steps -- back to back, nothing synchronises in between -- so `ms_median` of this leg against the timed leg's is
        # the instrument's leg-to-leg repeatability on one row buffer.  (No probe beside the ROLLOUT of a full chip: its workgroups
        # fill every SIMD's registers to the last one, and the workgroup whose SIMD the probe's wave holds starts a probe late.)
        side = torch.cuda.Stream(device=dev)
        clocked = {"planning_ms": [], "planning_clock_ghz": []}
        try:
            marks = []
            for i in range(10):
                e0, e1 = ev(), ev()
                e0.record()
                eng.replan(plan)
                e1.record()
                side.wait_event(e0)
                probe = eng.clock_probe_begin(1000, stream=side)
                fleet.reset()
                for _ in range(n_chunks):
                    fleet.rollout(CHUNK, state_log=log)
                torch.cuda.current_stream(dev).wait_stream(side)          # the next step's planning starts behind this probe's end
                marks.append((e0, e1, probe))
            torch.cuda.synchronize()
            for e0, e1, probe in marks[2:]:
                clocked["planning_ms"].append(round(e0.elapsed_time(e1), 4))
                clocked["planning_clock_ghz"].append(round(eng.clock_probe_ghz(probe), 3))
            clocked["ms_median"] = float(np.median(clocked["planning_ms"]))
            clocked["clock_ghz"] = float(np.median(clocked["planning_clock_ghz"]))
        except Exception as exc:                              # an extra: the line's measurements above must survive it
            clocked["error"] = f"{type(exc).__name__}: {exc}"
        # the same missions flown at half the speed (velocity 1.5: legs demand < 2.5 m/s^2): nobody departs
        slow = eng.plan(wps, VELOCITY / 2, DT, placement_trials=1)
        fl2 = eng.fleet(slow)
        a, b = ev(), ev()
        for _ in range(2):
            fl2.rollout(CHUNK, state_log=log)
        a.record()
        for _ in range(n_chunks):
            fl2.rollout(CHUNK, state_log=log)
        b.record()
        torch.cuda.synchronize()
        idx2 = fl2.trajectory_index.long()
        tgt2 = slow.traj[slow.row_offsets[:-1] + idx2, 0:3].T
        kept2 = float(((fl2.X[0:3] - tgt2).norm(dim=0) < 0.5).double().mean())
        if rank == 0:
            out["roofline"]["per_launch_ms_one_step"] = [round(x, 4) for x in per_launch]
            out["minsnap"]["sampler_ms_on_this_row_buffer"] = sampler_ms
            out["minsnap"]["clocked_steps"] = clocked
            if "clock_ghz" in clocked:
                out["minsnap"]["clock_ghz"] = clocked["clock_ghz"]
                out["minsnap"]["leg_to_leg"] = clocked["ms_median"] / out["minsnap"]["ms_median"]
            out["rollout_only_flyable"] = {"value": B * CHUNK * n_chunks / (a.elapsed_time(b) * 1e-3),
                                           "unit": "UAV control-steps/s per GPU", "velocity": VELOCITY / 2,
                                           "frac_uavs_within_0.5m_of_target_row": kept2,
                                           "note": "same missions at half the cruise speed, where the reference law "
                                                   "keeps every UAV on its trajectory"}
        del slow, fl2

    # ---- BASELINE configs[3] on the same N GPUs: 262 144 UAVs in total, m = 8, 5 000 ticks, gather to rank 0 ----------
    # Two forms of the job, side by side in the line:
    #   LITERAL  every rank plans WITH rows and flies, then the rows travel to rank 0 (uavac_gather_rows_dev: 18-20 GB into the
    #            root's links) -- or, round 3, the plan travels and rank 0 re-samples (the peers' rows are then sampled twice);
    #   ROUND 6  every rank plans ROWS-FREE (times, row counts, solve, first headings: not one row), flies plan-fed and ships its
    #            plan in parts; rank 0 samples everybody's rows ONCE, part by part as they arrive, beside its own flight
    #            (`config4.end_to_end`).  Same rows on rank 0, bit for bit -- verified against the literal row gather in this run.
    gather_err = None
    comm = None
    if not args.no_config4:
        del plan, fleet, log
        torch.cuda.empty_cache()
        from uav_ac.sharding import balanced_root_share, candidate_shard_sizes, measure_tick_table, shard_bounds, shard_sizes
        # contiguous blocks; rank 0 -- the root of the final gather, which writes everybody's rows into its HBM while it flies --
        # gets a smaller block so that it finishes with its peers.  How much smaller follows from what a shard of n missions costs
        # on THESE GPUs: every rank measures four candidate sizes (a few tens of ms), the slowest rank's numbers count, and the
        # table goes into the line (`config4.tick_table`).  UAVAC_BENCH_TICK_TABLE='[[n, us_per_tick, plan_ms_per_1000, rows_free_plan_ms_per_1000], ...]'
        # injects a table instead (tests).
        root_share, tick_table = None, None
        if world > 1 and args.root_share is not None:
            root_share = float(args.root_share)
        elif world > 1 and not args.equal_shards:
            injected = os.environ.get("UAVAC_BENCH_TICK_TABLE")
            if injected:
                tick_table = [tuple(float(v) for v in row) for row in json.loads(injected)]
            else:
                sizes_c = candidate_shard_sizes(C4_TOTAL, world)
                # (a rank whose measurement fails must not leave the others waiting in the reduction: every rank contributes a
                # table of the agreed shape, a failed one as +inf, and any +inf sends ALL ranks to the built-in table)
                try:
                    mine = measure_tick_table(eng, C4_SEGMENTS, sizes_c, VELOCITY, DT)
                    if [row[0] for row in mine] != sizes_c:
                        raise RuntimeError("tick table of another shape")
                except Exception as exc:
                    print(f"bench.py: rank {rank}: measuring the tick table failed ({type(exc).__name__}: {exc}); built-in table", file=sys.stderr, flush=True)
                    mine = [(n, float("inf"), float("inf"), float("inf")) for n in sizes_c]
                    torch.cuda.empty_cache()
                t = torch.tensor([list(row[1:]) for row in mine], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                if bool(torch.isinf(t).any()):
                    tick_table = None                      # balanced_root_share falls back to DEFAULT_TICK_TABLE
                else:
                    tick_table = [(row[0],) + tuple(float(v) for v in cols) for row, cols in zip(mine, t.cpu().tolist())]
            root_share = balanced_root_share(C4_TOTAL, world, C4_TICKS, C4_SEGMENTS, tick_table=tick_table)
        sizes4 = shard_sizes(C4_TOTAL, world, root_share, 0)
        lo4, hi4 = shard_bounds(C4_TOTAL, rank, world, root_share, 0)
        B4 = hi4 - lo4
        wps4 = missions(C4_TOTAL, C4_SEGMENTS, lo4, hi4)
        if B4 > 0:
            plan4 = eng.plan(wps4, VELOCITY, DT, placement_trials=1)          # the literal form: this rank's rows are sampled here
            plan4f = eng.plan(wps4, VELOCITY, DT, rows=False)                   # round 6: no rows here; rank 0 samples them all
            fleet4 = eng.fleet(plan4)
            fleet4f = eng.fleet(plan4f)                                         # plan-fed: all a rows-free plan can be
        else:
            # The root of an 8-rank job flies NOTHING when the measured table says so (`balanced_root_share` returned 0): sampling
            # 262 144 missions' rows is as much work as a peer's flight, and a flight beside its own sampler runs a quarter slower.
            # It takes part in every gather with an empty block.
            plan4, plan4f = eng.empty_plan(C4_SEGMENTS, VELOCITY, DT), eng.empty_plan(C4_SEGMENTS, VELOCITY, DT, rows=False)
            fleet4 = fleet4f = None
        pitch4 = -(-B4 // 16) * 16                                                               # rows on 128-byte lines for any B4
        # The 5 000 ticks in as few launches as a 40 GB log allows: ONE at 8 ranks (17-19 GB), two at 2 ranks, five on one GPU.  A
        # launch boundary costs 50-80 us below a full chip (every workgroup waits for the slowest; prologue; aligner): 32 768 UAVs fly
        # the 5 000 ticks in 4.19 ms as one launch against 4.48 as five (profiles/r05_config4_chunking.jsonl).
        chunk4 = next(c for c in (5000, 2500, 1000) if c == 1000 or c * 13 * pitch4 * 8 <= 40e9)
        log4 = torch.empty((chunk4, 13, pitch4), dtype=torch.float64, device=dev)

        def fly4(fl=fleet4, started=None):
            if fl is None:                                   # (a rank without missions)
                return
            fl.reset()
            if started is not None:                          # (an event behind the reset: the rollout's launches follow it at once)
                started.record()
            for _ in range(C4_TICKS // chunk4):
                fl.rollout(chunk4, state_log=log4, log_pitch=pitch4)

        def step4():
            eng.replan(plan4)
            fly4()

        def step4f():
            eng.replan(plan4f)
            fly4(fleet4f)

        def timed(fn, reps):
            """max over ranks of the mean wall time of `fn`, bracketed by barriers"""
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            barrier()
            dt_ = (time.perf_counter() - t0) / reps
            if multi:
                t = torch.tensor([dt_], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_ = float(t.item())
            return dt_

        step4()
        c4_compute = timed(step4, 3)
        kernel4 = eng.ctx.last_rollout_kernel()
        c4_fly = timed(fly4, 3)                              # the 5 000 ticks alone (planning is 12 % of this job)
        step4f()
        c4_compute_free = timed(step4f, 3)
        kernel4f = eng.ctx.last_rollout_kernel()
        # what re-sampling this rank's rows from its plan costs (at N = 1: all 262 144 missions = what the root of the plan
        # gather pays at any N, on top of its own share of the flight), and the two planning chains on their own
        a, b, c, d = ev(), ev(), ev(), ev()
        if B4 > 0:
            eng.sample(plan4)
        eng.replan(plan4f)
        a.record()
        for _ in range(3 if B4 > 0 else 0):
            eng.sample(plan4)
        b.record()
        for _ in range(3):
            eng.replan(plan4)
        c.record()
        for _ in range(3):
            eng.replan(plan4f)
        d.record()
        torch.cuda.synchronize()
        c4 = {"workload": "BASELINE.json configs[3]: 262144 UAVs in total (strong scaling), 8-segment missions, plan + "
                          "5000 fused ticks (state logged every tick; as few launches as a 40 GB log per rank allows: `ticks_per_launch`), "
                          "trajectories gathered to rank 0",
              "batch_total": C4_TOTAL, "batch_per_gpu": B4, "shard_sizes": sizes4, "root_share": root_share,
              "tick_table": ([[int(row[0])] + [round(v, 7) for v in row[1:]] for row in tick_table] if tick_table else None),
              "tick_table_columns": "missions on the GPU, us per logged tick, ms of planning per 1000 missions with rows, the same rows-free (max over ranks)",
              "log_pitch": pitch4, "ticks_per_launch": chunk4, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
              "segments": C4_SEGMENTS, "ticks": C4_TICKS,
              "rows_rank0": plan4.total_rows, "compute_ms": c4_compute * 1e3,
              "steps_per_s_compute_only": C4_TOTAL * C4_TICKS / c4_compute, "rollout_kernel": kernel4,
              "rollout_ms": c4_fly * 1e3, "steps_per_s_rollout_only": C4_TOTAL * C4_TICKS / c4_fly,
              # round 6: the same job with the rows-free planning chain on every rank (what `end_to_end` runs)
              "compute_rows_free_ms": c4_compute_free * 1e3, "steps_per_s_compute_only_rows_free": C4_TOTAL * C4_TICKS / c4_compute_free,
              "rollout_kernel_rows_free": kernel4f,
              "plan_chain_rank0_ms": {"with_rows": b.elapsed_time(c) / 3, "rows_free": c.elapsed_time(d) / 3},
              "resample_rank0_rows_ms": a.elapsed_time(b) / 3,
              "plan_bytes_rank0": int(plan4.B * C4_SEGMENTS * 204), "row_bytes_rank0": int(plan4.total_rows * 88)}
        if rank == 0:
            out["config4"] = c4

        if multi:
            def bail():
                # a stalled exchange must not lose the measurement -- and must not look like a success either
                if rank == 0:
                    out["gather_error"] = f"no completion within {GATHER_TIMEOUT_S} s"
                    out["gather_error_kind"] = "stall"
                    print(json.dumps(out), flush=True)
                    print(f"bench.py: config-4 gather stalled ({GATHER_TIMEOUT_S} s); the headline was measured before it", file=sys.stderr, flush=True)
                os._exit(GATHER_STALL_EXIT)
            watchdog = threading.Timer(GATHER_TIMEOUT_S, bail)
            watchdog.daemon = True
            watchdog.start()

            def everybody_fine(err):               # the same collective on every rank, whatever happened locally
                t = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return int(t.item()) == 0

            try:
                SLICE = 200000                             # rehearsal: rows per rank that cross gloo (control flow only)
                if rehearsal:
                    from uav_ac.comm_host import gather_plan, gather_rows      # host tensors over gloo: the rehearsal transport

                    def rows_gather():
                        g, c = gather_rows(plan4.traj[:SLICE].cpu(), dst=0)
                        return (g.to(dev) if g is not None else None), c

                    def plan_gather(p=plan4, parts=None):
                        return gather_plan(p, dst=0, engine=eng, parts=parts)
                else:
                    comm = RcclComm(eng)                   # ncclCommInitRank behind the C ABI; id travels over the process group
                    comm.gather_rows(plan4.traj[:1024], dst=0)      # connection set-up is not part of any timed gather
                    rccl_ranks, rccl_rank = comm.shape()            # what the communicator itself says (ncclCommCount / UserRank)
                    if rank == 0:
                        c4["rccl_ranks"] = rccl_ranks
                        built, runtime = ctypes.c_int(), ctypes.c_int()
                        from uav_ac import _native as nat
                        nat.lib().uavac_comm_versions(ctypes.byref(built), ctypes.byref(runtime))
                        c4["rccl_versions"] = {"built_with": int(built.value), "runtime": int(runtime.value)}
                    if (rccl_ranks, rccl_rank) != (world, rank):
                        raise RuntimeError(f"the RCCL communicator has {rccl_ranks} ranks / this is its rank {rccl_rank}; the process group says {world} / {rank}")

                    def rows_gather():
                        return comm.gather_rows(plan4.traj, dst=0)

                    def plan_gather(p=plan4, parts=None):
                        return comm.gather_plan(p, dst=0, parts=parts)

                # which physical GPU every rank flew on, gathered over the process group: N ranks must name N devices
                idents = [None] * world
                dist.all_gather_object(idents, eng.ctx.device_identity())
                # (distinct by uuid AND PCI address: a runtime that reports no uuid still tells its devices apart by bus id)
                if rank == 0:
                    c4["devices"] = idents
                    c4["distinct_devices"] = len(set(idents))
                if len(set(idents)) != world and not rehearsal:
                    raise RuntimeError(f"{world} ranks on {len(set(idents))} distinct GPUs: {idents}")

                # ---- the gather of the ROWS: once untimed (allocation, verification), then timed
                gathered, counts = rows_gather()
                if rank == 0 and (rehearsal or forced) and os.environ.get("UAVAC_BENCH_CORRUPT_GATHER") == "1":
                    gathered[gathered.shape[0] // 2, 4] += 1.0      # TEST HOOK (rehearsal / forced world 1 only): one wrong value must end in exit 3
                if rank == 0:
                    own = counts[0]
                    ok = sum(counts) == gathered.shape[0] and bool((gathered[:own] == plan4.traj[:own]).all())
                    # rank 0's own rows prove nothing about the transfers: every peer's block is checked against ITS rows below
                    # (first waypoint exactly) and, row for row, against the rows re-sampled from the gathered plan
                    # every peer's block starts with its first mission's first waypoint, at rest
                    offs = np.concatenate([[0], np.cumsum(counts)])
                    starts = [shard_bounds(C4_TOTAL, r, world, root_share, 0)[0] for r in range(world)]
                    first = missions(C4_TOTAL, C4_SEGMENTS, 0, C4_TOTAL)[starts, 0, :]
                    got = gathered[torch.as_tensor(offs[:-1], device=dev), 0:3].cpu().numpy()
                    ok = ok and bool(np.array_equal(got, first))
                    if not ok:
                        gather_err = "gathered rows do not match"
                holder = []
                gather_s = timed(lambda: holder.append(rows_gather()) or holder.clear(), 1)
                if rank == 0:
                    if rehearsal:
                        c4["REHEARSAL"] = "ranks share one GPU, gloo, a 200000-row slice per rank: control flow only, not a measurement"
                    c4.update({"gather_ms": gather_s * 1e3, "gather_rows_total": int(sum(counts)),
                               "gather_GBps_into_root": (sum(counts) - counts[0]) * 88 / gather_s / 1e9,
                               "steps_per_s_with_gather": C4_TOTAL * C4_TICKS / (c4_compute + gather_s),
                               "gather_verified": ok})

                # ---- the gather of the PLAN (coefficients, durations, rows per spline; the root re-samples): once untimed,
                # verified against the rows the row gather delivered, then timed -- in one shot from the full plans (round 3) and
                # PIPELINED from the rows-free plans (round 6: parts on a transfer stream, each sampled when it has arrived)
                def same_as_gathered(gp, pcounts):
                    total_rows = sum(pcounts)
                    if rehearsal:                          # only a slice of every peer's rows crossed gloo
                        offs_full = np.concatenate([[0], np.cumsum(pcounts)])
                        same = gp.traj.shape[0] == total_rows and bool((gp.traj[:pcounts[0]] == plan4.traj).all())
                        for r in range(world):
                            n = min(SLICE, pcounts[r])
                            same = same and bool((gp.traj[offs_full[r]:offs_full[r] + n] == gathered[offs[r]:offs[r] + n]).all())
                    else:
                        same = pcounts == counts and gp.traj.shape == gathered.shape and bool((gp.traj == gathered).all())
                    return same and bool((gp.coeffs[:plan4.B] == plan4.coeffs).all()) and \
                        bool((gp.row_offsets[:plan4.B + 1] == plan4.row_offsets).all()) and \
                        bool((gp.first_yaw[:plan4.B] == plan4.first_yaw).all())

                gp, pcounts = plan_gather()
                if rank == 0:
                    same = same_as_gathered(gp, pcounts)
                    if not same:
                        gather_err = gather_err or "rows re-sampled from the gathered plan differ from the gathered rows"
                    c4["plan_gather_verified"] = bool(same)
                    c4["plan_gather_bytes_into_root"] = int((C4_TOTAL - B4) * C4_SEGMENTS * 204)        # coefficients 192 + duration 8 + rows 4
                del gp
                gp, pcounts = plan_gather(plan4f, True)
                if rank == 0:
                    same = same_as_gathered(gp, pcounts)
                    if not same:
                        gather_err = gather_err or "rows sampled from the pipelined gather of the rows-free plans differ from the gathered rows"
                    c4["pipelined_plan_gather_verified"] = bool(same)
                del gp
                plan_s = timed(lambda: holder.append(plan_gather()) or holder.clear(), 2)
                pipe_s = timed(lambda: holder.append(plan_gather(plan4f, True)) or holder.clear(), 2)
                if rank == 0:
                    c4.update({"plan_gather_ms": plan_s * 1e3, "pipelined_plan_gather_ms": pipe_s * 1e3,
                               "steps_per_s_with_plan_gather": C4_TOTAL * C4_TICKS / (c4_compute + plan_s)})

                # The same job with the gather BESIDE the rollout instead of after it: the trajectories are final when
                # planning ends, so their transfer (rows), or the transfer of the plan and the root's sampling, starts
                # there on a second stream while the vehicles fly.  A problem here is reported (`overlap_error`) but does
                # not fail the run -- the verified serial gathers above are the ones that count.
                # (whether to go on is decided by all ranks together: only rank 0 knows what its verifications found)
                if everybody_fine(gather_err) and not rehearsal:
                    side = torch.cuda.Stream(device=dev)
                    known = comm.plan_counts(plan4f)                  # the all-gathers of the counts: once, outside the timed job
                    traj_all = torch.empty((sum(counts), 11), dtype=torch.float64, device=dev) if rank == 0 else None

                    def overlapped(p, fl, begin, root_flies_first=False):
                        def run():
                            eng.replan(p)
                            # The root's rollout must be RESIDENT before its sampler starts: the sampler's grid never leaves 256 free
                            # registers on a SIMD while it has workgroups left, and a rollout workgroup that arrives behind it waits
                            # for its last wave (profiles/r06_config4_root_overlap_order.jsonl: 8.1 ms against 5.1).  The peers
                            # enqueue their sends first: the root waits for them.
                            if root_flies_first and rank == 0:
                                started = torch.cuda.Event() if fl is not None else None
                                fly4(fl, started)
                                if started is not None:
                                    # (host side: begin the gather only when the flight's reset has run -- the rollout kernel is
                                    # then on the chip within microseconds, whatever the plan's arrival time turns out to be)
                                    while not started.query():
                                        pass
                                    time.sleep(50e-6)
                                ticket = begin(p, side)
                            else:
                                ticket = begin(p, side)
                                fly4(fl)
                            got, cnt = comm.gather_finish(ticket)
                            torch.cuda.synchronize()
                            return got, cnt
                        return run

                    variants = (("rows", overlapped(plan4, fleet4, lambda p, st: comm.gather_rows_begin(p.traj, dst=0, stream=st)),
                                 "overlapped_ms", "steps_per_s_gather_overlapped", "overlapped_verified",
                                 "plan with rows + 5000 ticks, the rows gathered to rank 0 beside the flight"),
                                ("plan", overlapped(plan4, fleet4, lambda p, st: comm.gather_plan_begin(p, dst=0, stream=st)),
                                 "plan_overlapped_ms", "steps_per_s_plan_gather_overlapped", "plan_overlapped_verified",
                                 "plan with rows on every rank + 5000 ticks, the plan gathered in one shot and re-sampled on rank 0 beside the "
                                 "flight (round 5: the peers' rows are sampled twice)"),
                                ("rows_free", overlapped(plan4f, fleet4f, lambda p, st: comm.gather_plan_begin(
                                    p, dst=0, stream=st, traj=traj_all, parts=True, known_counts=known), root_flies_first=True),
                                 "rows_free_overlapped_ms", "steps_per_s_rows_free_overlapped", "rows_free_overlapped_verified",
                                 "every rank plans ROWS-FREE (times, row counts, solve, first headings) and flies plan-fed; the plan travels in "
                                 "parts and rank 0 samples all rows once, part by part" +
                                 (", beside its own flight" if sizes4[0] > 0 else "; rank 0 flies nothing (shard_sizes[0] = 0: assembling the "
                                  "trajectories is its share of the job)") + " -- rank 0 alone sampled"))
                    for kind, run, k_ms, k_rate, k_ok, form in variants:
                        over_err, same = None, True
                        try:
                            got, cnt = run()
                            if rank == 0:
                                rows_ = got if kind == "rows" else got.traj
                                same = sum(cnt) == rows_.shape[0] and bool((rows_[:cnt[0]] == plan4.traj[:cnt[0]]).all())
                                if kind == "rows_free":               # ... and every peer's rows, against the literal row gather
                                    same = same and rows_.shape == gathered.shape and bool((rows_ == gathered).all())
                            del got
                        except Exception as exc:
                            over_err = f"{type(exc).__name__}: {exc}"
                        if everybody_fine(over_err):
                            try:
                                over_s = timed(lambda: holder.append(run()) or holder.clear(), 2)
                            except Exception as exc:
                                over_err, over_s = f"{type(exc).__name__}: {exc}", float("nan")
                            if rank == 0 and over_err is None:
                                c4.update({k_ms: over_s * 1e3, k_ok: bool(same), k_rate: C4_TOTAL * C4_TICKS / over_s})
                                # THE config-4 job: plan, fly, trajectories resident on rank 0 -- the fastest VERIFIED form of it
                                if kind != "rows" and same and ("end_to_end" not in c4 or over_s * 1e3 < c4["end_to_end"]["ms"]):
                                    c4["end_to_end"] = {"form": form, "ms": over_s * 1e3, "steps_per_s": C4_TOTAL * C4_TICKS / over_s}
                        if rank == 0 and over_err is not None:
                            c4["overlap_error"] = f"{kind}: {over_err}"
                        if not everybody_fine(over_err):
                            break
                del gathered
            except Exception as exc:                      # the timed result above must survive a collective problem
                gather_err = f"{type(exc).__name__}: {exc}"
            watchdog.cancel()

    def leave():
        if multi:
            if gather_err is None:
                if comm is not None:
                    comm.close()
                dist.destroy_process_group()
            else:                                     # a communicator that failed once may not shut down cleanly
                sys.stdout.flush()
                if rank == 0:
                    print(f"bench.py: config-4 gather FAILED ({gather_err}); the headline was measured before it; exit {GATHER_FAILURE_EXIT}",
                          file=sys.stderr, flush=True)
                os._exit(GATHER_FAILURE_EXIT)
        if gather_err is not None:
            sys.exit(GATHER_FAILURE_EXIT)

    if multi:                                     # every rank learns whether any rank failed
        bad = torch.tensor([0 if gather_err is None else 1], dtype=torch.int32, device=cdev)
        try:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()) and gather_err is None:
                gather_err = "gather failed on another rank"
        except Exception as exc:
            gather_err = gather_err or f"{type(exc).__name__}: {exc}"

    if rank != 0:
        leave()
        return
    if gather_err is not None:
        out["gather_error"] = gather_err
        out["gather_error_kind"] = "mismatch" if ("do not match" in gather_err or "differ" in gather_err) else "collective"
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(eng, wps)
        out["checks"]["yaw_2pi_forks"] = yaw_fork_census(eng)
    print(json.dumps(out), flush=True)
    leave()


if __name__ == "__main__":
    main()
