#!/usr/bin/env python3
"""Benchmark of the hot path on BASELINE.json's config 3: batch 65 536 UAVs per GPU, 12-segment
missions (start/end time factor 1.5), minimum-snap solve + sampling followed by 10 000 fused
controller + dynamics ticks, on synthetic missions (SURVEY.md 8(d) generator).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one whole job over one batch: times/row counts + coefficient solve + sampler (once per
mission), vehicle reset, then 10 000 control ticks as 10 launches of 1 000 ticks whose 6.8 GB state
log buffer is reused.  Inputs (waypoints) are resident in HBM before the timed region.  The batch
shards by mission index over ranks with no data-path collective (weak scaling: 65 536 UAVs per
GPU); the final gather of trajectories to rank 0 over RCCL is timed separately (`gather_ms`).

Prints ONE JSON line on rank 0.  `value` = UAV control ticks of all ranks / wall time of the K timed
steps (planning time included in the denominator).  `roofline` prices the dominant kernel
(control_rollout) against the HBM peak with the algorithmic 112.8 B per UAV tick of SURVEY.md 8(d);
`cpu_baseline` is the CPU oracle timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
B_PER_GPU = 65536
SEGMENTS = 12
TICKS = 10000
CHUNK = 1000
VELOCITY, DT, F = 3.0, 0.01, 10
# fp64 VALU instructions one lane executes per tick in control_rollout_kernel<1,true,false,false>, counted in the
# gfx950 ISA of the small-angle path every tick takes (207 in the per-tick body + 377 in the outer block / F);
# priced against the vector fp64 peak of MI355X_MICROARCH.md (78.6 TFLOP/s = 39.3 T lane-FMA/s).
FP64_VALU_PER_TICK = 245
FP64_LANE_INSTR_PEAK = 39.3e12
GATHER_TIMEOUT_S = 240


def missions(B_total, m, lo, hi):
    """SURVEY.md 8(d) generator restated for the bench (same draw order as the oracle's copy)."""
    rng = np.random.default_rng(20260807 + m)
    d = rng.standard_normal((B_total, m, 3)) * np.array([1, 1, 0.25])
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    L = rng.uniform(2.5, 3.5, (B_total, m, 1))
    w0 = np.concatenate([rng.uniform(0, 24, (B_total, 1, 1)), rng.uniform(0, 14, (B_total, 1, 1)),
                         np.full((B_total, 1, 1), -3.0)], axis=2)
    return np.concatenate([w0, w0 + np.cumsum(L * d, axis=1)], axis=1)[lo:hi]


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="UAVs per GPU (default: config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from uav_ac.fleet import Engine, Fleet, gather_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # UAVAC_BENCH_BACKEND=gloo is a rehearsal switch: it lets several ranks share one GPU (RCCL refuses that), so
    # the N > 1 control flow can be exercised on a 1-GPU box.  The driver's runs use nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("UAVAC_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    B, m = args.batch, SEGMENTS
    eng = Engine(dev)
    wps = missions(B * world, m, rank * B, (rank + 1) * B)
    plan = eng.plan(wps, VELOCITY, DT)                   # allocates; also the first warm-up
    eng.check(plan)
    fleet = eng.fleet(plan)
    log = torch.empty((CHUNK, 13, B), dtype=torch.float64, device=dev)
    n_chunks = TICKS // CHUNK
    ev = lambda: torch.cuda.Event(enable_timing=True)    # noqa: E731  (torch's current stream = the ctx's stream)

    def one_step(record=None):
        e0, e1 = ev(), ev()
        e0.record()
        eng.solve(plan)
        eng.sample(plan)
        e1.record()
        fleet.reset()
        pairs = []
        for _ in range(n_chunks):
            a, b = ev(), ev()
            a.record()
            fleet.rollout(CHUNK, state_log=log)
            b.record()
            pairs.append((a, b))
        if record is not None:
            record.append(((e0, e1), pairs))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
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
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    plan_ms = [a.elapsed_time(b) for (a, b), _ in rec]
    roll_ms = [a.elapsed_time(b) for _, pairs in rec for a, b in pairs]
    roll_avg_s = float(np.mean(roll_ms)) * 1e-3
    plan_avg_s = float(np.mean(plan_ms)) * 1e-3

    # sanity inside the bench: after TICKS ticks every cursor must sit at min(TICKS/F, rows-1) exactly, and
    # the share of UAVs within the reference's 0.5 m acceptance of their current target row is reported.
    # (The reference controller itself loses about 1 in 7 of the 8(d) missions -- the CPU oracle loses the
    # same lanes, tests/test_gpu_control.py -- the work per tick is identical either way.)
    nrows = plan.row_offsets[1:] - plan.row_offsets[:-1]
    idx = fleet.trajectory_index.long()
    cursor_ok = bool((idx == torch.clamp(torch.full_like(nrows, TICKS // F), max=nrows - 1)).all())
    target = plan.traj[plan.row_offsets[:-1] + idx, 0:3].T
    track = (fleet.X[0:3] - target).norm(dim=0)
    kept = track < 0.5
    frac_kept = float(kept.double().mean())
    finite_kept = bool(torch.isfinite(fleet.state[:, kept]).all())

    out = None
    if rank == 0:
        total_ticks = float(world) * B * TICKS * args.steps
        value = total_ticks / elapsed
        roll_bytes = Fleet.algorithmic_bytes(B, CHUNK, F)
        achieved = roll_bytes / roll_avg_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and B == B_PER_GPU:
            with open(tpath) as fh:
                traffic = json.load(fh).get("control_rollout_bytes_per_launch")
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
            "config": {"workload": "BASELINE.json configs[2]: batch 65536 UAVs/GPU, 12-segment missions with start/end "
                                   "time factor 1.5, min-snap solve+sample then 10000 fused controller+dynamics ticks "
                                   "(10 launches x 1000 ticks, 13-f64 state logged every tick)",
                       "batch_per_gpu": B, "segments": m, "ticks": TICKS, "ticks_per_launch": CHUNK,
                       "velocity": VELOCITY, "dt": DT, "inner_per_outer": F, "rows": plan.total_rows,
                       "parallelism": f"missions sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": "control_rollout_kernel<1, true, false, false, true>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": roll_bytes, "avg_launch_ms": roll_avg_s * 1e3,
                         "fp64_valu": {"lane_instr_per_tick": FP64_VALU_PER_TICK,
                                       "achieved": B * CHUNK * FP64_VALU_PER_TICK / roll_avg_s,
                                       "peak": FP64_LANE_INSTR_PEAK, "unit": "fp64 lane-instr/s",
                                       "frac": B * CHUNK * FP64_VALU_PER_TICK / roll_avg_s / FP64_LANE_INSTR_PEAK}},
            "minsnap": {"metric": "min-snap segments solved/sec", "value": B * m / plan_avg_s, "unit": "segments/s",
                        "ms_solve_plus_sample": plan_avg_s * 1e3,
                        "roofline": {"bound": "hbm", "achieved": plan.algorithmic_bytes / plan_avg_s / 1e9,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": plan.algorithmic_bytes / plan_avg_s / 1e9 / HBM_PEAK_GBS,
                                     "algorithmic_bytes": plan.algorithmic_bytes}},
            "rollout_only": {"value": B * CHUNK / roll_avg_s, "unit": "UAV control-steps/s per GPU",
                             "note": "SURVEY 8(d)(i): B x K / time of the rollout launches alone (`value` above also "
                                     "carries the planning time of every step)"},
            "checks": {"frac_uavs_within_0.5m_of_target_row": frac_kept, "tracking_lanes_finite": finite_kept,
                       "all_trajectory_cursors_exact": cursor_ok},
        }

    # final gather of the sampled trajectories to rank 0 (north_star: the only collective).  The timed result is
    # complete before it starts; a watchdog makes sure the result line still gets printed if the exchange stalls.
    gather_ms, gather_err = None, None
    if world > 1:
        import threading

        def bail():
            if rank == 0:
                out["gather_error"] = f"no completion within {GATHER_TIMEOUT_S} s"
                print(json.dumps(out), flush=True)
            os._exit(0)
        watchdog = threading.Timer(GATHER_TIMEOUT_S, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            barrier()
            g0 = time.perf_counter()
            gathered, counts = gather_rows(plan.traj, dst=0)
            barrier()
            gather_ms = (time.perf_counter() - g0) * 1e3
            if rank == 0 and (sum(counts) != gathered.shape[0] or not bool((gathered[:plan.total_rows] == plan.traj).all())):
                gather_err = "gathered rows do not match"
            del gathered
        except Exception as exc:                      # the timed result above must survive a collective problem
            gather_err = f"{type(exc).__name__}: {exc}"
        watchdog.cancel()

    def leave():
        if world > 1:
            if gather_err is None:
                dist.destroy_process_group()
            else:                                     # a communicator that failed once may not shut down cleanly
                sys.stdout.flush()
                os._exit(0)

    if rank != 0:
        leave()
        return
    if gather_ms is not None:
        out["gather_ms"] = gather_ms
        out["gather_GBps_into_root"] = (world - 1) * plan.total_rows * 88 / (gather_ms * 1e-3) / 1e9
    if gather_err is not None:
        out["gather_error"] = gather_err
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(eng, wps)
    print(json.dumps(out), flush=True)
    leave()


if __name__ == "__main__":
    main()
