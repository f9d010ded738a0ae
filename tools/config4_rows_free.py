"""BASELINE configs[3] with the rows sampled ONCE (round 6): what one GPU can measure of the 8-rank job in which every rank plans
rows-free (times, row counts, solve, first headings: Engine.plan(rows=False)), flies plan-fed and ships its plan, and rank 0
samples all 262 144 missions' rows from the gathered plan beside its own flight.

  peer:  a rank's chain at n UAVs (m = 8, 5 000 ticks in one launch) -- planning chain with rows / rows-free, then the flight
  root:  the re-sampling of all 262 144 missions (20.9 GB of rows) on a side stream BESIDE a flight of n_root UAVs, and each alone

    python3 tools/config4_rows_free.py            -> JSON lines (profiles/r06_config4_rows_free.jsonl)
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
import torch
from bench import missions, C4_TOTAL, C4_SEGMENTS, C4_TICKS, VELOCITY, DT
from uav_ac.fleet import Engine

eng = Engine("cuda:0")
dev = eng.device
ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
m, K = C4_SEGMENTS, C4_TICKS


def med(xs):
    return round(float(np.median(xs)), 4)


def peer(n, reps=7):
    wps = missions(C4_TOTAL, m, 0, n)
    pitch = -(-n // 16) * 16
    log = torch.empty((K, 13, pitch), dtype=torch.float64, device=dev)
    out = {"leg": "peer", "uavs": n, "ticks": K}
    for label, rows in (("with_rows", True), ("rows_free", False)):
        plan = eng.plan(wps, VELOCITY, DT, rows=rows)
        fleet = eng.fleet(plan, from_plan=True)
        chain, flight, both = [], [], []
        for rep in range(reps):
            a, b, c = ev(), ev(), ev()
            a.record()
            eng.replan(plan)
            b.record()
            fleet.reset()
            fleet.rollout(K, state_log=log, log_pitch=pitch)
            c.record()
            torch.cuda.synchronize()
            if rep >= 2:
                chain.append(a.elapsed_time(b)); flight.append(b.elapsed_time(c)); both.append(a.elapsed_time(c))
        out[label] = {"plan_ms": med(chain), "flight_ms": med(flight), "plan_plus_flight_ms": med(both),
                      "kernel": eng.ctx.last_rollout_kernel()}
        del plan, fleet
    out["saved_ms"] = round(out["with_rows"]["plan_plus_flight_ms"] - out["rows_free"]["plan_plus_flight_ms"], 4)
    del log
    torch.cuda.empty_cache()
    return out


def root(n_root, parts, total_rows, traj, reps=6, world=8):
    """rank 0 of the 8-rank job on ONE GPU: its own rows-free plan + flight of n_root UAVs on the current stream (enqueued FIRST:
    a rollout workgroup that arrives behind the sampler's grid waits for its last wave), the sampling of all 262 144 missions
    from the (already gathered) plan on a side stream beside it -- in one launch, and in the launches of the pipelined gather:
    part p (1/16, 3/16, 1/4, 1/2 of every block) of each of `world` ranks' blocks, 4 x world launches."""
    from uav_ac.sharding import PIPELINE_SHARES, part_bounds, shard_sizes
    co, tm, sr = parts
    wps = missions(C4_TOTAL, m, 0, n_root)
    plan = eng.plan(wps, VELOCITY, DT, rows=False)
    fleet = eng.fleet(plan)
    pitch = -(-n_root // 16) * 16
    log = torch.empty((K, 13, pitch), dtype=torch.float64, device=dev)
    side = torch.cuda.Stream(device=dev)
    helpers = [torch.cuda.Stream(device=dev) for _ in range(2)]
    here = torch.cuda.current_stream(dev)
    res = {"leg": "root", "uavs": n_root, "resampled_missions": C4_TOTAL, "rows": total_rows, "row_GB": round(total_rows * 88 / 1e9, 2)}
    sizes = shard_sizes(C4_TOTAL, world, n_root / C4_TOTAL, 0)
    base = np.concatenate([[0], np.cumsum(sizes)])
    bounds = [part_bounds(b, PIPELINE_SHARES) for b in sizes]

    def fly():
        eng.replan(plan)
        fleet.reset()
        fleet.rollout(K, state_log=log, log_pitch=pitch)

    def resample(stream, pipelined):
        with torch.cuda.stream(stream):
            got = eng.plan_from_parts(co, tm, sr, m, VELOCITY, DT, total_rows=total_rows, traj=traj, sample=not pipelined)
            if pipelined == 1:
                for p in range(len(PIPELINE_SHARES)):
                    for r in range(world):
                        eng.sample_range(got, base[r] + bounds[r][p], base[r] + bounds[r][p + 1])
        if pipelined == 3:                                    # the ranges of a part over three streams (what RcclComm._sample_parts does)
            lanes = [stream] + helpers
            ready = torch.cuda.Event(); ready.record(stream)
            for h in helpers:
                h.wait_event(ready)
            k = 0
            for p in range(len(PIPELINE_SHARES)):
                for r in range(world):
                    if bounds[r][p + 1] > bounds[r][p]:
                        with torch.cuda.stream(lanes[k % 3]):
                            eng.sample_range(got, base[r] + bounds[r][p], base[r] + bounds[r][p + 1])
                        k += 1
            for h in helpers:
                stream.wait_stream(h)
        eng._bind_stream()

    for label in ("flight_alone", "sample_alone_one_launch", "sample_alone_32_launches", "sample_alone_32_launches_3_streams", "both_sampler_first",
                  "both_flight_first_one_launch", "both_flight_first_32_launches", "both_flight_first_32_launches_3_streams"):
        ts = []
        for rep in range(reps):
            torch.cuda.synchronize()
            a, b = ev(), ev()
            a.record()
            pipelined = 3 if label.endswith("3_streams") else (1 if label.endswith("32_launches") else 0)
            if label == "both_sampler_first":
                side.wait_stream(here)
                resample(side, False)
                fly()
                here.wait_stream(side)
            elif label.startswith("both_flight_first"):
                fly()
                # (the plan arrives ~0.2 ms into the job: one wave sleeps that long in front of the sampler)
                eng.clock_probe_begin(int(os.environ.get("ROOT_DELAY_US", "200")), stream=side)
                resample(side, pipelined)
                here.wait_stream(side)
            elif label == "flight_alone":
                fly()
            else:
                resample(here, pipelined)
            b.record()
            torch.cuda.synchronize()
            if rep >= 2:
                ts.append(a.elapsed_time(b))
        res[label + "_ms"] = med(ts)
    res["kernel"] = eng.ctx.last_rollout_kernel()
    del plan, fleet, log
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    print(json.dumps({"build": __import__("bench").nat_build_info(), "gpu": eng.ctx.device_identity()}), flush=True)
    if "--root-only" not in sys.argv:
        for n in (32768, 35237, 36000, 36352, 36864, 37450):
            print(json.dumps(peer(n)), flush=True)
    # the gathered plan of the whole job (what arrives at rank 0: coefficients, durations, rows per spline)
    allp = eng.plan(missions(C4_TOTAL, m, 0, C4_TOTAL), VELOCITY, DT, rows=False)
    parts = (allp.coeffs.reshape(-1, 8, 3), allp.times.reshape(-1), allp.seg_rows.reshape(-1))
    total_rows = allp.total_rows
    traj = torch.empty((total_rows, 11), dtype=torch.float64, device=dev)
    for n_root in [int(x) for x in os.environ.get("ROOT_SIZES", "2048,4096,8192,12288").split(",")]:
        print(json.dumps(root(n_root, parts, total_rows, traj)), flush=True)
