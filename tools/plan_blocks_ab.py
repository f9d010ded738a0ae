"""A/B of the blocked planning chain (option "plan_blocks", round 6): uavac_minsnap_plan_dev with the batch cut into 1 (off), 2,
4, 8 mission blocks -- the sampling of block i on an auxiliary stream beside the solve of block i + 1.  Same process, same row
buffer, the variants interleaved round after round; every variant's plan is compared bit for bit with the un-blocked one.

    python3 tools/plan_blocks_ab.py [B m] ...        -> JSON lines (profiles/r06_plan_blocks_ab.jsonl)
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
import torch
from bench import missions, VELOCITY, DT
from uav_ac.fleet import Engine

eng = Engine("cuda:0")
ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
shapes = [(65536, 12), (65536, 8), (32768, 8), (65536, 20), (262144, 8)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
for B, m in shapes:
    plan = eng.plan(missions(B, m, 0, B), VELOCITY, DT)
    ref = {k: getattr(plan, k).clone() for k in ("times", "seg_rows", "row_offsets", "coeffs", "first_yaw")}
    ref_rows = plan.traj.clone()
    variants = (1, 2, 4, 8)
    times = {v: [] for v in variants}
    same = {}
    for v in variants:                                           # correctness first: poison, re-plan, compare
        eng.ctx.set_option("plan_blocks", v)
        plan.traj.fill_(float("nan")); plan.coeffs.fill_(float("nan")); plan.first_yaw.fill_(float("nan"))
        eng.replan(plan)
        torch.cuda.synchronize()
        same[v] = bool(torch.equal(plan.traj, ref_rows)) and all(bool(torch.equal(getattr(plan, k), t)) for k, t in ref.items())
    for rnd in range(9):
        for v in variants:
            eng.ctx.set_option("plan_blocks", v)
            eng.replan(plan)                                     # (the variant's own warm-up: streams, clocks)
            marks = []
            for _ in range(6):
                a, b = ev(), ev()
                a.record()
                eng.replan(plan)
                b.record()
                marks.append((a, b))
            torch.cuda.synchronize()
            if rnd:
                times[v].extend(a.elapsed_time(b) for a, b in marks[1:])
    eng.ctx.set_option("plan_blocks", 1)
    base = float(np.median(times[1]))
    print(json.dumps({"B": B, "m": m, "rows": plan.total_rows, "flags": eng.take_flags(),
                      **{f"blocks_{v}": {"ms_median": round(float(np.median(times[v])), 4), "ms_min": round(float(np.min(times[v])), 4),
                                         "vs_off": round(float(np.median(times[v])) / base, 4), "bit_identical": same[v]} for v in variants}}),
          flush=True)
    del plan, ref, ref_rows
    torch.cuda.empty_cache()
