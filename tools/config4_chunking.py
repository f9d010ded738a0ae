"""BASELINE configs[3] as one rank of eight flies it (32 768 UAVs, m = 8, 5 000 ticks after a reset): five launches of 1 000 ticks into
a reused log against ONE launch of 5 000 ticks into a 17 GB log, and the per-launch times of the former.   python3 tools/config4_chunking.py [B]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
import torch
from bench import missions
from uav_ac.fleet import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
m, K = 8, 5000
eng = Engine("cuda:0")
plan = eng.plan(missions(B, m, 0, B), 3.0, 0.01)
fleet = eng.fleet(plan)
pitch = -(-B // 16) * 16
ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
out = {"B": B, "m": m, "ticks": K, "kernel": None}
for chunk in (1000, 2500, 5000):
    log = torch.empty((chunk, 13, pitch), dtype=torch.float64, device="cuda:0")
    times, per = [], []
    for rep in range(5):
        marks = [ev()]
        marks[0].record()
        fleet.reset()
        for _ in range(K // chunk):
            fleet.rollout(chunk, state_log=log, log_pitch=pitch)
            marks.append(ev()); marks[-1].record()
        torch.cuda.synchronize()
        if rep:
            times.append(marks[0].elapsed_time(marks[-1]))
            per.append([a.elapsed_time(b) for a, b in zip(marks, marks[1:])])
    out[f"chunk_{chunk}"] = {"ms_reset_plus_flight": round(float(np.median(times)), 4), "G_steps_per_s": round(B * K / float(np.median(times)) / 1e6, 2),
                             "ms_per_launch": [round(float(x), 4) for x in np.median(np.array(per), axis=0)]}
    del log
# the job as a rank runs it: the planning chain in front of every flight (other kernels, other workgroup shapes: does the flight
# behind them start from the same wave placement?) -- flight part alone, one launch
log = torch.empty((K, 13, pitch), dtype=torch.float64, device="cuda:0")
for align in (1, 2):
    fl = []
    for rep in range(6):
        eng.replan(plan)
        a, b = ev(), ev()
        a.record()
        fleet.reset()
        if align == 2:                       # a second aligner launch in front (the library issues one itself)
            fleet.rollout(1, state_log=log, log_pitch=pitch)
            fleet.reset()
        fleet.rollout(K, state_log=log, log_pitch=pitch)
        b.record(); torch.cuda.synchronize()
        if rep:
            fl.append(a.elapsed_time(b))
    out["after_replan_one_launch" + ("" if align == 1 else "_behind_a_1_tick_launch")] = {"ms_reset_plus_flight": round(float(np.median(fl)), 4),
                                                                                       "G_steps_per_s": round(B * K / float(np.median(fl)) / 1e6, 2)}
del log
out["kernel"] = eng.ctx.last_rollout_kernel()
print(json.dumps(out))
