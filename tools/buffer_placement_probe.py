"""How much do the two big write streams depend on WHERE their output buffer lies?  One process, several row buffers /
log buffers allocated side by side (all kept allocated, so each gets different physical pages), the same kernels timed on
each in turn, twice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
eng = Engine("cuda:0")
B = 65536
plan = eng.plan(missions(B, 12, 0, B), 3.0, 0.01)
fleet = eng.fleet(plan)
def ev(): return torch.cuda.Event(enable_timing=True)
def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = ev(), ev()
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
trajs = [plan.traj] + [torch.empty_like(plan.traj) for _ in range(N - 1)]
logs = [torch.empty((1000, 13, B), dtype=torch.float64, device="cuda:0") for _ in range(N)]
for rnd in range(2):
    ts = []
    for t in trajs:
        plan.traj = t
        ts.append(timed(lambda: eng.sample(plan), 20))
    print(f"round {rnd}: sampler per row buffer (ms): " + " ".join(f"{x:.3f}" for x in ts))
    ls = []
    for lg in logs:
        fleet.reset(); fleet.rollout(2000)                         # past take-off
        ls.append(timed(lambda: fleet.rollout(1000, state_log=lg), 8))
    print(f"round {rnd}: rollout per log buffer (ms):  " + " ".join(f"{x:.3f}" for x in ls))
