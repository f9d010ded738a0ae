"""The rollout's time per 1 000 logged ticks into N log buffers of 6.8 GB allocated one after the other and all kept."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
eng = Engine("cuda:0")
B = 65536
plan = eng.plan(missions(B, 12, 0, B), 3.0, 0.01)
fleet = eng.fleet(plan)
def timed(lg):
    fleet.reset(); fleet.rollout(2000)
    for _ in range(2): fleet.rollout(1000, state_log=lg)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(6): fleet.rollout(1000, state_log=lg)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 6
logs, out = [], []
for i in range(N):
    logs.append(torch.empty((1000, 13, B), dtype=torch.float64, device="cuda:0"))
    out.append(timed(logs[-1]))
print("rollout ms per 1000 ticks per log buffer, in allocation order:", " ".join(f"{t:.3f}" for t in out))
out2 = [timed(l) for l in logs]
print("again:                                                          ", " ".join(f"{t:.3f}" for t in out2))
