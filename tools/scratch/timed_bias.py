import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
eng = Engine("cuda:0")
plan = eng.plan(missions(65536, 12, 0, 65536), 3.0, 0.01)
def loop(warm, n):
    for _ in range(warm): eng.sample(plan)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): eng.sample(plan)
    b.record(); b.synchronize()
    return round(a.elapsed_time(b) / n, 4)
print("3 warm + 5:", [loop(3, 5) for _ in range(4)])
print("1 warm + 3:", [loop(1, 3) for _ in range(6)])
print("0 warm + 1:", [loop(0, 1) for _ in range(6)])
print("8 warm + 3:", [loop(8, 3) for _ in range(4)])
time.sleep(0.2)
print("after 0.2 s idle, 1 warm + 3:", [loop(1, 3) for _ in range(6)])
x = torch.empty_like(plan.traj); old = plan.traj; plan.traj = x
print("new buffer, 1 warm + 3:", [loop(1, 3) for _ in range(6)])
plan.traj = old; del x; torch.cuda.empty_cache()
print("after empty_cache, 1 warm + 3:", [loop(1, 3) for _ in range(6)])
eng.place_rows(plan, 1)
print("place_rows(1):", plan.placement_ms)
eng.place_rows(plan, 12)
print("place_rows(12):", plan.placement_ms, [loop(3, 5) for _ in range(2)])
