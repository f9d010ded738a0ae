"""Rate of the literal drop-in path: one launch per tick (tc.step() + sim.step() for the whole fleet), eager
and replayed from a HIP graph captured through torch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
B = 65536
eng = Engine("cuda:0")
plan = eng.plan(missions(B, 12, 0, B), 3.0, 0.01)
a = eng.fleet(plan); b = eng.fleet(plan); c = eng.fleet(plan)
for _ in range(20): a.step()
torch.cuda.synchronize(); a.reset()
t0 = time.perf_counter()
for _ in range(1000): a.step()
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"eager  : {(t1-t0)*1e3:.1f} ms for 1000 single-tick launches -> {B*1000/(t1-t0)/1e9:.2f} G steps/s")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): b.step()
    torch.cuda.synchronize(); b.reset(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(100): b.step()
torch.cuda.synchronize(); b.reset(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"graph  : {(t1-t0)*1e3:.1f} ms for 10 replays x 100 ticks -> {B*1000/(t1-t0)/1e9:.2f} G steps/s")
c.rollout(1000); torch.cuda.synchronize()
print("graph == eager == fused:", bool((a.state == b.state).all()), bool((a.state == c.state).all()))
