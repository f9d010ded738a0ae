"""The fused rollout (B = 65 536, state log, 1 000 ticks per launch) in two regimes: while the missions are being
flown (every outer tick pulls a fresh 88-B row per UAV from HBM) and after their end (the last row is held: the same
instructions, but no new trajectory lines come from DRAM)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
B = 65536
eng = Engine("cuda:0")
plan = eng.plan(missions(B, 12, 0, B), 3.0, 0.01)
fleet = eng.fleet(plan)
log = torch.empty((1000, 13, B), dtype=torch.float64, device="cuda:0")
def timed(n):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fleet.rollout(1000, state_log=log)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
fleet.reset(); fleet.rollout(1000, state_log=log); torch.cuda.synchronize()
print(f"map={os.environ.get('UAVAC_XCD_MAP','1')} flying  (ticks 1000-9000): {timed(8):.3f} ms per 1000 ticks")
fleet.rollout(12000); torch.cuda.synchronize()
print(f"map={os.environ.get('UAVAC_XCD_MAP','1')} holding (ticks > 21000)  : {timed(8):.3f} ms per 1000 ticks")
