"""The sampler's speed mode of this process next to the device addresses of its buffers (is the mode an alignment effect?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
eng = Engine("cuda:0")
plan = eng.plan(missions(65536, 12, 0, 65536), 3.0, 0.01)
def timed(n=30):
    for _ in range(8): eng.sample(plan)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): eng.sample(plan)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
t0 = timed()
p0 = plan.traj.data_ptr()
keep = plan.traj
plan.traj = torch.empty_like(keep)            # a second row buffer, the first one still allocated
t1 = timed()
p1 = plan.traj.data_ptr()
print(f"traj {p0:#x} (mod 1 GiB {p0 % (1<<30):#x}): {t0:.3f} ms | second buffer {p1:#x} (mod 1 GiB {p1 % (1<<30):#x}): {t1:.3f} ms | "
      f"coeffs {plan.coeffs.data_ptr():#x} seg_rows {plan.seg_rows.data_ptr():#x}")
