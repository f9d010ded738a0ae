"""Time of the min-snap solve alone (block-Thomas kernel) at the BASELINE sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
eng = Engine("cuda:0")
for B, m in ((65536, 12), (65536, 20), (65536, 8), (32768, 8), (16384, 8), (8192, 8), (4096, 8), (65536, 4), (32768, 4), (65536, 1), (65536, 2), (1000, 64)):
    plan = eng.plan(missions(B, m, 0, B), 3.0, 0.01)
    out = []
    for park in (0, 1, -1):                      # forward sweep parked in the HBM workspace / in LDS (when it fits) / the launcher's choice
        eng.ctx.set_option("solve_park", park)
        for _ in range(3): eng.solve(plan)
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): eng.solve(plan)
        b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / 20)       # (row counts + solve: two small launches and the solve)
    print(f"B={B} m={m}: times + solve, parked in HBM {out[0]*1e3:.1f} us, in LDS {out[1]*1e3:.1f} us, launcher's choice {out[2]*1e3:.1f} us "
          f"= {B*m/out[2]/1e3:.0f} M segments/s")
    del plan
