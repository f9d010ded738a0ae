"""One sampler shape a few times (for rocprofv3 --pmc / --kernel-trace):  python3 tools/sampler_one.py WxG [reps] [B] [m]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
W, G = (int(v) for v in sys.argv[1].split("x"))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
m = int(sys.argv[4]) if len(sys.argv) > 4 else 12
eng = Engine("cuda:0")
eng.ctx.set_option("sampler_waves", W)
eng.ctx.set_option("sampler_group", G)
plan = eng.plan(missions(B, m, 0, B), 3.0, 0.01, placement_trials=1)
for _ in range(reps):
    eng.sample(plan)
torch.cuda.synchronize()
print("rows", plan.total_rows)
