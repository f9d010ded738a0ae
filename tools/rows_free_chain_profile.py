"""The rows-free planning chain of a config-4 peer (37 450 missions of 8 segments) under `rocprofv3 --kernel-trace --stats`: K0 (two
launches), K1, the first-heading kernel -- and, for comparison, the chain with rows (commit kernel + sampler on top).

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o chain -- python3 tools/rows_free_chain_profile.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions, VELOCITY, DT
from uav_ac.fleet import Engine

eng = Engine("cuda:0")
B, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (37450, 8)
wps = missions(262144, m, 0, B)
free = eng.plan(wps, VELOCITY, DT, rows=False)
full = eng.plan(wps, VELOCITY, DT)
for _ in range(30):
    eng.replan(free)
torch.cuda.synchronize()
for _ in range(10):
    eng.replan(full)
torch.cuda.synchronize()
print("rows-free == full:", all(bool(torch.equal(getattr(free, k), getattr(full, k))) for k in ("times", "seg_rows", "row_offsets", "coeffs", "first_yaw")))
