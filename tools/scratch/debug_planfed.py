import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np, torch
from oracle import minsnap_oracle as mo
from uav_ac.fleet import Engine
eng = Engine("cuda:0")
for B, m, K in ((130, 3, 2600), (64, 1, 900), (5000, 12, 1200)):
    wps = mo.synthetic_missions(B, m)
    plan = eng.plan(wps, 3.0, 0.01)
    a, b = eng.fleet(plan, from_plan=True), eng.fleet(plan, from_plan=False)
    la, _ = a.rollout(K, state_log=True)
    lb, _ = b.rollout(K, state_log=True)
    diff = (la != lb).any(dim=1)          # (K, B)
    bad_lanes = diff.any(dim=0).nonzero().flatten().tolist()
    print("B", B, "m", m, "mismatching lanes:", len(bad_lanes), bad_lanes[:10])
    sr = plan.seg_rows.cpu().numpy()
    for lane in bad_lanes[:4]:
        k0 = int(diff[:, lane].nonzero()[0])
        print("  lane", lane, "first bad tick", k0, "outer tick", k0 // 10, "seg_rows", sr[lane].tolist(), "cum", np.cumsum(sr[lane]).tolist())
