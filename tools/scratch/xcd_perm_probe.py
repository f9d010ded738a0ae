"""DIAGNOSTIC (needs a -DUAVAC_DIAG_XCD_PERM build of both sampler files as UAVAC_LIB): the sampler on N row buffers with the
eighths of the batch handed to the XCDs in different orders.   python3 tools/scratch/xcd_perm_probe.py [n_buffers]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
from uav_ac import _native
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, m = 65536, 12
eng = Engine("cuda:0")
lib = ctypes.CDLL(_native.LIB_PATH)
plan = eng.plan(missions(B, m, 0, B), 3.0, 0.01, placement_trials=1)
bufs = [plan.traj] + [torch.empty_like(plan.traj) for _ in range(NB - 1)]
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
perms = {"identity": [0, 1, 2, 3, 4, 5, 6, 7], "rot1": [1, 2, 3, 4, 5, 6, 7, 0], "rot2": [2, 3, 4, 5, 6, 7, 0, 1], "rot4": [4, 5, 6, 7, 0, 1, 2, 3],
         "reverse": [7, 6, 5, 4, 3, 2, 1, 0], "swap_pairs": [1, 0, 3, 2, 5, 4, 7, 6], "bitrev": [0, 4, 2, 6, 1, 5, 3, 7]}
for W in (1, 4):
    eng.ctx.set_option("sampler_waves", W)
    for name, p in perms.items():
        arr = (ctypes.c_int * 8)(*p)
        torch.cuda.synchronize()
        assert lib.uavac_diag_xcd_perm_one(arr) == 0 and lib.uavac_diag_xcd_perm_stream(arr) == 0
        out = []
        for t in bufs:
            plan.traj = t
            eng.sample(plan); eng.sample(plan)
            a.record()
            for _ in range(5):
                eng.sample(plan)
            b.record(); torch.cuda.synchronize()
            out.append(round(a.elapsed_time(b) / 5, 4))
        print(json.dumps({"waves": W, "perm": name, "ms_per_buffer": out}), flush=True)
