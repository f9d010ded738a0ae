"""Is the half-full chip's longer tick (0.87-0.93 us at 32 768 UAVs against 0.77 at <= 16 384) the HBM write path?  The logged
rollout timed through the library given by UAVAC_LIB: the product build, or a DIAGNOSTIC build in which every tick overwrites log
slot 0 (`-DUAVAC_DIAG_LOG_WRAP=1`, tools/ab/libuavac_logwrap.so: same instructions and stores, but the 3.4-6.8 MB stay in L2).
    python3 tools/log_wrap_probe.py;  UAVAC_LIB=tools/ab/libuavac_logwrap.so python3 tools/log_wrap_probe.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
K = 2000
eng = Engine("cuda:0")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in (8192, 16384, 24576, 32768, 49152, 65536):
    plan = eng.plan(missions(B, 8, 0, B), 3.0, 0.01)
    log = torch.empty((K, 13, B), dtype=torch.float64, device="cuda:0")
    fleet = eng.fleet(plan, from_plan=True)
    for _ in range(3):
        fleet.rollout(K, state_log=log)
    best = 1e9
    for _ in range(4):
        a.record()
        fleet.rollout(K, state_log=log)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    print(json.dumps({"lib": os.path.basename(os.environ.get("UAVAC_LIB", "libuavac.so")), "B": B, "us_per_tick": round(best / K * 1e3, 4),
                      "G_steps_per_s": round(B * K / best / 1e6, 2)}), flush=True)
    del log, plan, fleet
