"""The logged, plan-fed rollout with and without the LDS-sized cap on workgroups per CU (option "cu_balance"), interleaved rounds,
per batch size.   python3 tools/cu_balance_ab.py [m] [sizes...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
import torch
from bench import missions
from uav_ac.fleet import Engine
m = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sizes = [int(a) for a in sys.argv[2:]] or [4096, 12288, 16384, 20480, 24576, 28672, 32768, 35237, 40960, 49152, 57344, 65536]
eng = Engine("cuda:0")
K = 1000
for B in sizes:
    plan = eng.plan(missions(B, m, 0, B), 3.0, 0.01)
    pitch = -(-B // 16) * 16
    log = torch.empty((K, 13, pitch), dtype=torch.float64, device="cuda:0")
    res = {0: [], 1: []}
    for feed in (True, False) if B <= 20480 else (True,):
        fleet = eng.fleet(plan, from_plan=feed)
        res = {0: [], 1: []}
        for rnd in range(5):
            for bal in (0, 1) if rnd % 2 == 0 else (1, 0):
                eng.ctx.set_option("cu_balance", bal)
                fleet.reset()
                fleet.rollout(K, state_log=log, log_pitch=pitch)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(4):
                    fleet.rollout(K, state_log=log, log_pitch=pitch)
                b.record(); torch.cuda.synchronize()
                if rnd:
                    res[bal].append(a.elapsed_time(b) / 4)
        eng.ctx.set_option("cu_balance", 1)
        off, on = float(np.median(res[0])), float(np.median(res[1]))
        print(json.dumps({"B": B, "m": m, "feed": "plan" if feed else "rows", "tiles": -(-B // 64), "ms_off": round(off, 4), "ms_on": round(on, 4),
                          "gain_pct": round(100 * (off - on) / off, 2), "G_steps_per_s_on": round(B * K / on / 1e6, 2)}), flush=True)
    del log, plan, fleet
