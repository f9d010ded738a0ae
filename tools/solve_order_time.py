"""Row counts + coefficient solve (`Engine.solve`) with the one-ended kernel (solve_order 0) and the two-ended one (1), interleaved
rounds, per launch shape the launcher may pick: us per call.   python3 tools/solve_order_time.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
import torch
from bench import missions
from uav_ac.fleet import Engine
eng = Engine("cuda:0")
SHAPES = {"auto": (-1, -1, -1), "lanes64": (64, 0, 0), "lanes32": (32, 0, 0), "lanes16": (16, 0, 0), "keep": (64, 1, 0), "lds64": (64, 0, 1), "lds32": (32, 0, 1)}
for B, m in ((65536, 12), (65536, 20), (65536, 8), (32768, 8), (16384, 8), (4096, 8), (262144, 8), (65536, 4), (1000, 64)):
    plan = eng.plan(missions(B, m, 0, B), 3.0, 0.01)
    res = {}
    for rnd in range(3):
        for order in (0, 1):
            for name, (lanes, keep, park) in SHAPES.items():
                eng.ctx.set_option("solve_order", order); eng.ctx.set_option("solve_lanes", lanes)
                eng.ctx.set_option("solve_keep", keep); eng.ctx.set_option("solve_park", park)
                for _ in range(2): eng.solve(plan)
                a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10): eng.solve(plan)
                b.record(); torch.cuda.synchronize()
                res.setdefault((order, name), []).append(a.elapsed_time(b) * 100)
    for k in ("solve_order", "solve_lanes", "solve_keep", "solve_park"):
        eng.ctx.set_option(k, -1 if k != "solve_order" else 0)
    print(json.dumps({"B": B, "m": m, **{f"order{o}_{n}_us": round(float(np.median(v)), 1) for (o, n), v in res.items()}}), flush=True)
    del plan
