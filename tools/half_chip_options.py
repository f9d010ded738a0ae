"""The logged, plan-fed rollout between one and three workgroups per CU under every launch option that applies there: hand-over point
(late_handover 0 / 1), placeholder wave (idle_waves 0 / 1), who evaluates the target rows (coeff_dma 1 / 2), LDS padding (caps the
workgroups per CU).  Interleaved rounds; one JSON line per case with the median.   python3 tools/half_chip_options.py [m] [sizes...]"""
import itertools, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
import torch
from bench import missions
from uav_ac.fleet import Engine
m = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sizes = [int(a) for a in sys.argv[2:]] or [24576, 32768, 35237]
eng = Engine("cuda:0")
K = 1000
for B in sizes:
    plan = eng.plan(missions(B, m, 0, B), 3.0, 0.01)
    pitch = -(-B // 16) * 16
    log = torch.empty((K, 13, pitch), dtype=torch.float64, device="cuda:0")
    fleet = eng.fleet(plan, from_plan=True)
    if os.environ.get("SWEEP") == "launcher":        # what the launcher should pick: hand-over point x mode, placeholder wave as it picks it
        cases = [(-1, -1, -1)] + [(late, -1, mode) for late in (0, 1) for mode in (0, 1, 2) if not (mode == 2 and B > 32768)]
    else:
        cases = [(-1, -1, -1)] + list(itertools.product((0, 1), (0, 1), (1, 2)))
    res = {c: [] for c in cases}
    for rnd in range(4):
        for late, idle, mode in cases:
            eng.ctx.set_option("late_handover", late); eng.ctx.set_option("idle_waves", idle); eng.ctx.set_option("coeff_dma", mode)
            fleet.reset()
            fleet.rollout(K, state_log=log, log_pitch=pitch)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                fleet.rollout(K, state_log=log, log_pitch=pitch)
            b.record()
            torch.cuda.synchronize()
            if rnd:
                res[(late, idle, mode)].append(a.elapsed_time(b) / 4)
    for (late, idle, mode), v in res.items():
        print(json.dumps({"B": B, "m": m, "late_handover": late, "idle_waves": idle, "coeff_dma": mode, "ms_per_1000_ticks": round(float(np.median(v)), 4),
                          "G_steps_per_s": round(B * K / float(np.median(v)) / 1e6, 2)}), flush=True)
    del log, plan, fleet
