"""The sampler with 1 (minsnap_sample.hip), 4, 8 and 16 (minsnap_sample_wide.hip) wavefronts per mission, on N row buffers
allocated side by side (each buffer is of the fast or the slow kind for as long as it lives, DESIGN K2), alternating.
    python3 tools/sampler_wide_ab.py [n_buffers] [B] [m]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
m = int(sys.argv[3]) if len(sys.argv) > 3 else 12
eng = Engine("cuda:0")
plan = eng.plan(missions(B, m, 0, B), 3.0, 0.01, placement_trials=1)
bufs = [plan.traj] + [torch.empty_like(plan.traj) for _ in range(NB - 1)]
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {w: [[] for _ in bufs] for w in (1, 4, 8, 16)}
for rnd in range(3):
    for i, t in enumerate(bufs):
        plan.traj = t
        for w in (1, 4, 8, 16):
            eng.ctx.set_option("sampler_waves", w)
            eng.sample(plan); eng.sample(plan)
            a.record()
            for _ in range(5):
                eng.sample(plan)
            b.record(); torch.cuda.synchronize()
            res[w][i].append(a.elapsed_time(b) / 5)
for w in res:
    print(json.dumps({"B": B, "m": m, "rows": plan.total_rows, "sampler_waves": w,
                      "ms_per_buffer": [round(min(x), 4) for x in res[w]]}))
# the whole planning chain (one C call: counts, offsets, solve, sampler) on the first and on the best buffer
for w in (1, 16):
    eng.ctx.set_option("sampler_waves", w)
    out = []
    for t in bufs:
        plan.traj = t
        for _ in range(3):
            eng.replan(plan)
        a.record()
        for _ in range(5):
            eng.replan(plan)
        b.record(); torch.cuda.synchronize()
        out.append(round(a.elapsed_time(b) / 5, 4))
    print(json.dumps({"planning_chain_ms_per_buffer": out, "sampler_waves": w,
                      "frac_of_8TBps": [round(plan.algorithmic_bytes / (x * 1e-3) / 8e12, 3) for x in out]}))
