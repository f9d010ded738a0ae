"""Is a row buffer that is freed (back to the driver: empty_cache) and allocated again a fresh draw of its kind?
One buffer alive at a time, twelve draws; then the same holding the best one so far (two alive)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
eng = Engine("cuda:0")
eng.ctx.set_option("sampler_waves", 1)                  # the one-wave sampler tells the kinds apart best
plan = eng.plan(missions(65536, 12, 0, 65536), 3.0, 0.01, placement_trials=1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed():
    eng.sample(plan); eng.sample(plan)
    a.record()
    for _ in range(3): eng.sample(plan)
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / 3, 3)
shape = plan.traj.shape
out = [(hex(plan.traj.data_ptr()), timed())]
for _ in range(11):
    plan.traj = None
    torch.cuda.empty_cache()
    plan.traj = torch.empty(shape, dtype=torch.float64, device="cuda:0")
    out.append((hex(plan.traj.data_ptr()), timed()))
print("one alive:", out)
best = plan.traj; best_t = out[-1][1]; out2 = []
for _ in range(10):
    cand = torch.empty(shape, dtype=torch.float64, device="cuda:0")
    plan.traj = cand
    t = timed()
    out2.append((hex(cand.data_ptr()), t, "kept" if t < best_t else "freed"))
    if t < best_t: best, best_t = cand, t
    del cand
    plan.traj = best
    torch.cuda.empty_cache()
print("best + one candidate alive:", out2, "peak GB", torch.cuda.max_memory_allocated() / 1e9)
