"""BASELINE config 2 (B = 4 096, m = 8): the logged rollout alone, for `rocprofv3 --pmc ...` -- where do the cycles of
its 64 compute waves go?  Run as
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS \\
              --kernel-trace --output-format csv -d gpurun_out/pmc_small -o pmc -- python3 tools/small_batch_profile.py
and without the profiler for the plain timing."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import torch
from bench import missions
from uav_ac.fleet import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng = Engine("cuda:0")
plan = eng.plan(missions(B, 8, 0, B), 3.0, 0.01)
fleet = eng.fleet(plan)
log = torch.empty((1000, 13, B), dtype=torch.float64, device="cuda:0")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
t_log = timed(lambda: fleet.rollout(1000, state_log=log))
t_nolog = timed(lambda: fleet.rollout(1000))
print(json.dumps({"B": B, "kernel": eng.ctx.last_rollout_kernel(), "ms_per_1000_ticks_logged": t_log, "ms_per_1000_ticks_no_log": t_nolog,
                  "G_steps_per_s_logged": B * 1000 / t_log / 1e6, "ns_per_tick_logged": t_log * 1e3, "ns_per_tick_no_log": t_nolog * 1e3}))
