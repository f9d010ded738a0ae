"""Why are the first two rollout launches of every bench step ~30 % slower than the other eight (round-1 VERDICT,
weak point 2)?  Per-launch HIP-event times of the bench's own launch sequence under variations that separate an
after-effect of the planning kernels (caches, address translation, clocks) from a property of the flight phase those
launches simulate (ticks 0-2000: take-off from rest, yaw alignment).

    python3 tools/first_launch_bisect.py [B]      -> one JSON line per arm + a table
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
import torch
from bench import missions
from uav_ac.fleet import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
M, CH, NL = 12, 1000, 10
dev = "cuda:0"
eng = Engine(dev)
wps = missions(B, M, 0, B)
plan = eng.plan(wps, 3.0, 0.01, dense_yaw=True)
fleet = eng.fleet(plan)
log = torch.empty((CH, 13, B), dtype=torch.float64, device=dev)
log2 = torch.empty((CH, 13, B), dtype=torch.float64, device=dev)


def ev():
    return torch.cuda.Event(enable_timing=True)


def launches(n=NL, fl=fleet, buf=log, K=CH):
    out = []
    for _ in range(n):
        a, b = ev(), ev()
        a.record()
        fl.rollout(K, state_log=buf)
        b.record()
        out.append((a, b))
    return out


def run(name, step, reps=4):
    step()
    torch.cuda.synchronize()
    acc = []
    for _ in range(reps):
        pairs = step()
        torch.cuda.synchronize()
        acc.append([a.elapsed_time(b) for a, b in pairs])
    t = np.mean(np.array(acc), axis=0)
    print(json.dumps({"arm": name, "ms": [round(float(x), 3) for x in t]}), flush=True)
    return name, t


arms = []
# A: the bench step itself
def a_bench():
    eng.solve(plan); eng.sample(plan); fleet.reset()
    return launches()
arms.append(run("A bench step: solve+sample, reset, 10 launches", a_bench))

# B: no planning kernels between steps -> is it the flight phase?
def b_noplan():
    fleet.reset()
    return launches()
arms.append(run("B reset + 10 launches (no planning kernels)", b_noplan))

# C: planning kernels, but the fleet keeps flying far beyond the mission end (holding the last row)
fleet.reset(); fleet.rollout(20000)
def c_noreset():
    eng.solve(plan); eng.sample(plan)
    return launches()
arms.append(run("C solve+sample + 10 launches, no reset (fleet holds last row)", c_noreset))

# D: only launches in the holding regime
def d_hold():
    return launches()
arms.append(run("D 10 launches, holding (no planning, no reset)", d_hold))

# E: bench step with a dummy 1-tick rollout between planning and the timed launches
def e_dummy():
    eng.solve(plan); eng.sample(plan); fleet.reset()
    fleet.rollout(0 + 1)                      # 1 tick without a log
    return launches()
arms.append(run("E bench step + 1-tick rollout before the launches", e_dummy))

# F: bench step, alternate log buffer per step
flip = [0]
def f_altlog():
    eng.solve(plan); eng.sample(plan); fleet.reset()
    flip[0] ^= 1
    return launches(buf=log2 if flip[0] else log)
arms.append(run("F bench step, log buffer alternates between steps", f_altlog))

# G: row-fed rollout, bench step
fleet_rows = eng.fleet(plan, from_plan=False)
def g_rows():
    eng.solve(plan); eng.sample(plan); fleet_rows.reset()
    return launches(fl=fleet_rows)
arms.append(run("G bench step, row-fed kernel", g_rows))

# H: yaw removed from the picture: the vehicles start aligned with their first leg (target yaw == heading at start)
# -> done by zeroing the dense yaw column and the yaw column of the rows after sampling (psi_des = 0 = initial psi)
fleet_col = eng.fleet(plan, yaw_from="column")          # this arm needs the rollout that READS the yaw column
def h_noyaw():
    eng.solve(plan); eng.sample(plan); plan.yaw.zero_(); fleet_col.reset()
    return launches(fl=fleet_col)
arms.append(run("H bench step with the target yaw forced to 0 (no initial yaw slew)", h_noyaw))
eng.sample(plan)

# I: 20 launches of 500 ticks: where inside the flight does the time go?
def i_fine():
    eng.solve(plan); eng.sample(plan); fleet.reset()
    return launches(n=20, K=500)
arms.append(run("I bench step as 20 launches x 500 ticks", i_fine))

# J: flight-state census at the end of launch 1, 2, 3: saturated rotors, large-angle lanes, yaw error
fleet.reset()
V = fleet.vehicle
for k in range(1, 5):
    fleet.rollout(CH, state_log=log)
    torch.cuda.synchronize()
    st = fleet.state
    om_c = st[17:21]
    fmin, fmax = V.min_thrust, V.max_thrust
    f = V.kf * om_c * om_c
    sat = ((f <= fmin * (1 + 1e-12)) | (f >= fmax * (1 - 1e-12))).any(dim=0).double().mean().item()
    w = st[10:13].norm(dim=0)
    print(json.dumps({"census_after_launch": k, "frac_lanes_rotor_saturated": sat,
                      "max_body_rate": float(w.max()), "frac_body_rate_gt_63": float((w > 63.0).double().mean()),
                      "frac_nonfinite": float((~torch.isfinite(st).all(dim=0)).double().mean())}), flush=True)

print("\n| arm | " + " | ".join(str(i + 1) for i in range(NL)) + " |")
print("|---|" + "---|" * NL)
for name, t in arms:
    print(f"| {name} | " + " | ".join(f"{x:.2f}" for x in t[:NL]) + (" | ..." if len(t) > NL else " |"))
    if len(t) > NL:
        print(f"|   (cont.) | " + " | ".join(f"{x:.2f}" for x in t[NL:]) + " |")
