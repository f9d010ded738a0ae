"""The literal drop-in loop of uav_ac/main.py:113-118 -- `tc.step(); simulation.step()` once per inner tick, one UAV, all
state owned by Python objects -- timed through the facade classes (each call = one kernel launch + one stream
synchronisation on state in pinned mapped memory), next to the reference's own 77 us per step on one CPU core
(BASELINE.md section 2).  Also the same flight as ONE fused launch (`fly_mission`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]
import numpy as np
from uav_ac.control.controller import CascadedController
from uav_ac.main import TrajectoryController, _generate_mission_trajectory, fly_mission
from uav_ac.simulation.mujoco_sim import MujocoSimulation, DEFAULT_SCENE_PATH

sim = MujocoSimulation()
quad = sim.quad
F = 10
traj = _generate_mission_trajectory(sim.mission_waypoints, sim.obstacles, 2.0, quad.dt * F)
tc = TrajectoryController(CascadedController(g=quad.g, dt=quad.dt * F), quad, traj, F)
for _ in range(200):
    tc.step(); sim.step()
n = 4000
t0 = time.perf_counter()
for _ in range(n):
    tc.step()
t1 = time.perf_counter()
for _ in range(n):
    sim.step()
t2 = time.perf_counter()
for _ in range(n):
    tc.step(); sim.step()
t3 = time.perf_counter()
print(f"tc.step()              : {(t1 - t0) / n * 1e6:7.1f} us per call")
print(f"sim.step()             : {(t2 - t1) / n * 1e6:7.1f} us per call")
print(f"tc.step() + sim.step() : {(t3 - t2) / n * 1e6:7.1f} us per tick   (reference, one CPU core: 77 us controller step alone, BASELINE.md 2)")
t0 = time.perf_counter()
out = fly_mission(DEFAULT_SCENE_PATH, velocity=2.0, frequency=F, settle_ticks=0)
t1 = time.perf_counter()
k = len(out["states"])
print(f"fly_mission (plan + {k} ticks in one fused launch, B = 1): {(t1 - t0) * 1e3:.1f} ms end to end = {(t1 - t0) / k * 1e6:.2f} us per tick")
