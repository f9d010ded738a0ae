"""Single-UAV facade with the surface of the reference's `uav_ac/main.py`: `TrajectoryController`
(one inner body-rate cycle per `step()`, outer loop every `inner_loop_frequency` steps) and the
mission-assembly helpers.  `step()` is one call of `uavac_controller_tick` (B = 1); the vehicle
state is whatever the caller's simulation wrote into `quad.X`, as in the reference.  A free-flight
stand-in for `MujocoSimulation.step()` is `FreeFlightSimulation` below.  Fleet-scale: `uav_ac.fleet`."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat
from ._single import ctx, vehicle_from
from .control.controller import CascadedController
from .planning.minimum_snap import MinimumSnap
from .quadrotor.quad import Quad


class TrajectoryController:
    def __init__(self, controller: CascadedController, quad: Quad, trajectory: np.ndarray, inner_loop_frequency: int):
        self.controller = controller
        self.quad = quad
        self.trajectory = trajectory
        self.inner_loop_frequency = inner_loop_frequency
        self.trajectory_index = 0
        self.inner_step = 0
        self.thrust_cmd = 0.0
        self.pqr_cmd = np.zeros(3)

    def reset(self) -> None:
        self.controller.reset()
        self.trajectory_index = 0
        self.inner_step = 0
        self.thrust_cmd = 0.0
        self.pqr_cmd.fill(0.0)

    def _pilot(self) -> nat.Pilot:
        """Resident session for this trajectory (`uavac_pilot_*`): the rows go to the GPU once, not once per tick.
        The reference reads `self.trajectory[self.trajectory_index]` live on every outer tick (main.py:48), so an edit in
        place must not be missed: besides the identity of the object, the row an outer tick is about to consume is compared
        with what was uploaded (11 doubles), and the rows are uploaded again when it differs."""
        traj = self.trajectory
        key = (id(traj), getattr(traj, "shape", None), traj.ctypes.data if isinstance(traj, np.ndarray) else None, len(traj))
        stale = getattr(self, "_pilot_key", None) != key
        if not stale and self.inner_step % max(int(self.inner_loop_frequency), 1) == 0 and len(traj):
            i = min(max(int(self.trajectory_index), 0), len(traj) - 1)
            stale = not np.array_equal(np.asarray(traj[i], dtype=np.float64), self._pilot_rows[i])
        if stale:
            rows = nat.as_f64(traj)
            self._pilot_obj = nat.Pilot(ctx(), rows, np.array([0, len(rows)], dtype=np.int64))
            self._pilot_rows = rows.copy()          # what the GPU holds
            self._pilot_key = key
        return self._pilot_obj

    def step(self) -> None:
        """One `tc.step()` of the reference (main.py:37-61) on the GPU: `uavac_pilot_tick` (controller half) on state
        that lives in pinned memory the kernel reads and writes in place."""
        q = self.quad
        V = vehicle_from(q, g=self.controller.g, dt_outer=self.controller.dt)
        V.inner_per_outer = int(self.inner_loop_frequency)
        p = self._pilot()
        st, ist = p.state[:, 0], p.istate[:, 0]
        st[0:13] = q.X
        st[13:17] = q.omega
        st[17:21] = q.omega_command
        st[21] = self.controller.integral_error
        st[22] = self.thrust_cmd
        st[23:26] = self.pqr_cmd
        ist[0], ist[1] = self.trajectory_index, self.inner_step
        p.tick(V, nat.PILOT_CONTROLLER)
        q.omega = st[13:17].copy()
        q.omega_command = st[17:21].copy()
        self.controller.integral_error = float(st[21])
        self.thrust_cmd = float(st[22])
        self.pqr_cmd = st[23:26].copy()
        self.trajectory_index = int(ist[0])
        self.inner_step = int(ist[1])


class FreeFlightSimulation:
    """Free-flight subset of the reference's `MujocoSimulation` (mujoco_sim.py:144-151,232-251): `step()`
    applies the rotor wrench and advances `quad.X` by one semi-implicit Euler step on the GPU.  No contacts,
    no scene, no viewer."""

    def __init__(self, quad: Quad, obstacles=None):
        self.quad = quad
        self.obstacles = None if obstacles is None else nat.as_f64(obstacles).reshape(-1, 6)
        self.collision_detected = False
        self._pilot = None

    def step(self) -> np.ndarray:
        q = self.quad
        V = vehicle_from(q)
        if self._pilot is None:
            self._pilot = nat.Pilot(ctx(), np.zeros((1, nat.TRAJ_COLS)), np.array([0, 1], dtype=np.int64))
            self._pilot.set_obstacles(self.obstacles)
        st, ist = self._pilot.state[:, 0], self._pilot.istate[:, 0]
        st[0:13] = q.X
        st[13:17] = q.omega
        ist[2] = 0
        self._pilot.tick(V, nat.PILOT_DYNAMICS)
        q.X = st[0:13].copy()
        self.collision_detected = self.collision_detected or bool(ist[2])
        return q.X.copy()


def _trajectory_after_takeoff(trajectory: np.ndarray, takeoff_waypoint: np.ndarray) -> np.ndarray:
    """Trajectory from the sample nearest the takeoff waypoint (reference main.py:64-70)."""
    d = np.linalg.norm(trajectory[:, :3] - takeoff_waypoint, axis=1)
    return trajectory[np.argmin(d):]


def _generate_mission_trajectory(waypoints: np.ndarray, obstacles, velocity: float, dt: float) -> np.ndarray:
    """Isolated vertical takeoff spline followed by the course (reference main.py:73-84)."""
    takeoff = MinimumSnap(waypoints[:2], obstacles, velocity, dt).get_trajectory()
    course = MinimumSnap(waypoints[1:], obstacles, velocity, dt).get_trajectory()
    return np.vstack((takeoff, course))


def fly_mission(model_path, velocity: float = 3.0, frequency: int = 10, min_distance_target: float = 0.5,
                settle_ticks: int = 2000, ground: bool = True):
    """Head-less equivalent of the reference's `main()` (main.py:87-120): read the scene, plan takeoff + course
    around its obstacles, fly it from the scene's start state -- on the ground, rotors stopped (lab_course.xml:98)
    -- and report.  The whole flight is ONE fused rollout on the GPU (B = 1) with the per-tick AABB flag and, when
    the scene has a ground plane and `ground` is true, the build-defined ground contact + take-off bookkeeping
    (include/uavac.h, `uavac_vehicle.ground`); `ground=False` flies the same mission in free flight.  No viewer.

    Returns a dict: trajectory (N,11), states (K,13), distance_to_goal, goal_reached, collision_detected (an obstacle
    entered, or the ground touched after take-off: mujoco_sim.py:220-230), ground_bits, mean_tracking_error.
    """
    from .fleet import Engine
    from .simulation.mujoco_sim import MujocoSimulation

    sim = MujocoSimulation(model_path)
    quad = sim.quad
    dt_traj = quad.dt * frequency
    trajectory = _generate_mission_trajectory(sim.mission_waypoints, sim.obstacles, velocity, dt_traj)

    eng = Engine()
    torch = eng._torch
    rows = torch.as_tensor(trajectory, dtype=torch.float64, device=eng.device).contiguous()
    offsets = torch.tensor([0, len(trajectory)], dtype=torch.int64, device=eng.device)
    V = sim.vehicle(dt_outer=dt_traj)
    if not ground:
        V.ground = 0
    V.inner_per_outer = int(frequency)
    state = torch.zeros((nat.STATE_ROWS, 1), dtype=torch.float64, device=eng.device)
    state[0:13, 0] = torch.as_tensor(quad.X)
    istate = torch.zeros((nat.ISTATE_ROWS, 1), dtype=torch.int32, device=eng.device)
    K = len(trajectory) * frequency + settle_ticks
    log = torch.empty((K, 13, 1), dtype=torch.float64, device=eng.device)
    aabbs = torch.as_tensor(sim.obstacles, dtype=torch.float64, device=eng.device).contiguous()
    P = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    eng._bind_stream()
    eng.ctx.call("uavac_control_rollout_dev", C.byref(V), P(rows), P(offsets), P(state), P(istate), 1, K, P(log), None,
                 P(aabbs), int(aabbs.shape[0]))
    states = log[:, :, 0].cpu().numpy()
    quad.X = state[0:13, 0].cpu().numpy()
    n = min(len(trajectory), K // frequency)
    err = np.linalg.norm(states[::frequency][:n, 0:3] - trajectory[:n, 0:3], axis=1)
    dist = float(np.linalg.norm(quad.position - sim.goal_position))
    return {"trajectory": trajectory, "states": states, "distance_to_goal": dist,
            "goal_reached": dist < min_distance_target,
            "collision_detected": bool(istate[2, 0].item()) or bool(int(istate[3, 0].item()) & nat.GROUND_HIT_AFTER_TAKEOFF),
            "ground_bits": int(istate[3, 0].item()), "mean_tracking_error": float(err.mean())}


def main(model_path=None) -> None:
    import sys
    from .simulation.mujoco_sim import DEFAULT_SCENE_PATH
    path = model_path or (sys.argv[1] if len(sys.argv) > 1 else DEFAULT_SCENE_PATH)
    from . import utils
    cfg, cfg_flight = utils.get_config()                     # reference main.py:88-93
    out = fly_mission(path, velocity=cfg_flight.getfloat("velocity"), frequency=cfg.getint("frequency"),
                      min_distance_target=cfg_flight.getfloat("min_dist_target"))
    print(f"Flight finished {out['distance_to_goal']:.2f} m away from the goal "
          f"({'reached' if out['goal_reached'] else 'missed'}).")
    if out["collision_detected"]:
        print("At least one collision occurred during the flight.")


if __name__ == "__main__":
    main()
