#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE.

Runs only in the build container, where the upstream repository is mounted
read-only at /root/reference (it does not exist on the GPU box; nothing under
tests/ reads it at test time).  Only the *.npz outputs are committed: they are
data (inputs + the reference's outputs), never reference source.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

`mujoco` is not installed here; `uav_ac.main` imports it at module import, so a
MagicMock stands in for the module (TrajectoryController and
_generate_mission_trajectory never touch it).  `Quad` is built from the
constants of uav_ac/simulation/models/lab_course.xml exactly as
mujoco_sim.py:258-279 would pass them.

The closed-loop fixture (closed_loop.npz) drives the REFERENCE controller /
allocation / motor model with the BUILD-DEFINED free-body step
(oracle/control_oracle.py:dynamics_step) because the
reference's dynamics is MuJoCo's mj_step, which cannot run here.
"""
import os
import sys
import unittest.mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("UAVAC_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(1, REPO)
sys.modules["mujoco"] = unittest.mock.MagicMock()
sys.modules["mujoco.viewer"] = unittest.mock.MagicMock()

from uav_ac.planning.minimum_snap import MinimumSnap            # noqa: E402  (reference)
from uav_ac.control.controller import CascadedController        # noqa: E402  (reference)
from uav_ac.quadrotor.quad import Quad                          # noqa: E402  (reference)
from uav_ac.main import TrajectoryController, _generate_mission_trajectory  # noqa: E402

from oracle import control_oracle as co                         # noqa: E402  (ours: dynamics only)
from oracle.minsnap_oracle import synthetic_missions            # noqa: E402  (ours: input generator)

assert os.path.realpath(sys.modules["uav_ac.planning.minimum_snap"].__file__).startswith(os.path.realpath(REF))

LAB_WAYPOINTS = np.array([            # tests/unit/simulation/test_mujoco_sim.py:40-50
    [1.0, 7.0, -0.021], [1.0, 7.0, -1.3], [4.0, 7.0, -1.3], [7.5, 4.0, -3.0], [11.0, 7.0, -3.5],
    [14.0, 10.0, -2.5], [17.0, 10.0, -3.2], [20.5, 7.0, -1.4], [23.0, 7.0, -2.0]])
LAB_AABBS = np.array([                # lab_course.xml:37,53,54,67 through mujoco_sim.py:282-300
    [3.7, 4.3, 4.0, 10.0, -3.4, -2.8], [10.7, 11.3, 4.0, 10.0, -2.2, 0.0],
    [13.3, 14.7, 6.3, 7.7, -6.0, 0.0], [20.2, 20.8, 4.0, 10.0, -3.3, -2.7]])
CONFIG1_WAYPOINTS = LAB_WAYPOINTS[1:6].copy()


def make_quad() -> Quad:
    return Quad(g=9.81, dt=0.001, mass=0.5, inertia=np.array([0.0023, 0.0023, 0.0046]),
                arm_length=0.120208, force_coefficient=1.0, drag_to_thrust=0.016,
                thrust_limits=np.array([0.1, 4.5]), motor_time_constants=np.array([0.0125, 0.025]),
                flight_limits=np.array([3.0, 2.0, 3.0, 12.0, 0.7]))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  keys={sorted(arrays)}")


# ------------------------------------------------------------------ planner
def gen_polynom():
    ts = np.array([0.0, 0.01, 0.5, 1.37, 3.0])
    out = np.array([[MinimumSnap.polynom(8, k, t) for t in ts] for k in range(7)])
    save("polynom.npz", t=ts, rows=out)


def ref_plan(wp, obstacles, velocity, dt, method="lstsq"):
    ms = MinimumSnap(np.array(wp, dtype=float), obstacles, velocity, dt)
    if obstacles is None and method != "lstsq":
        ms._generate_trajectory(method)
        return ms
    ms.get_trajectory()
    return ms


def gen_fixed_missions():
    ms = ref_plan(CONFIG1_WAYPOINTS, None, 3.0, 0.01)
    ms_solve = ref_plan(CONFIG1_WAYPOINTS, None, 3.0, 0.01, method="solve")
    H = ms._create_snap_cost_matrix()
    lab_free = _generate_mission_trajectory(LAB_WAYPOINTS.copy(), None, 3.0, 0.01)
    lab_obs = _generate_mission_trajectory(LAB_WAYPOINTS.copy(), LAB_AABBS.copy(), 3.0, 0.01)
    course = MinimumSnap(LAB_WAYPOINTS[1:].copy(), LAB_AABBS.copy(), 3.0, 0.01)
    course.get_trajectory()
    # in-tree obstacle case: tests/unit/planning/test_minimum_snap.py:171-183
    wp_t = np.array([[0., 0., 1.], [3., 0., 1.], [3., 3., 1.]])
    obs_t = np.array([[3.2, 4.0, 0.5, 1.5, 0., 2.]])
    case = MinimumSnap(wp_t.copy(), obs_t.copy(), velocity=2.0, dt=0.01)
    case_traj = case.get_trajectory()
    save("fixed_missions.npz",
         config1_wp=CONFIG1_WAYPOINTS, config1_times=np.array(ms.times), config1_A=ms.A, config1_b=ms.b,
         config1_H=H, config1_coeffs=ms.coeffs, config1_coeffs_solve=ms_solve.coeffs,
         config1_traj=ms.full_trajectory,
         lab_wp=LAB_WAYPOINTS, lab_aabbs=LAB_AABBS, lab_traj_free=lab_free, lab_traj_obs=lab_obs,
         lab_course_final_wp=np.asarray(course.waypoints),
         obs_case_wp=wp_t, obs_case_aabb=obs_t, obs_case_traj=case_traj,
         obs_case_final_wp=np.asarray(case.waypoints))


def gen_synthetic_missions():
    """missions from the 8(d) generator for m in {1,2,8,12,20}."""
    out = {}
    for m, nb, nfull in ((1, 16, 2), (2, 16, 2), (8, 32, 2), (12, 32, 2), (20, 16, 1)):
        wps = synthetic_missions(nb, m)
        times, c_l, c_s, counts, sub = [], [], [], [], []
        for i in range(nb):
            ms = ref_plan(wps[i], None, 3.0, 0.01)
            times.append(ms.times)
            c_l.append(ms.coeffs)
            c_s.append(ref_plan(wps[i], None, 3.0, 0.01, method="solve").coeffs)
            tr = ms.full_trajectory
            counts.append(np.bincount(tr[:, 10].astype(int), minlength=m))
            sub.append(tr[::16])
            if i < nfull:
                out[f"m{m}_traj{i}"] = tr
        out[f"m{m}_wp"] = wps
        out[f"m{m}_times"] = np.array(times)
        out[f"m{m}_coeffs_lstsq"] = np.array(c_l)
        out[f"m{m}_coeffs_solve"] = np.array(c_s)
        out[f"m{m}_rows_per_segment"] = np.array(counts)
        out[f"m{m}_traj_every16"] = np.vstack(sub)
    save("synthetic_missions.npz", **out)


def gen_yaws():
    rng = np.random.default_rng(7)
    cases = {
        "hold": np.array([[0., 0., -1.], [0., 2., 0.], [0., 0., 1.]]),       # test_minimum_snap.py:105-114
        "cross_pi": np.array([[-1., 0.01, 0.], [-1., -0.01, 0.]]),           # :117-126
        "none_valid": np.array([[0., 0., -1.], [0., 0., 0.], [0., 0., 1.]]),  # :129-136
        "exact_pi_steps": np.array([[1., 0., 0.], [-1., 0., 0.], [1., 0., 0.], [-1., -0., 0.], [0., -1., 0.]]),
        "leading_invalid": np.vstack([np.zeros((5, 3)), [[1e-4, 0, 0]], [[-1, 1e-3, 0]], [[-1, -1e-3, 0]],
                                      np.zeros((3, 3)), [[0.5, -0.5, 1]]]),
    }
    ang = np.cumsum(rng.uniform(-1.5, 1.5, 400))
    spin = np.stack([np.cos(ang), np.sin(ang), rng.standard_normal(400)], axis=1) * rng.uniform(0, 2, (400, 1))
    spin[rng.random(400) < 0.2] *= 1e-5
    cases["random_spin"] = spin
    out = {}
    for k, v in cases.items():
        out[k + "_vel"] = v
        out[k + "_yaw"] = MinimumSnap._calculate_yaws(v)
    save("yaws.npz", **out)


def gen_derivatives():
    """Jerk and snap samples exactly as minimum_snap.py:111-112 would compute them (the two lines are comments
    upstream): polynom(8, 3 | 4, t) @ coeffs over the sampler's own np.arange(0, T, dt) grid, with the reference's
    polynom and the reference's coefficients."""
    out = {}
    cases = {"config1": CONFIG1_WAYPOINTS}
    for i, w in enumerate(synthetic_missions(3, 12)):
        cases[f"m12_{i}"] = w
    for name, wp in cases.items():
        ms = ref_plan(wp, None, 3.0, 0.01)
        jerk, snap = [], []
        for it in range(ms.nb_splines):
            c = ms.coeffs[it * 8:(it + 1) * 8]
            for t in np.arange(0.0, ms.times[it], ms.dt):
                jerk.append(MinimumSnap.polynom(8, 3, t) @ c)
                snap.append(MinimumSnap.polynom(8, 4, t) @ c)
        out[name + "_wp"] = np.asarray(wp)
        out[name + "_coeffs"] = ms.coeffs
        out[name + "_jerk"] = np.array(jerk)
        out[name + "_snap"] = np.array(snap)
        assert len(jerk) == len(ms.full_trajectory)
    save("derivatives.npz", **out)


def gen_yaws_long():
    """_calculate_yaws on sequences far longer than one mission (any length is legal upstream)."""
    rng = np.random.default_rng(11)
    out = {}
    for name, n in (("n5000", 5000), ("n20000", 20000)):
        ang = np.cumsum(rng.uniform(-0.9, 0.9, n))
        v = np.stack([np.cos(ang), np.sin(ang), rng.standard_normal(n)], axis=1) * rng.uniform(0, 2, (n, 1))
        v[rng.random(n) < 0.15] *= 1e-5
        v[:300] *= 1e-6                                   # a long run of rows without heading first (back-fill)
        out[name + "_vel"] = v
        out[name + "_yaw"] = MinimumSnap._calculate_yaws(v)
    save("yaws_long.npz", **out)


# --------------------------------------------------------------- controller
def random_states(rng, n):
    X = np.zeros((n, 13))
    X[:, 0:3] = rng.uniform(-25, 25, (n, 3))
    tilt = rng.uniform(0, 1.2, n)
    az = rng.uniform(0, 2 * np.pi, n)
    yaw = rng.uniform(-np.pi, np.pi, n)
    for i in range(n):
        ax = np.array([np.cos(az[i]), np.sin(az[i]), 0.0])
        qt = np.concatenate([[np.cos(tilt[i] / 2)], np.sin(tilt[i] / 2) * ax])
        qy = np.array([np.cos(yaw[i] / 2), 0, 0, np.sin(yaw[i] / 2)])
        a, b = qy, qt
        X[i, 3:7] = [a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]]
    X[:, 3:7] *= rng.uniform(0.98, 1.02, (n, 1))     # stored quaternion is not exactly unit
    X[:, 7:10] = rng.uniform(-5, 5, (n, 3))
    X[:, 10:13] = rng.uniform(-5, 5, (n, 3))
    return X


def gen_controller_io(n=2048):
    rng = np.random.default_rng(11)
    quad = make_quad()
    X = random_states(rng, n)
    omega0 = rng.uniform(0, 2.2, (n, 4))
    tgt = np.zeros((n, 11))
    tgt[:, 0:3] = X[:, 0:3] + rng.standard_normal((n, 3)) * np.array([1.0, 1.0, 0.5]) * rng.choice([0.05, 0.5, 3.0], (n, 1))
    tgt[:, 3:6] = rng.uniform(-4.5, 4.5, (n, 3))
    tgt[:, 6:9] = rng.uniform(-6, 6, (n, 3))
    tgt[:, 9] = rng.uniform(-3 * np.pi, 3 * np.pi, n)
    integ0 = rng.uniform(-10.5, 10.5, n)
    pqr_cmd = rng.uniform(-6, 6, (n, 3)) * rng.choice([0.1, 1.0, 8.0], (n, 1))
    thrust_in = rng.uniform(-2, 22, n)

    R = np.empty((n, 3, 3)); eul = np.empty((n, 3)); thrust = np.empty(n); integ1 = np.empty(n)
    bxy = np.empty((n, 2)); pq = np.empty((n, 2)); pqr = np.empty((n, 3)); moment = np.empty((n, 3))
    forces = np.empty((n, 4)); omega_cmd = np.empty((n, 4)); omega1 = np.empty((n, 4))
    for i in range(n):
        quad.X = X[i].copy()
        ctrl = CascadedController(9.81, 0.01)
        ctrl.integral_error = integ0[i]
        R[i] = quad.R()
        eul[i] = quad.euler_angles
        thrust[i] = ctrl.altitude(quad, tgt[i, [2, 5, 8]], R[i], quad.kp_z, quad.kd_z, quad.ki_z)
        integ1[i] = ctrl.integral_error
        bxy[i] = ctrl.lateral(quad, tgt[i, [0, 3, 6]], tgt[i, [1, 4, 7]], thrust[i], quad.kp_xy, quad.kd_xy)
        pq[i] = ctrl.roll_pitch_controller(bxy[i], R[i], quad.kp_roll, quad.kp_pitch)
        pqr[i] = ctrl.reduced_attitude(quad, bxy[i], tgt[i, 9], R[i], quad.kp_roll, quad.kp_pitch, quad.kp_yaw)
        moment[i] = ctrl.body_rate_controller(quad, pqr_cmd[i], quad.kp_p, quad.kp_q, quad.kp_r)
        forces[i] = quad._allocate_rotor_forces(thrust_in[i], moment[i])
        quad.omega = omega0[i].copy()
        quad.set_propeller_speed(thrust_in[i], moment[i])
        omega_cmd[i] = quad.omega_command
        omega1[i] = quad.omega
    # integral clamp sequence: test_controller.py:136-146
    quad.X = np.zeros(13); quad.X[3] = 1.0
    ctrl = CascadedController(9.81, 0.01)
    seq = np.empty(2500)
    for k in range(2500):
        ctrl.altitude(quad, np.array([5.0, 0.0, 0.0]), np.eye(3), quad.kp_z, quad.kd_z, quad.ki_z)
        seq[k] = ctrl.integral_error
    gains = np.array([quad.kp_xy, quad.kd_xy, quad.kp_z, quad.kd_z, quad.ki_z, quad.kp_roll, quad.kp_pitch,
                      quad.kp_yaw, quad.kp_p, quad.kp_q, quad.kp_r])
    wraps_in = np.concatenate([rng.uniform(-20, 20, 200), [0.1, 2 * np.pi + 0.1, -1.5 * np.pi, 1.5 * np.pi,
                                                             -0.1, np.pi, -np.pi, 0.0, 3 * np.pi]])
    save("controller_io.npz", X=X, omega0=omega0, target=tgt, integ0=integ0, pqr_cmd_in=pqr_cmd,
         thrust_in=thrust_in, R=R, euler=eul, thrust=thrust, integ1=integ1, bxy=bxy, pq=pq, pqr=pqr,
         moment=moment, forces=forces, omega_cmd=omega_cmd, omega1=omega1, integral_sequence=seq, gains=gains,
         wraps_in=wraps_in,
         wrap_pi=np.array([CascadedController.wrap_to_pi(a) for a in wraps_in]),
         wrap_2pi=np.array([CascadedController.wrap_to_2pi(a) for a in wraps_in]))


def gen_open_loop():
    """8(c)-6: TrajectoryController on a frozen state, 50 ticks -> scheduling of main.py:37-61."""
    rng = np.random.default_rng(13)
    traj = ref_plan(CONFIG1_WAYPOINTS, None, 3.0, 0.01).full_trajectory
    quad = make_quad()
    X0 = random_states(rng, 1)[0]
    X0[0:3] = CONFIG1_WAYPOINTS[0] + np.array([0.05, -0.08, 0.03])
    X0[7:13] *= 0.1
    quad.X = X0.copy()
    quad.omega = np.full(4, np.sqrt(0.5 * 9.81 / 4))
    tc = TrajectoryController(CascadedController(9.81, 0.01), quad, traj[40:], 10)
    log = np.empty((50, 14))
    for k in range(50):
        tc.step()
        log[k] = np.concatenate([[tc.thrust_cmd], tc.pqr_cmd, quad.omega_command, quad.omega,
                                 [tc.trajectory_index, tc.controller.integral_error]])
    save("open_loop.npz", X0=X0, traj=traj[40:120], log=log)


class _Shim:
    """Lets oracle.control_oracle.dynamics_step advance the reference Quad's state."""
    def __init__(self, quad):
        self.V = co.Vehicle()
        self.quad = quad

    @property
    def X(self):
        return self.quad.X

    @property
    def omega(self):
        return self.quad.omega


def gen_closed_loop():
    """8(c)-7: reference controller + build-defined dynamics on (i) config 1 and (ii) the lab course."""
    out = {}
    for name, traj, K in (
            ("config1", ref_plan(CONFIG1_WAYPOINTS, None, 3.0, 0.01).full_trajectory, 8000),
            ("lab_v2", _generate_mission_trajectory(LAB_WAYPOINTS.copy(), LAB_AABBS.copy(), 2.0, 0.01), 17000)):
        quad = make_quad()
        quad.X[0:3] = traj[0, 0:3]
        w = np.sqrt(quad.m * quad.g / (4 * quad.kf))
        quad.omega = np.full(4, w)
        quad.omega_command = np.full(4, w)
        tc = TrajectoryController(CascadedController(quad.g, quad.dt * 10), quad, traj, 10)
        shim = _Shim(quad)
        slog = np.empty((K, 13)); clog = np.empty((K, 12))
        for k in range(K):
            tc.step()
            clog[k] = np.concatenate([[tc.thrust_cmd], tc.pqr_cmd, quad.omega_command, quad.omega])
            co.dynamics_step(shim)
            slog[k] = quad.X
        err = np.linalg.norm(slog[::10][:len(traj), 0:3] - traj[:len(slog[::10]), 0:3][:len(slog[::10])], axis=1)
        print(f"closed loop {name}: rows={len(traj)} final_dist={np.linalg.norm(slog[-1, :3] - traj[-1, :3]):.4f} "
              f"mean_err={err.mean():.4f} max_err={err.max():.4f}")
        out[name + "_traj"] = traj
        out[name + "_state_every10"] = slog[9::10]
        out[name + "_cmd_every10"] = clog[9::10]
        out[name + "_state_first200"] = slog[:200]
        out[name + "_cmd_first200"] = clog[:200]
    save("closed_loop.npz", **out)


if __name__ == "__main__":
    all_gens = {"polynom": gen_polynom, "fixed_missions": gen_fixed_missions, "synthetic_missions": gen_synthetic_missions,
                "yaws": gen_yaws, "derivatives": gen_derivatives, "yaws_long": gen_yaws_long,
                "controller_io": gen_controller_io, "open_loop": gen_open_loop, "closed_loop": gen_closed_loop}
    for name in (sys.argv[1:] or list(all_gens)):          # no arguments: everything; else only the named fixtures
        all_gens[name]()
