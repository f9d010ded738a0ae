"""The reference's own unit tests, replayed against the drop-in facade classes (same module paths,
same signatures; upstream tests/unit/{planning,control,quadrotor}/ and tests/unit/test_main.py), plus
golden traces produced by the reference.  Every numeric call lands in a HIP kernel through the C ABI."""
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import col_err, load_golden

pytestmark = pytest.mark.gpu

G, DT = 9.81, 0.01


@pytest.fixture
def quad():
    from uav_ac.quadrotor.quad import Quad
    return Quad.laboratory()


@pytest.fixture
def controller():
    from uav_ac.control.controller import CascadedController
    return CascadedController(g=G, dt=DT)


# ------------------------------------------------------------------ planning (upstream test_minimum_snap.py)
def test_minimum_snap_reference_unit_tests():
    from uav_ac.planning.minimum_snap import MinimumSnap
    pts = np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2], [3, 3, 3]])
    exp = np.array([[0, 0, 0], [.5, .5, .5], [1, 1, 1], [2, 2, 2], [2.5, 2.5, 2.5], [3, 3, 3]])
    assert MinimumSnap.insert_midpoints_at_indexes(pts, [1, 3]) == pytest.approx(exp)
    assert MinimumSnap.insert_midpoints_at_indexes(pts, []) == pytest.approx(pts)
    assert MinimumSnap.polynom(8, 3, 0) == pytest.approx([0, 0, 0, 6, 0, 0, 0, 0])
    assert MinimumSnap.polynom(8, 2, 3) == pytest.approx([0., 0., 2., 18., 108., 540., 2430., 10206.])
    assert MinimumSnap.polynom(8, 6, 3) == pytest.approx([0., 0., 0., 0., 0., 0., 720., 15120.])
    # passes through all waypoints (:64-76)
    wp = np.array([[0., 0., 1.], [3., 0., 1.], [3., 3., 1.]])
    traj = MinimumSnap(wp, None, velocity=2.0, dt=0.01).get_trajectory()
    for w in wp:
        assert np.linalg.norm(traj[:, :3] - w, axis=1).min() == pytest.approx(0.0, abs=0.05)
    # velocity continuity (:139-151)
    assert np.linalg.norm(np.diff(traj[:, 3:6], axis=0), axis=1).max() < 0.5
    # layout (:79-90) and heading along +y (:93-102)
    t1 = MinimumSnap(np.array([[0., 0., 1.], [3., 0., 1.]]), None, velocity=2.0, dt=0.01).get_trajectory()
    assert t1.shape[1] == 11 and t1[:, 9] == pytest.approx(np.zeros(len(t1))) and np.all(t1[:, 10] == 0.0)
    t2 = MinimumSnap(np.array([[0., 0., -1.], [0., 3., -1.]]), None, velocity=2.0, dt=0.01).get_trajectory()
    assert t2[:, 9] == pytest.approx(np.pi / 2)
    # yaw scan (:105-136)
    assert MinimumSnap._calculate_yaws(np.array([[0., 0., -1.], [0., 2., 0.], [0., 0., 1.]])) == pytest.approx(np.full(3, np.pi / 2))
    y = MinimumSnap._calculate_yaws(np.array([[-1., 0.01, 0.], [-1., -0.01, 0.]]))
    assert abs(y[1] - y[0]) < 0.1 and y[1] > np.pi
    assert MinimumSnap._calculate_yaws(np.array([[0., 0., -1.], [0., 0., 0.], [0., 0., 1.]])) == pytest.approx(np.zeros(3))
    # KKT optimality: projected snap-cost gradient on null(A) (:154-168)
    ms = MinimumSnap(np.array([[0., 0., -1.], [2., 1., -1.], [4., -1., -2.], [6., 0., -2.]]), None, velocity=2.0, dt=0.01)
    ms._compute_spline_parameters("lstsq")
    H = ms._create_snap_cost_matrix()
    _, sv, vt = np.linalg.svd(ms.A, full_matrices=True)
    null = vt[np.sum(sv > 1e-10):].T
    assert np.linalg.norm(null.T @ H @ ms.coeffs) < 1e-6
    assert np.abs(ms.A @ ms.coeffs - ms.b).max() < 1e-9            # and the constraints hold
    # time factor (:203-216)
    ms = MinimumSnap(np.array([[0., 0., -1.], [2., 0., -1.], [4., 0., -1.], [6., 0., -1.]]), None, velocity=2.0, dt=0.01)
    ms._setup()
    assert ms.times == pytest.approx([1.5, 1.0, 1.5])
    # AABB truth table (:186-200)
    cub = np.array([1, 5, 2, 6, 3, 7])
    assert [MinimumSnap.is_collision_cuboid(*p, cub) for p in ((2, 3, 4), (0, 0, 0), (1, 6, 3), (5, 2, 8))] == [True, False, True, False]
    # idempotent (deliberate difference from the reference, which raises on the second call)
    ms = MinimumSnap(wp, None, velocity=2.0, dt=0.01)
    assert np.array_equal(ms.get_trajectory(), ms.get_trajectory())


def test_obstacle_replan_and_lab_mission_match_reference_golden():
    """N1: midpoint re-plan loop, host-side orchestration over GPU plans."""
    from uav_ac.main import _generate_mission_trajectory, _trajectory_after_takeoff
    from uav_ac.planning.minimum_snap import MinimumSnap
    g = load_golden("fixed_missions.npz")
    raw = MinimumSnap(g["obs_case_wp"], None, velocity=2.0, dt=0.01).get_trajectory()
    assert any(MinimumSnap.is_collision_cuboid(*p, g["obs_case_aabb"][0]) for p in raw[:, :3])
    ms = MinimumSnap(g["obs_case_wp"], g["obs_case_aabb"], velocity=2.0, dt=0.01)
    traj = ms.get_trajectory()
    assert not any(MinimumSnap.is_collision_cuboid(*p, g["obs_case_aabb"][0]) for p in traj[:, :3])
    assert np.array_equal(ms.waypoints, g["obs_case_final_wp"])
    assert col_err(traj, g["obs_case_traj"]) < 1e-5
    lab = _generate_mission_trajectory(g["lab_wp"], g["lab_aabbs"], 3.0, 0.01)
    assert lab.shape == g["lab_traj_obs"].shape and col_err(lab, g["lab_traj_obs"]) < 1e-5
    free = _generate_mission_trajectory(g["lab_wp"], None, 3.0, 0.01)
    assert free.shape == (1076, 11) and col_err(free, g["lab_traj_free"]) < 1e-5
    # upstream tests/unit/test_main.py:37-74
    assert np.all(lab[:, 2] <= 0)
    takeoff_rows = len(MinimumSnap(g["lab_wp"][:2], None, 3.0, 0.01).get_trajectory())
    assert np.allclose(lab[:takeoff_rows, 0:2], g["lab_wp"][0, 0:2])
    vis = _trajectory_after_takeoff(lab, g["lab_wp"][1])
    assert np.linalg.norm(vis[0, :3] - g["lab_wp"][1]) < 0.05 and len(vis) < len(lab)


# ------------------------------------------------------------------ controller (upstream test_controller.py)
def test_controller_reference_unit_tests(controller, quad):
    from uav_ac.control.controller import CascadedController
    assert CascadedController.wrap_to_pi(-3 * np.pi / 2) == pytest.approx(np.pi / 2)
    assert CascadedController.wrap_to_2pi(-0.1) == pytest.approx(2 * np.pi - 0.1)
    assert CascadedController._pid(2.0, 3.0, 0.5, 1.5, 0.5, 4.0, 0.25) == pytest.approx(2.0 * 1.5 + 3.0 * 0.5 + 0.5 * 4.0 + 0.25)
    eye = np.eye(3)
    assert controller.altitude(quad, np.array([quad.z, 0.0, 0.0]), eye, quad.kp_z, quad.kd_z, quad.ki_z) == pytest.approx(quad.m * G)
    for big, lim in ((100.0, quad.max_descent_rate), (-100.0, -quad.max_ascent_rate)):
        a = CascadedController(G, DT).altitude(quad, np.array([quad.z, big, 0.0]), eye, quad.kp_z, quad.kd_z, quad.ki_z)
        b = CascadedController(G, DT).altitude(quad, np.array([quad.z, lim, 0.0]), eye, quad.kp_z, quad.kd_z, quad.ki_z)
        assert a == pytest.approx(b)
    t = controller.altitude(quad, np.array([quad.z - 100.0, 0.0, 0.0]), eye, quad.kp_z, quad.kd_z, quad.ki_z)
    assert quad.min_thrust * 4 <= t <= quad.max_thrust * 4
    c2 = CascadedController(G, DT)
    for _ in range(1200):
        c2.altitude(quad, np.array([quad.z + 5.0, 0.0, 0.0]), eye, quad.kp_z, quad.kd_z, quad.ki_z)
    assert abs(c2.integral_error) <= CascadedController.INTEGRAL_ERROR_LIMIT
    assert np.allclose(c2.integral_error, load_golden("controller_io.npz")["integral_sequence"][1199], atol=1e-12)
    b = controller.lateral(quad, np.array([100.0, 0, 0]), np.array([-100.0, 0, 0]), quad.m * G, quad.kp_xy, quad.kd_xy)
    assert np.all(np.abs(b) <= quad.max_tilt_angle)
    assert controller.body_rate_controller(quad, np.array([1.0, 0, 0]), quad.kp_p, quad.kp_q, quad.kp_r) == pytest.approx([quad.i_x * quad.kp_p, 0, 0])
    quad.X[10:13] = [1.0, 2.0, 3.0]
    I = np.array([quad.i_x, quad.i_y, quad.i_z])
    got = controller.body_rate_controller(quad, quad.body_angular_velocity.copy(), quad.kp_p, quad.kp_q, quad.kp_r)
    assert got == pytest.approx(np.cross(quad.body_angular_velocity, I * quad.body_angular_velocity))
    stub = SimpleNamespace(phi=0.0, theta=0.0, psi=0.1)
    assert controller.yaw_controller(stub, -0.1, 2.0) == pytest.approx(2.0 * -0.2)
    stub = SimpleNamespace(phi=0.3, theta=-0.2, psi=0.1)
    exp = (2.0 * (0.4 - 0.1) * np.cos(-0.2) - 0.5 * np.sin(0.3)) / np.cos(0.3)
    assert controller.yaw_controller(stub, 0.4, 2.0, 0.5) == pytest.approx(exp)


# ------------------------------------------------------------------ quad (upstream test_quad.py)
def test_quad_reference_unit_tests(quad):
    from uav_ac.quadrotor.quad import Quad
    assert Quad.quat_to_rot(np.array([1.0, 0, 0, 0])) == pytest.approx(np.eye(3))
    h = np.pi / 4
    assert Quad.quat_to_rot(np.array([np.cos(h), 0, 0, np.sin(h)])) @ [1, 0, 0] == pytest.approx([0, 1, 0], abs=1e-12)
    R = Quad.quat_to_rot(np.array([0.4, -0.3, 0.5, 0.2]))
    assert R.T @ R == pytest.approx(np.eye(3)) and np.linalg.det(R) == pytest.approx(1.0)
    quad.X[3:7] = [np.cos(0.15), np.sin(0.15), 0, 0]
    assert quad.euler_angles == pytest.approx([0.3, 0, 0])
    quad.X[3:7] = [np.cos(0.6), 0, 0, np.sin(0.6)]
    assert quad.euler_angles == pytest.approx([0, 0, 1.2]) and quad.psi == pytest.approx(1.2)
    quad.set_propeller_speed(2.0, np.zeros(3))
    assert np.sum(quad.kf * quad.omega_command ** 2) == pytest.approx(2.0)
    quad.set_propeller_speed(4.0, np.array([0.2, 0, 0]))
    f = quad.kf * quad.omega_command ** 2
    assert quad.l * (f[0] + f[3] - f[1] - f[2]) == pytest.approx(0.2)
    quad.set_propeller_speed(4.0, np.array([0, 0, 0.5]))
    f = quad.kf * quad.omega_command ** 2
    assert np.all(f >= quad.min_thrust) and np.all(f <= quad.max_thrust) and f.sum() == pytest.approx(4.0)
    q2 = Quad.laboratory()
    q2.set_propeller_speed(4.0, np.zeros(3))
    assert q2.omega_command == pytest.approx(np.ones(4))
    assert q2.omega == pytest.approx(np.full(4, 1 - np.exp(-q2.dt / q2.motor_rise_time_constant)))
    assert (quad.kp_xy, quad.kd_xy, quad.kp_p) == pytest.approx((1 / 0.25 ** 2, 2 * 0.875 / 0.25, 1 / 0.008))


# ------------------------------------------------------------------ main (upstream test_main.py + golden traces)
def test_trajectory_controller_open_loop_trace_matches_reference_golden(quad):
    """frozen state, 50 ticks -- pins the multi-rate scheduling of main.py:37-61."""
    from uav_ac.control.controller import CascadedController
    from uav_ac.main import TrajectoryController
    g = load_golden("open_loop.npz")
    quad.X = g["X0"].copy()
    quad.omega = np.full(4, np.sqrt(0.5 * 9.81 / 4))
    tc = TrajectoryController(CascadedController(9.81, 0.01), quad, g["traj"], 10)
    log = np.empty_like(g["log"])
    for k in range(len(log)):
        tc.step()
        log[k] = np.concatenate([[tc.thrust_cmd], tc.pqr_cmd, quad.omega_command, quad.omega,
                                 [tc.trajectory_index, tc.controller.integral_error]])
    assert np.array_equal(log[:, 12], g["log"][:, 12])
    assert col_err(log, g["log"]) < 1e-11
    tc.reset()                                                     # upstream test_main.py:13-34
    assert (tc.trajectory_index, tc.inner_step, tc.thrust_cmd, tc.controller.integral_error) == (0, 0, 0.0, 0)
    assert np.all(tc.pqr_cmd == 0)
    # round-2 ADVICE: the reference reads self.trajectory[idx] live (main.py:48) -- an edit IN PLACE of a row that has not
    # been consumed yet must reach the GPU: same trace as a controller built on the edited rows from the start
    edited = g["traj"].copy()
    edited[2:, 2] -= 0.7                                           # other altitude targets from the third row on
    def trace(tc_, q_, n=30):
        out = []
        for _ in range(n):
            tc_.step()
            out.append(np.concatenate([[tc_.thrust_cmd], tc_.pqr_cmd, q_.omega_command, [tc_.trajectory_index]]))
        return np.array(out)
    import copy
    q1, q2 = copy.deepcopy(quad), copy.deepcopy(quad)
    for q_ in (q1, q2):
        q_.X = g["X0"].copy()
        q_.omega = np.full(4, np.sqrt(0.5 * 9.81 / 4))
        q_.omega_command = np.zeros(4)
    live = g["traj"].copy()
    tc1 = TrajectoryController(CascadedController(9.81, 0.01), q1, live, 10)
    head = trace(tc1, q1, 15)                                      # rows 0 and 1 consumed
    live[2:, 2] -= 0.7                                             # in place: same object, same shape, same pointer
    tail = trace(tc1, q1, 15)
    tc2 = TrajectoryController(CascadedController(9.81, 0.01), q2, edited, 10)
    assert np.array_equal(np.vstack([head, tail]), trace(tc2, q2, 30))


def test_facade_closed_loop_matches_reference_golden(quad):
    """tc.step() + FreeFlightSimulation.step(), 600 ticks of the lab course, against the golden trace."""
    from uav_ac.control.controller import CascadedController
    from uav_ac.main import FreeFlightSimulation, TrajectoryController
    g = load_golden("closed_loop.npz")
    traj = g["lab_v2_traj"]
    quad.X[0:3] = traj[0, 0:3]
    w = np.sqrt(quad.m * quad.g / (4 * quad.kf))
    quad.omega = np.full(4, w)
    quad.omega_command = np.full(4, w)
    tc = TrajectoryController(CascadedController(quad.g, quad.dt * 10), quad, traj, 10)
    sim = FreeFlightSimulation(quad)
    K = 600
    slog = np.empty((K, 13)); clog = np.empty((K, 12))
    for k in range(K):
        tc.step()
        clog[k] = np.concatenate([[tc.thrust_cmd], tc.pqr_cmd, quad.omega_command, quad.omega])
        slog[k] = sim.step()
    assert col_err(slog[:200], g["lab_v2_state_first200"]) < 1e-9
    assert col_err(clog[:200], g["lab_v2_cmd_first200"]) < 1e-9
    assert col_err(slog[9::10], g["lab_v2_state_every10"][:K // 10]) < 1e-5
    assert col_err(clog[9::10], g["lab_v2_cmd_every10"][:K // 10]) < 1e-5


def test_lab_mission_from_scene_meets_reference_integration_bounds():
    """Upstream tests/integration/test_mujoco_trajectory_tracking.py:11-36 (v = 2 m/s, dt = 0.01, F = 10) in
    free flight: scene reader -> obstacle-aware plan -> one fused rollout with the AABB flag."""
    import os
    from conftest import GOLDEN
    from uav_ac.main import fly_mission
    out = fly_mission(os.path.join(GOLDEN, "lab_scene_min.xml"), velocity=2.0, frequency=10)
    g = load_golden("closed_loop.npz")
    assert out["trajectory"].shape == g["lab_v2_traj"].shape
    assert col_err(out["trajectory"], g["lab_v2_traj"]) < 1e-5
    assert out["distance_to_goal"] < 0.5 and out["goal_reached"]
    assert out["mean_tracking_error"] < 0.5
    assert out["collision_detected"] is False


def test_end_to_end_example_runs():
    """examples/plan_and_fly.py: RRT* -> thinning -> obstacle-aware minimum snap -> fused flight, 64 vehicles."""
    import importlib.util
    import os
    from conftest import REPO
    spec = importlib.util.spec_from_file_location("plan_and_fly", os.path.join(REPO, "examples", "plan_and_fly.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(64)
    assert out["found"] >= 60
    assert out["arrived"] >= 0.9 * out["flown"]
    assert out["collided"] <= 0.05 * out["flown"]
