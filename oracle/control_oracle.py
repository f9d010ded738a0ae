"""CPU oracle for the cascaded controller, rotor allocation, motor lag, the
multi-rate trajectory scheduler and the free-body NED dynamics step
--  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Scalar Python, one UAV at a time: use for small cases only (the C
twin oracle/uavac_oracle.c is the one timed as the CPU baseline).

Restates (reference paths relative to the upstream repository root):
  * CascadedController        uav_ac/control/controller.py:26-191
  * Quad state / allocation   uav_ac/quadrotor/quad.py:75-155, 189-213
  * TrajectoryController      uav_ac/main.py:10-61
  * rotor wrench              uav_ac/simulation/mujoco_sim.py:232-251
  * ENU/FLU -> NED/FRD        uav_ac/simulation/mujoco_sim.py:20-45
  * free-body step            MuJoCo `mj_step`, Euler integrator, free joint
                              (third-party; uav_ac/simulation/mujoco_sim.py:147,
                              models/lab_course.xml:3) restated in NED/FRD.

Parity pinning: controller / allocation / scheduler are pinned by golden
vectors produced by importing the reference (tests/golden/make_golden.py) and
by the known answers in the reference's tests/unit/control/test_controller.py
and tests/unit/quadrotor/test_quad.py.  The dynamics step is PARITY UNPINNED
against MuJoCo 3.11.0 (the library is absent from the build container): its
definition is SURVEY.md 8(a) row D2 and it is validated only by invariants
(hover, free fall, closed-loop tracking bounds of
tests/integration/test_mujoco_trajectory_tracking.py:34-36).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

INTEGRAL_ERROR_LIMIT = 10.0     # controller.py:10
TWO_PI = 2.0 * math.pi


@dataclass
class Vehicle:
    """Table V of SURVEY.md 8(a): constants of models/lab_course.xml + gains of quad.py:42-73."""
    g: float = 9.81
    dt: float = 0.001
    mass: float = 0.5
    inertia: tuple = (0.0023, 0.0023, 0.0046)
    arm: float = 0.120208
    kf: float = 1.0
    kappa: float = 0.016
    min_thrust: float = 0.1
    max_thrust: float = 4.5
    tau_rise: float = 0.0125
    tau_fall: float = 0.025
    max_ascent: float = 3.0
    max_descent: float = 2.0
    max_speed_xy: float = 3.0
    max_horiz_accel: float = 12.0
    max_tilt: float = 0.7
    inner_per_outer: int = 10
    # ground plane: 0 = free flight.  BUILD-DEFINED contact law, see dynamics_step
    ground: int = 0
    ground_z: float = 0.0            # NED z of the plane (lab_course.xml:34)
    ground_clearance: float = 0.02   # half height of the body box (lab_course.xml:101)
    ground_timeconst: float = 0.02   # MuJoCo's default solref time constant
    # gains: quad.py:42-73
    kp_xy: float = field(init=False)
    kd_xy: float = field(init=False)
    kp_z: float = field(init=False)
    kd_z: float = field(init=False)
    ki_z: float = 0.1
    kp_roll: float = field(init=False)
    kp_pitch: float = field(init=False)
    kp_yaw: float = field(init=False)
    kp_p: float = field(init=False)
    kp_q: float = field(init=False)
    kp_r: float = field(init=False)

    def __post_init__(self):
        self.kp_xy, self.kd_xy = 1 / 0.25 ** 2, 2 * 0.875 / 0.25
        self.kp_z, self.kd_z = 1 / 0.2 ** 2, 2 * 0.8 / 0.2
        self.kp_roll = 1 / 0.07
        self.kp_pitch = 1 / 0.07
        self.kp_yaw = 1 / 0.25
        self.kp_p = 1 / 0.008
        self.kp_q = 1 / 0.008
        self.kp_r = 1 / 0.09

    @property
    def dt_outer(self) -> float:
        return self.dt * self.inner_per_outer          # main.py:97


def pymod(a: float, b: float) -> float:
    """Python floored modulo for b > 0 (controller.py:173,178 use `%`)."""
    r = math.fmod(a, b)
    if r != 0.0 and (r < 0.0) != (b < 0.0):
        r += b
    return r


def wrap_to_pi(a: float) -> float:
    return pymod(a + math.pi, TWO_PI) - math.pi         # controller.py:170-173


def wrap_to_2pi(a: float) -> float:
    return pymod(a, TWO_PI)                             # controller.py:175-178


def clip(x, lo, hi):
    return lo if x < lo else (hi if x > hi else x)


def quat_to_rot(q) -> np.ndarray:
    """Body->world rotation from a scalar-first quaternion (quad.py:133-155)."""
    q = np.asarray(q, dtype=float)
    q = q / math.sqrt(float(np.sum(q * q)))
    S = np.array([[0.0, -q[3], q[2]], [q[3], 0.0, -q[1]], [-q[2], q[1], 0.0]])
    return np.eye(3) + 2.0 * S @ S + 2.0 * q[0] * S


def euler(q):
    """phi, theta, psi from the stored (un-normalised) quaternion (quad.py:189-213)."""
    phi = math.atan2(2 * (q[0] * q[1] + q[2] * q[3]), 1 - 2 * (q[1] ** 2 + q[2] ** 2))
    st = clip(2 * (q[0] * q[2] - q[3] * q[1]), -1.0, 1.0)
    theta = math.asin(st)
    psi = math.atan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] ** 2 + q[3] ** 2))
    return phi, theta, psi


class UAV:
    """State of one vehicle + its controller/scheduler memory (quad.py:75-86, main.py:21-27)."""

    def __init__(self, V: Vehicle, position=(0.0, 0.0, 0.0), hover: bool = True):
        self.V = V
        self.X = np.zeros(13)
        self.X[0:3] = position
        self.X[3] = 1.0
        w = math.sqrt(V.mass * V.g / (4 * V.kf)) if hover else 0.0
        self.omega = np.full(4, w)
        self.omega_cmd = np.full(4, w) if hover else np.zeros(4)
        self.integral = 0.0
        self.thrust_cmd = 0.0
        self.pqr_cmd = np.zeros(3)
        self.traj_index = 0
        self.inner_step = 0
        self.collided = 0


# ------------------------------------------------------------------ controller
def altitude(u: UAV, des_z, R) -> float:
    """controller.py:26-56 (integral is updated before use, then clamped)."""
    V = u.V
    zd_des = clip(des_z[1], -V.max_ascent, V.max_descent)
    err = des_z[0] - u.X[2]
    err_dot = zd_des - u.X[9]
    u.integral = clip(u.integral + err * V.dt_outer, -INTEGRAL_ERROR_LIMIT, INTEGRAL_ERROR_LIMIT)
    acc = V.kp_z * err + V.ki_z * u.integral + V.kd_z * err_dot + des_z[2] - V.g
    acc = acc / R[2, 2]
    c = -V.mass * acc
    return clip(c, 4 * V.min_thrust, 4 * V.max_thrust)


def lateral(u: UAV, des_x, des_y, thrust_cmd: float):
    """controller.py:58-97."""
    V = u.V
    vdx, vdy = des_x[1], des_y[1]
    vmag = math.sqrt(vdx * vdx + vdy * vdy)
    if vmag > V.max_speed_xy:
        vdx = vdx / vmag * V.max_speed_xy
        vdy = vdy / vmag * V.max_speed_xy
    ax = V.kp_xy * (des_x[0] - u.X[0]) + V.kd_xy * (vdx - u.X[7]) + des_x[2]
    ay = V.kp_xy * (des_y[0] - u.X[1]) + V.kd_xy * (vdy - u.X[8]) + des_y[2]
    amag = math.sqrt(ax * ax + ay * ay)
    if amag > V.max_horiz_accel:
        ax = ax / amag * V.max_horiz_accel
        ay = ay / amag * V.max_horiz_accel
    acc_z = -thrust_cmd / V.mass
    return (clip(ax / acc_z, -V.max_tilt, V.max_tilt), clip(ay / acc_z, -V.max_tilt, V.max_tilt))


def roll_pitch(V: Vehicle, bxy_cmd, R):
    """controller.py:132-154."""
    bdx = V.kp_roll * (bxy_cmd[0] - R[0, 2])
    bdy = V.kp_pitch * (bxy_cmd[1] - R[1, 2])
    p_c = (R[1, 0] / R[2, 2]) * bdx + (-R[0, 0] / R[2, 2]) * bdy
    q_c = (R[1, 1] / R[2, 2]) * bdx + (-R[0, 1] / R[2, 2]) * bdy
    return p_c, q_c


def yaw_rate(V: Vehicle, q, psi_des: float, q_cmd: float) -> float:
    """controller.py:156-168."""
    phi, theta, psi = euler(q)
    psi_des = wrap_to_2pi(psi_des)
    err = wrap_to_pi(psi_des - psi)
    return (V.kp_yaw * err * math.cos(theta) - q_cmd * math.sin(phi)) / math.cos(phi)


def body_rate(u: UAV):
    """controller.py:115-130: I*kp*(cmd - w) + w x (I w)."""
    V = u.V
    I = np.array(V.inertia)
    kp = np.array([V.kp_p, V.kp_q, V.kp_r])
    w = u.X[10:13]
    return I * kp * (u.pqr_cmd - w) + np.cross(w, I * w)


# ------------------------------------------------------------------- actuation
MIX = np.array([[1, 1, 1, 1], [-1, 1, -1, 1], [-1, -1, 1, 1], [1, -1, -1, 1]], dtype=float)  # quad.py:157-166


def allocate(V: Vehicle, thrust_cmd: float, moment) -> np.ndarray:
    """Constrained rotor allocation (quad.py:105-122)."""
    c_bar = clip(thrust_cmd, 4 * V.min_thrust, 4 * V.max_thrust)
    pb = moment[0] / V.arm
    qb = moment[1] / V.arm
    rb = -moment[2] / V.kappa
    mf = MIX @ np.array([pb, qb, rb, 0.0]) / 4
    col = c_bar / 4
    lim = np.ones(4)
    for i in range(4):
        if mf[i] > 0:
            lim[i] = (V.max_thrust - col) / mf[i]
        elif mf[i] < 0:
            lim[i] = (V.min_thrust - col) / mf[i]
    s = clip(float(np.min(lim)), 0.0, 1.0)
    return np.clip(col + s * mf, V.min_thrust, V.max_thrust)


def set_propeller_speed(u: UAV, thrust_cmd: float, moment) -> None:
    """quad.py:88-103: omega_cmd = sqrt(f/kf); first-order lag with rise/fall constants."""
    V = u.V
    f = allocate(V, thrust_cmd, moment)
    u.omega_cmd = np.sqrt(f / V.kf)
    for i in range(4):
        tau = V.tau_rise if u.omega_cmd[i] > u.omega[i] else V.tau_fall
        resp = 1 - math.exp(-V.dt / tau)
        u.omega[i] += resp * (u.omega_cmd[i] - u.omega[i])


# ------------------------------------------------------------------- scheduler
def controller_tick(u: UAV, trajectory: np.ndarray) -> np.ndarray:
    """One TrajectoryController.step() (main.py:37-61).  Returns the moment command."""
    V = u.V
    if u.inner_step % V.inner_per_outer == 0:
        tgt = trajectory[u.traj_index]
        R = quat_to_rot(u.X[3:7])
        thrust = altitude(u, tgt[[2, 5, 8]], R)
        bxy = lateral(u, tgt[[0, 3, 6]], tgt[[1, 4, 7]], thrust)
        u.thrust_cmd = thrust
        p_c, q_c = roll_pitch(V, bxy, R)
        r_c = yaw_rate(V, u.X[3:7], tgt[9], q_c)
        u.pqr_cmd = np.array([p_c, q_c, r_c])
        u.traj_index = min(u.traj_index + 1, len(trajectory) - 1)
    moment = body_rate(u)
    set_propeller_speed(u, u.thrust_cmd, moment)
    u.inner_step += 1
    return moment


# -------------------------------------------------------------------- dynamics
def rotor_wrench(V: Vehicle, omega):
    """Net thrust and body torques in FRD (mujoco_sim.py:232-251 mapped by :20-45;
    sign pattern pinned by tests/unit/quadrotor/test_quad.py:93-94,109)."""
    f = V.kf * np.asarray(omega) ** 2
    T = f[0] + f[1] + f[2] + f[3]
    tx = V.arm * (f[0] + f[3] - f[1] - f[2])
    ty = V.arm * (f[0] + f[1] - f[2] - f[3])
    tz = V.kappa * (-f[0] + f[1] - f[2] + f[3])
    return T, np.array([tx, ty, tz])


TAKEOFF_HEIGHT = 0.1                                                  # mujoco_sim.py:17
GROUND_IN_CONTACT, GROUND_TAKEN_OFF, GROUND_HIT_AFTER_TAKEOFF = 1, 2, 4


def dynamics_step(u: UAV) -> None:
    """Semi-implicit Euler free-body step, NED world / FRD body.

    v' = g e3 - (T/m) R e3 ; w' = I^-1 (tau - w x I w) ; v += dt v' ; w += dt w' ;
    p += dt v_new ; q <- normalise(q (x) exp(dt w_new)).
    """
    V = u.V
    dt = V.dt
    T, tau = rotor_wrench(V, u.omega)
    R = quat_to_rot(u.X[3:7])
    I = np.array(V.inertia)
    w = u.X[10:13]
    acc = np.array([0.0, 0.0, V.g]) - (T / V.mass) * R[:, 2]
    wdot = (tau - np.cross(w, I * w)) / I
    v_new = u.X[7:10] + dt * acc
    if V.ground:
        # BUILD-DEFINED ground contact (MuJoCo's soft-constraint solver cannot run here): while the body's lowest point
        # is below the plane, the vertical velocity update may not exceed the critically damped reference
        # vz + dt (-b vz - k r), b = 2/tc, k = 1/tc^2; the plane only pushes; no friction, no contact torque
        r = u.X[2] - (V.ground_z - V.ground_clearance)
        if r > 0.0:
            tc = V.ground_timeconst
            vz_ref = u.X[9] + dt * -(2.0 / tc * u.X[9] + r / (tc * tc))
            if vz_ref < v_new[2]:
                v_new[2] = vz_ref
    w_new = w + dt * wdot
    p_new = u.X[0:3] + dt * v_new
    q = u.X[3:7]
    wn = math.sqrt(float(np.sum(w_new * w_new)))
    if wn > 0.0:
        half = 0.5 * wn * dt
        s = math.sin(half) / wn
        dq = np.array([math.cos(half), s * w_new[0], s * w_new[1], s * w_new[2]])
        q = np.array([
            q[0] * dq[0] - q[1] * dq[1] - q[2] * dq[2] - q[3] * dq[3],
            q[0] * dq[1] + q[1] * dq[0] + q[2] * dq[3] - q[3] * dq[2],
            q[0] * dq[2] - q[1] * dq[3] + q[2] * dq[0] + q[3] * dq[1],
            q[0] * dq[3] + q[1] * dq[2] - q[2] * dq[1] + q[3] * dq[0],
        ])
    q = q / math.sqrt(float(np.sum(q * q)))
    u.X[0:3] = p_new
    u.X[3:7] = q
    u.X[7:10] = v_new
    u.X[10:13] = w_new
    if V.ground:        # bookkeeping of MujocoSimulation._record_collisions (mujoco_sim.py:220-230)
        bits = getattr(u, "ground_bits", 0)
        if V.ground_z - p_new[2] >= TAKEOFF_HEIGHT:
            bits |= GROUND_TAKEN_OFF
        touching = p_new[2] - (V.ground_z - V.ground_clearance) > 0.0
        bits = (bits | GROUND_IN_CONTACT) if touching else (bits & ~GROUND_IN_CONTACT)
        if touching and bits & GROUND_TAKEN_OFF:
            bits |= GROUND_HIT_AFTER_TAKEOFF
        u.ground_bits = bits


def in_any_aabb(p, aabbs) -> bool:
    """Inclusive AABB membership (minimum_snap.py:327-357 semantics) for config 5's flag."""
    for c in aabbs:
        if c[0] <= p[0] <= c[1] and c[2] <= p[1] <= c[3] and c[4] <= p[2] <= c[5]:
            return True
    return False


def rollout(u: UAV, trajectory: np.ndarray, K: int, aabbs=None):
    """K ticks of controller + dynamics.  Returns (state_log (K,13), cmd_log (K,12)).

    cmd_log columns: thrust_cmd, pqr_cmd(3), omega_cmd(4), omega(4) after the
    controller tick (before the dynamics step).  The collision flag is sticky
    and evaluated on the position after every dynamics step.
    """
    slog = np.empty((K, 13))
    clog = np.empty((K, 12))
    for k in range(K):
        controller_tick(u, trajectory)
        clog[k, 0] = u.thrust_cmd
        clog[k, 1:4] = u.pqr_cmd
        clog[k, 4:8] = u.omega_cmd
        clog[k, 8:12] = u.omega
        dynamics_step(u)
        if aabbs is not None and in_any_aabb(u.X[0:3], aabbs):
            u.collided = 1
        slog[k] = u.X
    return slog, clog


def mujoco_to_ned_state(position, quaternion, velocity) -> np.ndarray:
    """ENU/FLU -> NED/FRD state conversion (mujoco_sim.py:20-45)."""
    position = np.asarray(position, dtype=float)
    quaternion = np.asarray(quaternion, dtype=float)
    velocity = np.asarray(velocity, dtype=float)
    if position.shape != (3,) or quaternion.shape != (4,) or velocity.shape != (6,):
        raise ValueError("bad shape")
    n = math.sqrt(float(np.sum(quaternion ** 2)))
    if n == 0:
        raise ValueError("MuJoCo quaternion cannot be zero")
    flip = np.array([1.0, -1.0, -1.0])
    out = np.empty(13)
    out[0:3] = flip * position
    out[3:7] = quaternion / n * np.array([1.0, 1.0, -1.0, -1.0])
    out[7:10] = flip * velocity[:3]
    out[10:13] = flip * velocity[3:]
    return out
