"""Single-vehicle facade with the surface of the reference's `uav_ac/quadrotor/quad.py` (`Quad`).
State lives in host NumPy arrays exactly like the reference (`X`, `omega`, `omega_command` are
mutable); every computation (rotation, Euler angles, allocation, motor lag) runs in the HIP probe
kernels through the C ABI.  The fleet-scale path is `uav_ac.fleet`."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _native as nat
from .._single import ctx, vehicle_from


class Quad:
    def __init__(self, g, dt, mass, inertia, arm_length, force_coefficient, drag_to_thrust, thrust_limits,
                 motor_time_constants, flight_limits):
        self.g = g
        self.dt = dt
        self.l = float(arm_length)
        self.m = float(mass)
        self.kf = float(force_coefficient)
        self.kappa = float(drag_to_thrust)
        self.i_x, self.i_y, self.i_z = np.asarray(inertia, dtype=float)
        self.min_thrust, self.max_thrust = np.asarray(thrust_limits, dtype=float)
        (self.max_ascent_rate, self.max_descent_rate, self.max_speed_xy, self.max_horiz_accel,
         self.max_tilt_angle) = np.asarray(flight_limits, dtype=float)
        # response parameters and derived gains: reference quad.py:42-73
        self.tau_xy, self.zeta_xy = 0.25, 0.875
        self.tau_altitude, self.zeta_altitude = 0.2, 0.8
        self.tau_roll = self.tau_pitch = 0.07
        self.tau_yaw = 0.25
        self.tau_p = self.tau_q = 0.008
        self.tau_r = 0.09
        self.kp_xy, self.kd_xy = Quad.second_order_gains(self.tau_xy, self.zeta_xy)
        self.kp_z, self.kd_z = Quad.second_order_gains(self.tau_altitude, self.zeta_altitude)
        self.ki_z = 0.1
        self.kp_roll, self.kp_pitch, self.kp_yaw = 1 / self.tau_roll, 1 / self.tau_pitch, 1 / self.tau_yaw
        self.kp_p, self.kp_q, self.kp_r = 1 / self.tau_p, 1 / self.tau_q, 1 / self.tau_r
        self.X = np.zeros(13)
        self.X[3] = 1.0
        self.motor_rise_time_constant, self.motor_fall_time_constant = np.asarray(motor_time_constants, dtype=float)
        self.omega = np.zeros(4)
        self.omega_command = np.zeros(4)

    @classmethod
    def laboratory(cls) -> "Quad":
        """The vehicle of the reference's models/lab_course.xml (:3,9-13,100,116)."""
        return cls(g=9.81, dt=0.001, mass=0.5, inertia=np.array([0.0023, 0.0023, 0.0046]), arm_length=0.120208,
                   force_coefficient=1.0, drag_to_thrust=0.016, thrust_limits=np.array([0.1, 4.5]),
                   motor_time_constants=np.array([0.0125, 0.025]), flight_limits=np.array([3.0, 2.0, 3.0, 12.0, 0.7]))

    # ---------------------------------------------------------------- GPU-backed computations
    def _vehicle(self) -> nat.Vehicle:
        return vehicle_from(self)

    def _probe_outer(self, X) -> np.ndarray:
        rec = np.zeros((1, 41))
        rec[0, 0:13] = X
        out = np.empty((1, 21))
        ctx().call("uavac_probe_outer", C.byref(self._vehicle()), nat.np_ptr(rec), 1, 0, nat.np_ptr(out))
        return out[0]

    def set_propeller_speed(self, thrust_cmd: float, moment_cmd: np.ndarray):
        """Allocation + first-order motor response (reference quad.py:88-122)."""
        rec = np.zeros((1, 24))
        rec[0, 0:13] = self.X
        rec[0, 16] = thrust_cmd
        rec[0, 17:21] = self.omega
        rec[0, 21:24] = moment_cmd
        out = np.empty((1, 15))
        ctx().call("uavac_probe_inner", C.byref(self._vehicle()), nat.np_ptr(rec), 1, 1, nat.np_ptr(out))
        self.omega_command = out[0, 7:11].copy()
        self.omega = out[0, 11:15].copy()

    def _allocate_rotor_forces(self, thrust_cmd: float, moment_cmd: np.ndarray) -> np.ndarray:
        rec = np.zeros((1, 24))
        rec[0, 3] = 1.0
        rec[0, 16] = thrust_cmd
        rec[0, 21:24] = moment_cmd
        out = np.empty((1, 15))
        ctx().call("uavac_probe_inner", C.byref(self._vehicle()), nat.np_ptr(rec), 1, 1, nat.np_ptr(out))
        return out[0, 3:7].copy()

    def R(self):
        return self._probe_outer(self.X)[0:9].reshape(3, 3).copy()

    @staticmethod
    def quat_to_rot(q: np.ndarray) -> np.ndarray:
        X = np.zeros(13)
        X[3:7] = q
        rec = np.zeros((1, 41))
        rec[0, 0:13] = X
        out = np.empty((1, 21))
        V = nat.Vehicle.default()
        ctx().call("uavac_probe_outer", C.byref(V), nat.np_ptr(rec), 1, 0, nat.np_ptr(out))
        return out[0, 0:9].reshape(3, 3).copy()

    @property
    def euler_angles(self):
        return self._probe_outer(self.X)[9:12].copy()

    @property
    def phi(self):
        return self.euler_angles[0]

    @property
    def theta(self):
        return self.euler_angles[1]

    @property
    def psi(self):
        return self.euler_angles[2]

    # ---------------------------------------------------------------- plain accessors (reference quad.py:168-250)
    @staticmethod
    def second_order_gains(time_constant: float, damping_ratio: float):
        return 1 / time_constant ** 2, 2 * damping_ratio / time_constant

    @staticmethod
    def propeller_coeffs() -> np.ndarray:
        """Mixing matrix rows FL, FR, RR, RL (reference quad.py:157-166; its row comments swap the last two)."""
        return np.array([[1, 1, 1, 1], [-1, 1, -1, 1], [-1, -1, 1, 1], [1, -1, -1, 1]])

    x = property(lambda s: s.X[0])
    y = property(lambda s: s.X[1])
    z = property(lambda s: s.X[2])
    position = property(lambda s: np.array(s.X[0:3]))
    quaternion = property(lambda s: s.X[3:7])
    x_vel = property(lambda s: s.X[7])
    y_vel = property(lambda s: s.X[8])
    z_vel = property(lambda s: s.X[9])
    velocity = property(lambda s: np.array(s.X[7:10]))
    p = property(lambda s: s.X[10])
    q = property(lambda s: s.X[11])
    r = property(lambda s: s.X[12])
    body_angular_velocity = property(lambda s: np.array(s.X[10:13]))
