"""Single-UAV facade with the surface of the reference's `uav_ac/control/controller.py`
(`CascadedController`).  Each method packs its arguments into one probe record and runs the HIP
stage through the C ABI (`uavac_probe_outer` / `uavac_probe_inner`); `integral_error` is carried on
the host object exactly like the reference.  `quad` is duck-typed like in the reference's tests."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _native as nat
from .._single import ctx, vehicle_from


def _state_of(quad) -> np.ndarray:
    X = getattr(quad, "X", None)
    if X is not None:
        return np.asarray(X, dtype=float)
    X = np.zeros(13)
    X[3] = 1.0
    return X


class CascadedController:
    INTEGRAL_ERROR_LIMIT = 10.0          # applied inside the kernel (reference controller.py:10)

    def __init__(self, g: float, dt: float):
        self.g = g
        self.dt = dt
        self.integral_error = 0

    def reset(self) -> None:
        self.integral_error = 0

    def _outer(self, V, mask, X=None, R=None, target=None, thrust=None, bxy=None, euler=None, q_cmd=None):
        rec = np.zeros((1, 41))
        rec[0, 3] = 1.0
        if X is not None:
            rec[0, 0:13] = X
        if R is not None:
            rec[0, 13:22] = np.asarray(R, dtype=float).reshape(9)
        if target is not None:
            rec[0, 22:33] = target
        rec[0, 33] = self.integral_error
        if thrust is not None:
            rec[0, 34] = thrust
        if bxy is not None:
            rec[0, 35:37] = bxy
        if euler is not None:
            rec[0, 37:40] = euler
        if q_cmd is not None:
            rec[0, 40] = q_cmd
        out = np.empty((1, 21))
        ctx().call("uavac_probe_outer", C.byref(V), nat.np_ptr(rec), 1, mask, nat.np_ptr(out))
        return out[0]

    def altitude(self, quad, des_z, rot_mat, kp_z, kd_z, ki_z):
        """Collective thrust from the altitude PID (reference controller.py:26-56)."""
        V = vehicle_from(quad, g=self.g, dt_outer=self.dt, kp_z=kp_z, kd_z=kd_z, ki_z=ki_z)
        tgt = np.zeros(11)
        tgt[[2, 5, 8]] = des_z
        out = self._outer(V, nat_mask("R"), X=_state_of(quad), R=rot_mat, target=tgt)
        self.integral_error = out[13]
        return out[12]

    def lateral(self, quad, des_x, des_y, thrust_cmd, kp_xy, kd_xy):
        """Commanded tilt from the lateral PD (reference controller.py:58-97)."""
        V = vehicle_from(quad, g=self.g, dt_outer=self.dt, kp_xy=kp_xy, kd_xy=kd_xy)
        tgt = np.zeros(11)
        tgt[[0, 3, 6]] = des_x
        tgt[[1, 4, 7]] = des_y
        return self._outer(V, nat_mask("thrust"), X=_state_of(quad), target=tgt, thrust=thrust_cmd)[14:16].copy()

    def roll_pitch_controller(self, bxy_cmd, rot_mat, kp_roll, kp_pitch):
        """Body roll / pitch rate commands (reference controller.py:132-154)."""
        V = vehicle_from(None, g=self.g, dt_outer=self.dt, kp_roll=kp_roll, kp_pitch=kp_pitch)
        return self._outer(V, nat_mask("R", "bxy"), R=rot_mat, bxy=bxy_cmd)[16:18].copy()

    def yaw_controller(self, quad, psi_des, kp_yaw, q_cmd=0.0):
        """Body yaw rate command (reference controller.py:156-168); quad only needs phi / theta / psi."""
        V = vehicle_from(None, g=self.g, dt_outer=self.dt, kp_yaw=kp_yaw)
        tgt = np.zeros(11)
        tgt[9] = psi_des
        return self._outer(V, nat_mask("euler", "q_cmd"), target=tgt, euler=(quad.phi, quad.theta, quad.psi),
                           q_cmd=q_cmd)[20]

    def reduced_attitude(self, quad, bxy_cmd, psi_des, rot_mat, kp_roll, kp_pitch, kp_yaw):
        """[p_c, q_c, r_c] (reference controller.py:99-113)."""
        V = vehicle_from(quad, g=self.g, dt_outer=self.dt, kp_roll=kp_roll, kp_pitch=kp_pitch, kp_yaw=kp_yaw)
        tgt = np.zeros(11)
        tgt[9] = psi_des
        return self._outer(V, nat_mask("R", "bxy"), X=_state_of(quad), R=rot_mat, target=tgt, bxy=bxy_cmd)[18:21].copy()

    def body_rate_controller(self, quad, pqr_cmd, kp_p, kp_q, kp_r):
        """Moment command (reference controller.py:115-130)."""
        V = vehicle_from(quad, g=self.g, dt_outer=self.dt, kp_p=kp_p, kp_q=kp_q, kp_r=kp_r)
        rec = np.zeros((1, 24))
        rec[0, 0:13] = _state_of(quad)
        rec[0, 13:16] = pqr_cmd
        out = np.empty((1, 15))
        ctx().call("uavac_probe_inner", C.byref(V), nat.np_ptr(rec), 1, 0, nat.np_ptr(out))
        return out[0, 0:3].copy()

    # scalar conveniences kept on the host (reference controller.py:170-191): not part of the GPU path
    @staticmethod
    def wrap_to_pi(angle):
        return (angle + np.pi) % (2 * np.pi) - np.pi

    @staticmethod
    def wrap_to_2pi(angle):
        return angle % (2 * np.pi)

    @staticmethod
    def _pd(kp, kd, error, error_dot, des):
        return kp * error + kd * error_dot + des

    @staticmethod
    def _pid(kp, kd, ki, error, error_dot, i_error, des):
        return kp * error + ki * i_error + kd * error_dot + des


_MASK = {"R": 1, "thrust": 2, "bxy": 4, "euler": 8, "q_cmd": 16}


def nat_mask(*names) -> int:
    m = 0
    for n in names:
        m |= _MASK[n]
    return m
