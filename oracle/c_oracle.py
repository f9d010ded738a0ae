"""ctypes wrapper of oracle/_build/liboracle.so (oracle/uavac_oracle.c) -- TEST INFRASTRUCTURE,
NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_P = C.c_void_p


class Vehicle(C.Structure):
    """Field order of `oracle_vehicle` (== uavac_vehicle).  Defaults: lab_course.xml + quad.py:42-73."""
    _fields_ = [(n, C.c_double) for n in ("g", "dt", "dt_outer", "mass")] + [("inertia", C.c_double * 3)] + \
               [(n, C.c_double) for n in (
                   "arm", "kf", "kappa", "min_thrust", "max_thrust", "tau_rise", "tau_fall",
                   "max_ascent", "max_descent", "max_speed_xy", "max_horiz_accel", "max_tilt",
                   "kp_xy", "kd_xy", "kp_z", "kd_z", "ki_z", "kp_roll", "kp_pitch", "kp_yaw",
                   "kp_p", "kp_q", "kp_r")] + [("inner_per_outer", C.c_int32), ("reserved", C.c_int32)]

    @classmethod
    def default(cls):
        from .control_oracle import Vehicle as PyVehicle
        p = PyVehicle()
        v = cls()
        for n, _ in cls._fields_:
            if n == "inertia":
                v.inertia[:] = p.inertia
            elif n == "dt_outer":
                v.dt_outer = p.dt_outer
            elif n != "reserved":
                setattr(v, n, getattr(p, n))
        return v


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _lib = C.CDLL(_SO)
        _lib.oracle_solve.restype = C.c_int
        _lib.oracle_solve.argtypes = [_P, C.c_int, C.c_double, _P, _P]
        _lib.oracle_row_count.restype = C.c_int64
        _lib.oracle_row_count.argtypes = [_P, C.c_int, C.c_double]
        _lib.oracle_sample.restype = C.c_int64
        _lib.oracle_sample.argtypes = [_P, _P, C.c_int, C.c_double, _P]
        _lib.oracle_rollout.restype = None
        _lib.oracle_rollout.argtypes = [C.POINTER(Vehicle), _P, C.c_int64, _P, _P, C.c_int, _P, _P, _P, C.c_int]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


def plan(waypoints, velocity: float, dt: float):
    """One un-obstructed mission -> (traj (N,11), coeffs (8m,3), times (m,))."""
    wp = np.ascontiguousarray(waypoints, dtype=np.float64)
    m = wp.shape[0] - 1
    coeffs = np.empty((8 * m, 3))
    times = np.empty(m)
    rc = lib().oracle_solve(_p(wp), m, velocity, _p(coeffs), _p(times))
    if rc != 0:
        raise RuntimeError(f"oracle_solve failed ({rc})")
    n = lib().oracle_row_count(_p(times), m, dt)
    traj = np.empty((n, 11))
    assert lib().oracle_sample(_p(coeffs), _p(times), m, dt, _p(traj)) == n
    return traj, coeffs, times


def initial_state(position, V: Vehicle | None = None, hover: bool = True):
    V = V or Vehicle.default()
    state = np.zeros(26)
    state[0:3] = position
    state[3] = 1.0
    if hover:
        state[13:21] = np.sqrt(V.mass * V.g / (4 * V.kf))
    return state, np.zeros(3, dtype=np.int32)


def rollout(traj, state, istate, K: int, V: Vehicle | None = None, log_state=True, log_cmd=True, aabbs=None):
    """K ticks in place on (state[26], istate[3]) -> (state_log (K,13) | None, cmd_log (K,12) | None)."""
    V = V or Vehicle.default()
    traj = np.ascontiguousarray(traj, dtype=np.float64)
    slog = np.empty((K, 13)) if log_state else None
    clog = np.empty((K, 12)) if log_cmd else None
    ab = None if aabbs is None else np.ascontiguousarray(aabbs, dtype=np.float64)
    lib().oracle_rollout(C.byref(V), _p(traj), len(traj), _p(state), _p(istate), K, _p(slog), _p(clog), _p(ab),
                         0 if ab is None else len(ab))
    return slog, clog
