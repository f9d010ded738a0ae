"""ctypes wrapper of oracle/_build/liboracle.so (oracle/uavac_oracle.c) -- TEST INFRASTRUCTURE,
NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# UAVAC_ORACLE_SO: another build of the same oracle (the sanitizer build of `make -C oracle asan`, tests/test_sanitizers.py)
_SO = os.environ.get("UAVAC_ORACLE_SO") or os.path.join(_HERE, "_build", "liboracle.so")
_P = C.c_void_p


class Vehicle(C.Structure):
    """Field order of `oracle_vehicle` (== uavac_vehicle).  Defaults: lab_course.xml + quad.py:42-73."""
    _fields_ = [(n, C.c_double) for n in ("g", "dt", "dt_outer", "mass")] + [("inertia", C.c_double * 3)] + \
               [(n, C.c_double) for n in (
                   "arm", "kf", "kappa", "min_thrust", "max_thrust", "tau_rise", "tau_fall",
                   "max_ascent", "max_descent", "max_speed_xy", "max_horiz_accel", "max_tilt",
                   "kp_xy", "kd_xy", "kp_z", "kd_z", "ki_z", "kp_roll", "kp_pitch", "kp_yaw",
                   "kp_p", "kp_q", "kp_r")] + [("inner_per_outer", C.c_int32), ("ground", C.c_int32)] + \
               [(n, C.c_double) for n in ("ground_z", "ground_clearance", "ground_timeconst")]

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
            else:
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
        _lib.oracle_bench_threads.restype = C.c_int64
        _lib.oracle_bench_threads.argtypes = [C.POINTER(Vehicle), _P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                              C.c_int, C.c_double, _P]
        _lib.oracle_segment_intersects_cuboid.restype = C.c_int
        _lib.oracle_segment_intersects_cuboid.argtypes = [_P, _P, _P]
        _lib.oracle_rrt_distances.restype = None
        _lib.oracle_rrt_distances.argtypes = [_P, C.c_int, _P, _P]
        _lib.oracle_rrt_edge_lengths.restype = None
        _lib.oracle_rrt_edge_lengths.argtypes = [_P, _P, C.c_int, C.c_int, _P]
        _lib.oracle_rrt_segment_hits.restype = None
        _lib.oracle_rrt_segment_hits.argtypes = [_P, _P, C.c_int, _P, C.c_int, _P]
        _lib.oracle_rrt_steer.restype = None
        _lib.oracle_rrt_steer.argtypes = [_P, _P, C.c_double, _P]
        _lib.oracle_rrt_star.restype = C.c_int
        _lib.oracle_rrt_star.argtypes = [_P, _P, C.c_double, C.c_int, _P, _P, C.c_int] + [_P] * 11
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
    return state, np.zeros(4, dtype=np.int32)


def rollout(traj, state, istate, K: int, V: Vehicle | None = None, log_state=True, log_cmd=True, aabbs=None):
    """K ticks in place on (state[26], istate[4]) -> (state_log (K,13) | None, cmd_log (K,12) | None)."""
    V = V or Vehicle.default()
    traj = np.ascontiguousarray(traj, dtype=np.float64)
    slog = np.empty((K, 13)) if log_state else None
    clog = np.empty((K, 12)) if log_cmd else None
    ab = None if aabbs is None else np.ascontiguousarray(aabbs, dtype=np.float64)
    lib().oracle_rollout(C.byref(V), _p(traj), len(traj), _p(state), _p(istate), K, _p(slog), _p(clog), _p(ab),
                         0 if ab is None else len(ab))
    return slog, clog


def bench_threads(wps, velocity: float, dt: float, ticks: int, n_threads: int, budget_s: float, V: Vehicle | None = None):
    """Whole missions (plan + `ticks` control ticks with the state log) on n_threads POSIX threads for budget_s of wall
    time, entirely inside the C library.  -> (missions completed, wall seconds)."""
    V = V or Vehicle.default()
    wps = np.ascontiguousarray(wps, dtype=np.float64)
    elapsed = C.c_double()
    done = lib().oracle_bench_threads(C.byref(V), _p(wps), wps.shape[0], wps.shape[1] - 1, float(velocity), float(dt),
                                      int(ticks), int(n_threads), float(budget_s), C.addressof(elapsed))
    return int(done), elapsed.value


# ---------------------------------------------------------------------------------------- RRT* (rrt_oracle.c)
RRT_STATUS = {0: "ok", 1: "no path found", 2: "cost increased after rewiring", 3: "KeyError"}


def segment_intersects_cuboid(n1, n2, cuboid) -> bool:
    a, b, c = (np.ascontiguousarray(x, dtype=np.float64) for x in (n1, n2, cuboid))
    return bool(lib().oracle_segment_intersects_cuboid(_p(a), _p(b), _p(c)))


def rrt_distances(nodes, query):
    nodes = np.ascontiguousarray(nodes, dtype=np.float64).reshape(-1, 3)
    q = np.ascontiguousarray(query, dtype=np.float64)
    out = np.empty(len(nodes))
    lib().oracle_rrt_distances(_p(nodes), len(nodes), _p(q), _p(out))
    return out


def rrt_edge_lengths(p0, p1):
    p0 = np.ascontiguousarray(p0, dtype=np.float64).reshape(-1, 3)
    p1 = np.ascontiguousarray(p1, dtype=np.float64)
    out = np.empty(len(p0))
    lib().oracle_rrt_edge_lengths(_p(p0), _p(p1), int(p1.ndim == 1), len(p0), _p(out))
    return out


def rrt_segment_hits(p0, p1, cuboids):
    p0 = np.ascontiguousarray(p0, dtype=np.float64).reshape(-1, 3)
    p1 = np.ascontiguousarray(p1, dtype=np.float64).reshape(-1, 3)
    cub = np.ascontiguousarray(cuboids, dtype=np.float64).reshape(-1, 6)
    hit = np.zeros(len(p0), dtype=np.int32)
    lib().oracle_rrt_segment_hits(_p(p0), _p(p1), len(p0), _p(cub), len(cub), _p(hit))
    return hit.astype(bool)


def rrt_steer(sample, nearest, step: float):
    a, b = np.ascontiguousarray(sample, dtype=np.float64), np.ascontiguousarray(nearest, dtype=np.float64)
    out = np.empty(3)
    lib().oracle_rrt_steer(_p(a), _p(b), float(step), _p(out))
    return out


def rrt_star(start, goal, step: float, samples, cuboids=None):
    """One RRT* run on the node sequence `samples` (max_iter, 3) that _generate_random_node returned.
    -> dict(status, iters, nodes (n,3), canon (n,), parent (n,), best_n, best_parent, best_path (len,3), best_cost)."""
    samples = np.ascontiguousarray(samples, dtype=np.float64).reshape(-1, 3)
    max_iter = len(samples)
    cap = max_iter + 1
    start = np.ascontiguousarray(start, dtype=np.float64)
    goal = np.ascontiguousarray(goal, dtype=np.float64)
    cub = None if cuboids is None else np.ascontiguousarray(cuboids, dtype=np.float64).reshape(-1, 6)
    n_obs = 0 if cub is None else len(cub)
    nodes = np.zeros((cap, 3)); path = np.zeros((cap, 3))
    canon = np.zeros(cap, np.int32); parent = np.zeros(cap, np.int32); bparent = np.zeros(cap, np.int32)
    n = C.c_int(); bn = C.c_int(); blen = C.c_int(); iters = C.c_int(); bcost = C.c_double(); cnt = C.c_int()
    st = lib().oracle_rrt_star(_p(start), _p(goal), float(step), max_iter, _p(samples), _p(cub), n_obs,
                               C.addressof(n), _p(nodes), _p(canon), _p(parent), C.addressof(bn), _p(bparent),
                               C.addressof(blen), _p(path), C.addressof(bcost), C.addressof(iters), C.addressof(cnt))
    return {"status": st, "iters": iters.value, "dynamic_it_counter": cnt.value, "nodes": nodes[:n.value].copy(), "canon": canon[:n.value].copy(),
            "parent": parent[:n.value].copy(), "best_n": bn.value, "best_parent": bparent[:n.value].copy(),
            "best_path": path[:blen.value].copy(), "best_cost": bcost.value}
