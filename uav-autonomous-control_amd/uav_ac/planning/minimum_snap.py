"""Single-mission facade with the surface of the reference's `uav_ac/planning/minimum_snap.py`
(`MinimumSnap`), computing on the GPU through the C ABI (B = 1 host-pointer entry points).

Same constructor, `get_trajectory()` -> (N, 11) float64 rows `[x y z vx vy vz ax ay az yaw spline_id]`,
attributes `times`, `coeffs`, `nb_splines`, `waypoints`, `A`, `b`, class constants and static helpers.
Deliberate differences: `get_trajectory()` is idempotent (the reference appends to `times` on a second
call with obstacles=None and raises IndexError, minimum_snap.py:94-95); the obstacle loop is bounded.
Batched planning lives in `uav_ac.fleet.Engine.plan`.
"""
from __future__ import annotations

from typing import List, Set

import numpy as np

from .. import _native as nat
from .._single import ctx




class MinimumSnap:
    START_END_TIME_FACTOR = 1.5                 # reference minimum_snap.py:10 (applied inside the kernel)
    MIN_HORIZONTAL_SPEED_FOR_YAW = 1e-3         # reference minimum_snap.py:11 (applied inside the kernel)
    MAX_REPLAN_ITERATIONS = 64

    def __init__(self, path, obstacles, velocity=1.0, dt=0.01):
        self.coord_obstacles = obstacles
        self.waypoints = path
        self.velocity = velocity
        self.dt = dt
        self.n_coeffs = 8
        self.reset()

    def reset(self):
        self.times = []
        self.spline_id = []
        self.nb_splines = None
        self.positions, self.velocities, self.accelerations, self.yaws = [], [], [], []
        self.full_trajectory = None
        self.A = None
        self.b = None
        self.coeffs = None

    # ------------------------------------------------------------------ GPU path
    def _setup(self):
        """Time allocation (reference :288-325) -- computed by the row-count kernel."""
        wp = nat.as_f64(self.waypoints)
        m = wp.shape[0] - 1
        times = np.empty((1, m))
        seg_rows = np.empty((1, m), dtype=np.int32)
        offs = np.empty(2, dtype=np.int64)
        ctx().call("uavac_minsnap_row_counts", nat.np_ptr(wp), 1, m, float(self.velocity), float(self.dt),
                   nat.np_ptr(times), nat.np_ptr(seg_rows), nat.np_ptr(offs))
        self.nb_splines = m
        self.times = list(times[0])
        self._offs = offs
        self.A, self.b = self._constraint_matrices()

    def _compute_spline_parameters(self, method="lstsq"):
        """Joint minimum-snap QP (reference :138-153).  `method` is accepted for signature parity; the HIP
        solver returns the unique optimum of the same KKT system either way."""
        self._setup()
        wp = nat.as_f64(self.waypoints)
        m = self.nb_splines
        coeffs = np.empty((1, 8 * m, 3))
        ctx().call("uavac_minsnap_solve", nat.np_ptr(wp), 1, m, float(self.velocity), nat.np_ptr(coeffs), None)
        self.coeffs = coeffs[0]

    def _generate_trajectory(self, method="lstsq"):
        self._compute_spline_parameters(method)
        m = self.nb_splines
        traj = np.empty((int(self._offs[1]), nat.TRAJ_COLS))
        times = nat.as_f64(self.times)[None]
        ctx().call("uavac_minsnap_sample", nat.np_ptr(nat.as_f64(self.coeffs[None])), nat.np_ptr(times), 1, m,
                   float(self.dt), nat.np_ptr(self._offs), nat.np_ptr(traj))
        self.positions, self.velocities, self.accelerations = traj[:, 0:3], traj[:, 3:6], traj[:, 6:9]
        self.yaws, self.spline_id = traj[:, 9:10], traj[:, 10:11]
        self.full_trajectory = traj
        return traj

    def get_trajectory(self):
        self._generate_collision_free_trajectory()
        return self.full_trajectory

    def _generate_collision_free_trajectory(self):
        """Obstacle-by-obstacle midpoint insertion (reference :63-95); each re-plan runs on the GPU."""
        if self.coord_obstacles is None:
            self.reset()
            self._generate_trajectory()
            return
        # the whole loop in one call of the C ABI (host buffers; every round runs on the GPU): uavac_minsnap_obstacle_waypoints
        wp = nat.as_f64(self.waypoints)
        m = len(wp) - 1
        cub = nat.as_f64(self.coord_obstacles).reshape(-1, 6)
        so = np.array([0, m], dtype=np.int64)
        cap = nat.MAX_SEGMENTS + 1
        wp_out, so_out = np.empty((cap, 3)), np.empty(2, dtype=np.int64)
        ok = np.zeros(1, dtype=np.int32)
        ctx().call("uavac_minsnap_obstacle_waypoints", nat.np_ptr(wp), nat.np_ptr(so), 1, float(self.velocity), float(self.dt),
                   nat.np_ptr(cub), len(cub), int(self.MAX_REPLAN_ITERATIONS), 0, nat.np_ptr(wp_out), cap, nat.np_ptr(so_out),
                   nat.np_ptr(ok))
        if not ok[0]:            # the reference would loop for ever (a waypoint inside a cuboid, a leg crossing one squarely)
            raise RuntimeError("obstacle correction did not converge (a waypoint inside an obstacle?)")
        final = wp_out[:int(so_out[1]) + 1].copy()
        self.reset()
        self.waypoints = final
        self._generate_trajectory()              # fills times / coeffs / A / b for the final waypoint list

    # ------------------------------------------------------- inspection helpers (host, not on the GPU path)
    def _constraint_matrices(self):
        """A (6m+2, 8m), b (6m+2, 3) in the reference's row order (:171-255) -- for callers that inspect
        them (e.g. the reference's KKT-optimality test); the HIP solver never builds them."""
        wp = nat.as_f64(self.waypoints)
        m, n, T = self.nb_splines, self.n_coeffs, self.times
        A = np.zeros((6 * m + 2, n * m))
        b = np.zeros((6 * m + 2, wp.shape[1]))
        r = 0
        for s in range(m):
            A[r, s * n:(s + 1) * n] = self.polynom(n, 0, 0.0); b[r] = wp[s]; r += 1
        for s in range(m):
            A[r, s * n:(s + 1) * n] = self.polynom(n, 0, T[s]); b[r] = wp[s + 1]; r += 1
        for k in (1, 2, 3):
            A[r, 0:n] = self.polynom(n, k, 0.0); r += 1
        for k in (1, 2, 3):
            A[r, (m - 1) * n:m * n] = self.polynom(n, k, T[-1]); r += 1
        for s in range(1, m):
            for k in (1, 2, 3, 4):
                A[r, (s - 1) * n:s * n] = self.polynom(n, k, T[s - 1])
                A[r, s * n:(s + 1) * n] = -self.polynom(n, k, 0.0)
                r += 1
        return A, b

    def _create_snap_cost_matrix(self):
        """Block-diagonal integral of snap^2 (reference :155-169), inspection helper."""
        n, m = self.n_coeffs, self.nb_splines
        H = np.zeros((n * m, n * m))
        i = np.arange(4, n)
        f = i * (i - 1) * (i - 2) * (i - 3)
        e = i[:, None] + i[None, :] - 7
        for s, T in enumerate(self.times):
            H[s * n + 4:(s + 1) * n, s * n + 4:(s + 1) * n] = f[:, None] * f[None, :] * T ** e / e
        return H

    @staticmethod
    def polynom(n_coeffs, order, t):
        """k-th derivative of the monomial basis at t, ascending powers (reference :257-286)."""
        i = np.arange(n_coeffs, dtype=float)
        fall = np.ones(n_coeffs)
        for j in range(order):
            fall *= (i - j)
        expo = np.maximum(i - order, 0.0)
        return np.where(i >= order, fall * np.power(float(t), expo), 0.0)

    @staticmethod
    def _calculate_yaws(velocities: np.ndarray) -> np.ndarray:
        """Heading hold / unwrap / back-fill (reference :126-136) for a velocity sequence of ANY length, through the
        HIP yaw scan on its own (`uavac_yaw_scan`, the scan the sampler fuses)."""
        vel = nat.as_f64(velocities)
        if vel.ndim != 2 or vel.shape[1] < 2:
            raise IndexError("velocities must have shape (N, >= 2)")        # what velocities[:, :2] would raise upstream
        n = len(vel)
        if n == 0:
            return np.zeros(0)
        v3 = np.zeros((n, 3))
        v3[:, :2] = vel[:, :2]
        yaws = np.empty(n)
        ctx().call("uavac_yaw_scan", nat.np_ptr(v3), n, nat.np_ptr(yaws))
        return yaws

    @staticmethod
    def is_collision_cuboid(x: float, y: float, z: float, cuboid_params: np.ndarray) -> bool:
        """Inclusive AABB membership (reference :327-357)."""
        x_min, x_max, y_min, y_max, z_min, z_max = cuboid_params
        return bool(x_min <= x <= x_max and y_min <= y <= y_max and z_min <= z <= z_max)

    @staticmethod
    def insert_midpoints_at_indexes(points: np.ndarray, indexes: List[int] | Set[int]) -> np.ndarray:
        """Insert the midpoint of (points[i-1], points[i]) before every i in indexes (reference :359-391)."""
        points = np.asarray(points, dtype=float)
        idx = sorted({int(i) for i in indexes})
        if not idx:
            return points.copy()
        mids = (points[[i - 1 for i in idx]] + points[idx]) / 2
        return np.insert(points, idx, mids, axis=0)
