"""MuJoCo-free stand-in for the reference's `uav_ac/simulation/mujoco_sim.py`.

`MujocoSimulation(model_path)` reads the MJCF scene WITHOUT MuJoCo (plain XML: vehicle constants from
`<custom><numeric>`, body mass / inertia, rotor sites and spin signs, `waypoint_NN` / `goal` sites,
`obstacle_*` boxes) into the same attributes the reference adapter exposes -- `quad`,
`mission_waypoints`, `obstacles`, `start_position`, `goal_position`, `space_limits` -- and `step()`
advances the vehicle on the GPU with the build-defined free-body step (`uavac_dynamics_step`): rotor
wrench + semi-implicit Euler, no viewer.  The vehicle starts where the scene puts it -- ON THE GROUND with
stopped rotors, like the reference (lab_course.xml:98, mujoco_sim.py:190-199) -- when the scene has a `ground`
plane: the plane and the half height of the `body` box go into `uavac_vehicle.ground*`, the contact is the
build-defined one of control_law.h (critically damped normal push, no friction, no torque; NOT MuJoCo's solver:
contact forces are not comparable), and the take-off bookkeeping of `_record_collisions` (:220-230) comes back in
istate row 3.  Collisions with obstacles are the position-in-AABB test (the reference asks MuJoCo for geometry
contacts).  Validation errors mirror the reference's `ValueError`s (mujoco_sim.py:261-266, 289-291, 317-320,
331-332, 339-347).

Only world-frame geometry directly under `<worldbody>` is understood (that is all the reference's
scene uses for planning data); `<replicate>` visual helpers and assets are ignored.
"""
from __future__ import annotations

import ctypes as C
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

from .. import _native as nat
from .._single import ctx, vehicle_from
from ..quadrotor.quad import Quad

ENU_TO_NED = np.diag([1.0, -1.0, -1.0])


def mujoco_to_ned_state(position, quaternion, velocity) -> np.ndarray:
    """ENU/FLU free-joint state -> NED/FRD controller state (reference mujoco_sim.py:20-45)."""
    position = _vector(position, 3, "position")
    quaternion = _vector(quaternion, 4, "quaternion")
    velocity = _vector(velocity, 6, "velocity")
    norm = np.linalg.norm(quaternion)
    if norm == 0:
        raise ValueError("MuJoCo quaternion cannot be zero")
    state = np.empty(13)
    state[:3] = ENU_TO_NED @ position
    state[3:7] = quaternion / norm * np.array([1.0, 1.0, -1.0, -1.0])
    state[7:10] = ENU_TO_NED @ velocity[:3]
    state[10:13] = ENU_TO_NED @ velocity[3:]
    return state


def _vector(values, size: int, name: str) -> np.ndarray:
    v = np.asarray(values, dtype=float)
    if v.shape != (size,) or not np.all(np.isfinite(v)):
        raise ValueError(f"MuJoCo {name} must contain {size} finite values")
    return v


def _floats(text, n=None, what="attribute"):
    vals = np.array([float(t) for t in str(text).split()], dtype=float)
    if n is not None and len(vals) != n:
        raise ValueError(f"MuJoCo {what} must contain {n} values")
    return vals


DEFAULT_SCENE_PATH = Path(__file__).parent / "models" / "lab_course.xml"


class MujocoSimulation:
    """Scene data + free-flight vehicle.  Same attribute names as the reference adapter."""

    def __init__(self, model_path: str | Path = DEFAULT_SCENE_PATH):
        root = ET.parse(str(model_path)).getroot()
        world = root.find("worldbody")
        if world is None:
            raise ValueError("MuJoCo scene is missing required element 'worldbody'")
        option = root.find("option")
        self.timestep = float(option.get("timestep", "0.002")) if option is not None else 0.002
        gravity = _floats(option.get("gravity", "0 0 -9.81"), 3, "gravity") if option is not None else np.array([0, 0, -9.81])
        g = float(np.linalg.norm(gravity))
        if g == 0:
            raise ValueError("MuJoCo gravity must be non-zero")
        self._numerics = {n.get("name"): _floats(n.get("data", "")) for n in root.iterfind("custom/numeric")}

        body = next((b for b in world.iter("body") if b.get("name") == "quadrotor"), None)
        if body is None:
            raise ValueError("MuJoCo scene is missing required element 'quadrotor'")
        inertial = body.find("inertial")
        if inertial is None:
            raise ValueError("MuJoCo scene is missing required element 'inertial'")
        rotors = {s.get("name"): s for s in body.iterfind("site")}
        rotor_pos, spins = [], []
        for i in range(4):
            site = rotors.get(f"rotor_{i}")
            if site is None:
                raise ValueError(f"MuJoCo scene is missing required element 'rotor_{i}'")
            rotor_pos.append(_floats(site.get("pos"), 3, "rotor position"))
            spins.append(_floats(site.get("user", "0"))[0])
        arm = np.abs(np.array(rotor_pos)[:, :2])
        if not np.allclose(arm, arm[0, 0]):
            raise ValueError("MuJoCo rotor sites must use a symmetric X configuration")
        self.rotor_spin_directions = np.array(spins)

        self.quad = Quad(
            g=g, dt=self.timestep, mass=float(inertial.get("mass")),
            inertia=_floats(inertial.get("diaginertia"), 3, "diaginertia"), arm_length=arm[0, 0],
            force_coefficient=self._numeric("rotor_force_coefficient", 1)[0],
            drag_to_thrust=self._numeric("rotor_drag_to_thrust", 1)[0],
            thrust_limits=self._numeric("rotor_thrust_limits", 2),
            motor_time_constants=self._numeric("motor_time_constants", 2),
            flight_limits=self._numeric("flight_limits", 5))
        start_enu = _floats(body.get("pos", "0 0 0"), 3, "body position")
        self.quad.X = mujoco_to_ned_state(start_enu, np.array([1.0, 0, 0, 0]), np.zeros(6))
        self.start_position = self.quad.position.copy()

        sites = {s.get("name"): s for s in world.findall("site")}
        if "goal" not in sites:
            raise ValueError("MuJoCo scene is missing required element 'goal'")
        self.goal_position = ENU_TO_NED @ _floats(sites["goal"].get("pos"), 3, "goal")
        names = sorted(n for n in sites if n and n.startswith("waypoint_"))
        if names != [f"waypoint_{i:02d}" for i in range(len(names))]:
            raise ValueError("MuJoCo mission waypoints must be consecutively numbered from waypoint_00")
        if not names:
            raise ValueError("MuJoCo scene must define at least one mandatory waypoint")
        mandatory = np.array([ENU_TO_NED @ _floats(sites[n].get("pos"), 3, n) for n in names])
        self.mission_waypoints = np.vstack((self.start_position, mandatory, self.goal_position))
        self.space_limits = self._numeric("planning_bounds", 6).reshape(2, 3)

        obstacles = []
        for geom in world.findall("geom"):
            name = geom.get("name")
            if not name or not name.startswith("obstacle_"):
                continue
            if geom.get("type") != "box":
                raise ValueError(f"MuJoCo planning obstacle '{name}' must be an axis-aligned box")
            if any(geom.get(a) is not None for a in ("quat", "euler", "axisangle", "xyaxes", "zaxis")):
                raise ValueError(f"MuJoCo planning obstacle '{name}' must be axis-aligned")
            c = ENU_TO_NED @ _floats(geom.get("pos", "0 0 0"), 3, name)
            h = _floats(geom.get("size"), 3, name)
            obstacles.append([c[0] - h[0], c[0] + h[0], c[1] - h[1], c[1] + h[1], c[2] - h[2], c[2] + h[2]])
        self.obstacles = np.asarray(obstacles, dtype=float).reshape(-1, 6)
        # ground plane (lab_course.xml:34) and the body box that rests on it (:101): NED height of the plane and the
        # distance from the body centre to its lowest point
        ground = next((g_ for g_ in world.findall("geom") if g_.get("name") == "ground" and g_.get("type") == "plane"), None)
        box = next((g_ for g_ in body.findall("geom") if g_.get("name") == "body" and g_.get("type", "sphere") == "box"), None)
        self.ground_z = None if ground is None else float(-_floats(ground.get("pos", "0 0 0"), 3, "ground")[2])
        self.ground_clearance = 0.0 if box is None else float(_floats(box.get("size"), 3, "body size")[2])
        self._collision_detected = False
        self._ground_bits = 0
        if self.has_collision:
            raise ValueError("quadrotor starts in collision")

    def _numeric(self, name: str, expected_size: int) -> np.ndarray:
        if name not in self._numerics:
            raise ValueError(f"MuJoCo scene is missing required element '{name}'")
        if len(self._numerics[name]) != expected_size:
            raise ValueError(f"MuJoCo numeric '{name}' must contain {expected_size} values")
        return self._numerics[name].copy()

    def vehicle(self, **kw) -> nat.Vehicle:
        """`uavac_vehicle` of this scene: the quad's constants and gains + the ground plane when the scene has one."""
        V = vehicle_from(self.quad, **kw)
        if self.ground_z is not None:
            V.ground, V.ground_z, V.ground_clearance = 1, self.ground_z, self.ground_clearance
        return V

    def _touches_ground(self) -> bool:
        return self.ground_z is not None and bool(self.quad.position[2] - (self.ground_z - self.ground_clearance) > 0.0)

    @property
    def has_collision(self) -> bool:
        """The vehicle touches world geometry right now (reference mujoco_sim.py:93-96 asks MuJoCo for contacts): its
        lowest point is below the ground plane, or its position is inside a planning obstacle."""
        p, o = self.quad.position, self.obstacles
        inside = bool(len(o)) and bool(np.any((p[0] >= o[:, 0]) & (p[0] <= o[:, 1]) & (p[1] >= o[:, 2]) & (p[1] <= o[:, 3]) &
                                              (p[2] >= o[:, 4]) & (p[2] <= o[:, 5])))
        return inside or self._touches_ground()

    @property
    def collision_detected(self) -> bool:
        """Sticky (reference :98-101, 220-230): an obstacle has been entered, or the ground touched after take-off
        (ground contact before reaching TAKEOFF_HEIGHT is the start, not a collision)."""
        return self._collision_detected

    def step(self) -> np.ndarray:
        """One inner-loop time step from the current rotor speeds (reference mujoco_sim.py:144-151): `uavac_pilot_tick`
        (vehicle half) on state in pinned memory the kernel updates in place; the obstacle list is resident."""
        q = self.quad
        if getattr(self, "_pilot", None) is None:
            self._pilot = nat.Pilot(ctx(), np.zeros((1, nat.TRAJ_COLS)), np.array([0, 1], dtype=np.int64))
            self._pilot.set_obstacles(self.obstacles if len(self.obstacles) else None)
        st, ist = self._pilot.state[:, 0], self._pilot.istate[:, 0]
        st[0:13] = q.X
        st[13:17] = q.omega
        ist[2], ist[3] = 0, self._ground_bits
        self._pilot.tick(self.vehicle(), nat.PILOT_DYNAMICS)
        q.X = st[0:13].copy()
        self._ground_bits = int(ist[3])
        self._collision_detected = (self._collision_detected or bool(ist[2]) or
                                    bool(self._ground_bits & nat.GROUND_HIT_AFTER_TAKEOFF))
        return q.X.copy()
