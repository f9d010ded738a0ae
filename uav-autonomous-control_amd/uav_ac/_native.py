"""ctypes binding of libuavac.so (include/uavac.h).

There is no CPU fallback: if the shared library is missing or no GPU is usable,
importing the symbols works (so that `-m "not gpu"` tests can check the ABI) but
creating a context raises.  Build with `make -C uav-autonomous-control_amd` or
`python -c "import __graft_entry__ as g; g.build()"`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# UAVAC_LIB: another build of the same ABI (development: A/B of two builds on one box, tools/)
LIB_PATH = os.environ.get("UAVAC_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libuavac.so")

OK, EINVAL, ENONFINITE, EHIP, ESINGULAR, ENOMEM, ECOMM, ETOOLCHAIN = 0, -1, -2, -3, -4, -5, -6, -7
COMM_ID_BYTES = 128
MAX_SEGMENTS = 64
TRAJ_COLS, STATE_ROWS, ISTATE_ROWS, CMD_COLS = 11, 30, 4, 12
VERSION = 310
GROUND_IN_CONTACT, GROUND_TAKEN_OFF, GROUND_HIT_AFTER_TAKEOFF = 1, 2, 4       # istate row 3 (include/uavac.h)


class UavacError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libuavac error {code}: {msg}")
        self.code = code


class Vehicle(C.Structure):
    """Mirror of `uavac_vehicle` (include/uavac.h); defaults = lab_course.xml + quad.py:42-73."""
    _fields_ = [(n, C.c_double) for n in ("g", "dt", "dt_outer", "mass")] + [("inertia", C.c_double * 3)] + \
               [(n, C.c_double) for n in (
                   "arm", "kf", "kappa", "min_thrust", "max_thrust", "tau_rise", "tau_fall",
                   "max_ascent", "max_descent", "max_speed_xy", "max_horiz_accel", "max_tilt",
                   "kp_xy", "kd_xy", "kp_z", "kd_z", "ki_z", "kp_roll", "kp_pitch", "kp_yaw",
                   "kp_p", "kp_q", "kp_r")] + [("inner_per_outer", C.c_int32), ("ground", C.c_int32)] + \
               [(n, C.c_double) for n in ("ground_z", "ground_clearance", "ground_timeconst")]

    @classmethod
    def default(cls) -> "Vehicle":
        v = cls()
        lib().uavac_vehicle_default(C.byref(v))
        return v

    def copy(self) -> "Vehicle":
        v = Vehicle()
        C.memmove(C.byref(v), C.byref(self), C.sizeof(Vehicle))
        return v


_P = C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "uavac_version": (C.c_int, []),
    "uavac_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "uavac_destroy": (None, [_P]),
    "uavac_last_error": (C.c_char_p, [_P]),
    "uavac_set_stream": (C.c_int, [_P, _P]),
    "uavac_reset_stream": (C.c_int, [_P]),
    "uavac_synchronize": (C.c_int, [_P]),
    "uavac_device": (C.c_int, [_P]),
    "uavac_last_rollout_kernel": (C.c_char_p, [_P]),
    "uavac_last_rollout_vgprs": (C.c_int, [_P]),
    "uavac_device_identity": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "uavac_clock_probe_dev": (C.c_int, [_P, C.c_int, _P]),
    "uavac_build_info": (C.c_char_p, []),
    "uavac_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "uavac_take_flags": (C.c_int, [_P, _P]),
    "uavac_vehicle_default": (None, [C.POINTER(Vehicle)]),
    "uavac_minsnap_row_counts_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_double, C.c_double, _P, _P, _P]),
    "uavac_minsnap_solve_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "uavac_minsnap_solve_banded_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "uavac_minsnap_sample_dev": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_double, _P]),
    "uavac_minsnap_sample_yaw_dev": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P]),
    "uavac_minsnap_sample_hits_dev": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P, _P]),
    "uavac_minsnap_sample_derivs_dev": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P, _P]),
    "uavac_minsnap_plan_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_double, C.c_double, _P, _P, _P, _P, _P, _P,
                                          C.c_int64, _P, _P]),
    "uavac_minsnap_first_yaw_dev": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_double, _P]),
    "uavac_minsnap_row_offsets_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "uavac_minsnap_row_offsets_ragged_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "uavac_minsnap_plan_ragged": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_double, _P, _P, _P, _P, C.c_int64]),
    "uavac_minsnap_row_counts_ragged_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, C.c_double, _P, _P, _P]),
    "uavac_minsnap_solve_ragged_dev": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "uavac_minsnap_sample_ragged_dev": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int64, C.c_double, _P,
                                                   C.c_int64, _P, _P, _P]),
    "uavac_minsnap_obstacle_round_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, C.c_double] + [_P] * 12),
    "uavac_minsnap_obstacle_waypoints": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_double, _P, C.c_int, C.c_int, C.c_int, _P,
                                                    C.c_int64, _P, _P]),
    "uavac_yaw_scan_dev": (C.c_int, [_P, _P, _P, C.c_int, _P]),
    "uavac_yaw_scan": (C.c_int, [_P, _P, C.c_int64, _P]),
    "uavac_minsnap_row_counts": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_double, C.c_double, _P, _P, _P]),
    "uavac_minsnap_solve": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_double, _P, _P]),
    "uavac_minsnap_sample": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_double, _P, _P]),
    "uavac_state_init_dev": (C.c_int, [_P, C.POINTER(Vehicle), _P, C.c_int, C.c_int, _P, _P]),
    "uavac_control_rollout_dev": (C.c_int, [_P, C.POINTER(Vehicle), _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int]),
    "uavac_control_rollout_plan_dev": (C.c_int, [_P, C.POINTER(Vehicle), _P, _P, _P, _P, _P, C.c_int, C.c_double, _P, _P,
                                                  C.c_int, C.c_int, _P, _P, _P, C.c_int]),
    "uavac_control_rollout_plan_ragged_dev": (C.c_int, [_P, C.POINTER(Vehicle), _P, _P, _P, _P, _P, C.c_int, C.c_double, _P, _P,
                                                         C.c_int, C.c_int, _P, _P, _P, C.c_int]),
    "uavac_control_step_dev": (C.c_int, [_P, C.POINTER(Vehicle), _P, _P, _P, _P, C.c_int]),
    "uavac_state_init": (C.c_int, [_P, C.POINTER(Vehicle), _P, C.c_int, C.c_int, _P, _P]),
    "uavac_control_rollout": (C.c_int, [_P, C.POINTER(Vehicle), _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, C.c_int]),
    "uavac_controller_tick_dev": (C.c_int, [_P, C.POINTER(Vehicle), _P, _P, _P, _P, C.c_int]),
    "uavac_dynamics_step_dev": (C.c_int, [_P, C.POINTER(Vehicle), _P, _P, C.c_int, _P, C.c_int]),
    "uavac_controller_tick": (C.c_int, [_P, C.POINTER(Vehicle), _P, _P, _P, _P, C.c_int]),
    "uavac_dynamics_step": (C.c_int, [_P, C.POINTER(Vehicle), _P, _P, C.c_int, _P, C.c_int]),
    "uavac_pilot_create": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(_P)]),
    "uavac_pilot_destroy": (None, [_P]),
    "uavac_pilot_state": (C.POINTER(C.c_double), [_P]),
    "uavac_pilot_istate": (C.POINTER(C.c_int32), [_P]),
    "uavac_pilot_set_obstacles": (C.c_int, [_P, _P, C.c_int]),
    "uavac_pilot_tick": (C.c_int, [_P, C.POINTER(Vehicle), C.c_int]),
    "uavac_probe_outer": (C.c_int, [_P, C.POINTER(Vehicle), _P, C.c_int, C.c_int, _P]),
    "uavac_probe_inner": (C.c_int, [_P, C.POINTER(Vehicle), _P, C.c_int, C.c_int, _P]),
    "uavac_probe_heading_dev": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P]),
    "uavac_rrt_star_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_int, _P, _P, C.c_int] + [_P] * 7),
    "uavac_rrt_star": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_int, _P, _P, C.c_int] + [_P] * 7),
    "uavac_rrt_segment_hits_dev": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, _P]),
    "uavac_rrt_segment_hits": (C.c_int, [_P, _P, _P, C.c_int, _P, C.c_int, _P]),
    "uavac_rrt_edge_lengths_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "uavac_rrt_edge_lengths": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P]),
    "uavac_rrt_draw_nodes_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P, C.c_double, _P, _P]),
    "uavac_rrt_simplify_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "uavac_rrt_simplify": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "uavac_rrt_path_cost_dev": (C.c_int, [_P, _P, C.c_int, _P]),
    "uavac_rrt_path_cost": (C.c_int, [_P, _P, C.c_int, _P]),
    "uavac_rrt_steer_dev": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, _P]),
    "uavac_rrt_steer": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, _P]),
    "uavac_comm_unique_id": (C.c_int, [_P, _P]),
    "uavac_comm_init_rank": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "uavac_comm_destroy": (C.c_int, [_P, _P]),
    "uavac_comm_abort": (C.c_int, [_P, _P]),
    "uavac_comm_shape": (C.c_int, [_P, _P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "uavac_gather_counts": (C.c_int, [_P, _P, C.c_int64, _P]),
    "uavac_gather_rows_dev": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P, C.c_int, _P]),
    "uavac_gather_plan_dev": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _P, C.c_int, _P, _P, _P]),
    "uavac_gather_plan_part_dev": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P]),
    "uavac_comm_versions": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "uavac_comm_finish": (C.c_int, [_P, _P]),
    "uavac_comm_loopback_dev": (C.c_int, [_P, _P, _P, _P, C.c_int64]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libuavac.so once.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            # never a fallback: either the HIP library gets built from source on request (UAVAC_AUTOBUILD=1; hipcc
            # cross-compiles for gfx950), or loading fails loudly with the command that builds it
            pkg = os.path.dirname(os.path.dirname(LIB_PATH))
            if os.environ.get("UAVAC_AUTOBUILD") != "1":
                raise UavacError(EHIP, f"{LIB_PATH} has not been built: run `make -C {pkg}` (or set UAVAC_AUTOBUILD=1 "
                                       "to let the first import do it); there is no CPU fallback")
            import subprocess
            try:
                subprocess.run(["make", "-C", pkg, "-j4", "lib"], check=True, capture_output=True)
            except Exception as exc:
                raise UavacError(EHIP, f"{LIB_PATH} not built and `make -C {pkg}` failed: {exc}") from exc
            # a library built here is CHECKED here, like one built by __graft_entry__.build(): register budgets and the
            # hand-placed prefetches in the disassembly (the compiler on this box may not be the one the checks last saw)
            from . import _buildcheck
            try:
                _buildcheck.run_all()
            except Exception as exc:
                try:
                    os.replace(LIB_PATH, LIB_PATH + ".failed-buildcheck")      # never load (or leave for the next process) an unchecked build
                except OSError:
                    pass
                raise UavacError(EHIP, f"{LIB_PATH} was built but FAILED its build checks and was set aside: {exc}") from exc
        try:                                   # share torch's HIP runtime when torch is in the process
            import torch  # noqa: F401
        except Exception:                      # pragma: no cover - torch is plumbing, not a requirement of the ABI
            pass
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(_lib, name)
            except AttributeError:
                if os.environ.get("UAVAC_LIB"):        # an older build loaded for an A/B: it may lack the newest entry points
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


_live_contexts = None


def _close_live_contexts():
    for ref in list(_live_contexts or ()):
        ctx = ref()
        if ctx is not None:
            ctx.close()


class Context:
    """Owns one `uavac_ctx` (one HIP stream on one GPU)."""

    def __init__(self, device: int = -1):
        global _live_contexts
        self._h = _P()
        rc = lib().uavac_create(C.byref(self._h), -1 if device is None else int(device))
        if rc != OK:
            self._h = _P()
            if rc == ETOOLCHAIN:
                raise UavacError(rc, "uavac_create refused: this build's heading() and the atan2 of the device library it linked differ "
                                     "(see stderr): heading() in csrc/minsnap_yaw.h must be re-derived for that library")
            raise UavacError(rc, "uavac_create failed: no usable MI355X / HIP runtime (there is no CPU fallback)")
        # contexts are destroyed by an atexit hook registered AFTER torch's HIP runtime came up, i.e. run BEFORE its
        # teardown -- not left to __del__ during interpreter shutdown, when the runtime may already be gone
        import atexit
        import weakref
        if _live_contexts is None:
            _live_contexts = set()
            atexit.register(_close_live_contexts)
        self._ref = weakref.ref(self)
        _live_contexts.add(self._ref)

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and _lib is not None:      # at interpreter shutdown the module may be gone
            try:
                _lib.uavac_destroy(h)
            except Exception:                                    # pragma: no cover
                pass
            self._h = _P()
        if _live_contexts is not None:
            _live_contexts.discard(getattr(self, "_ref", None))

    __del__ = close

    def check(self, rc: int):
        if rc != OK:
            msg = lib().uavac_last_error(self._h)
            raise UavacError(rc, msg.decode() if msg else "")

    def call(self, name: str, *args):
        self.check(getattr(lib(), name)(self._h, *args))

    def set_stream(self, stream_handle):
        """Borrow a hipStream_t handle; 0 / None is HIP's legacy default stream (torch's default)."""
        self.call("uavac_set_stream", _P(stream_handle or None))

    def reset_stream(self):
        self.call("uavac_reset_stream")

    def synchronize(self):
        self.call("uavac_synchronize")

    def set_option(self, name: str, value: int):
        self.call("uavac_set_option", name.encode(), int(value))

    def last_rollout_kernel(self) -> str:
        return (lib().uavac_last_rollout_kernel(self._h) or b"").decode()

    def last_rollout_vgprs(self) -> int:
        return int(lib().uavac_last_rollout_vgprs(self._h))

    def device_identity(self) -> str:
        """"uuid=...;pci=...;name=..." of the GPU this ctx runs on."""
        buf = C.create_string_buffer(160)
        self.call("uavac_device_identity", buf, 160)
        return buf.value.decode()


PILOT_CONTROLLER, PILOT_DYNAMICS = 1, 2


class Pilot:
    """Resident tick-by-tick session (`uavac_pilot_*`): trajectory rows on the device, state in pinned mapped host memory
    exposed as NumPy views `state` (30, B) and `istate` (4, B) that the kernels update in place."""

    def __init__(self, ctx: Context, traj: np.ndarray, row_offsets: np.ndarray):
        self._ctx = ctx
        self._h = _P()
        traj = as_f64(traj)
        offs = np.ascontiguousarray(row_offsets, dtype=np.int64)
        self.B = len(offs) - 1
        ctx.check(lib().uavac_pilot_create(ctx._h, np_ptr(traj), np_ptr(offs), self.B, C.byref(self._h)))
        self.state = np.ctypeslib.as_array(lib().uavac_pilot_state(self._h), shape=(STATE_ROWS, self.B))
        self.istate = np.ctypeslib.as_array(lib().uavac_pilot_istate(self._h), shape=(ISTATE_ROWS, self.B))

    def set_obstacles(self, aabbs):
        a = None if aabbs is None else as_f64(aabbs).reshape(-1, 6)
        n = 0 if a is None else len(a)
        self._ctx.check(lib().uavac_pilot_set_obstacles(self._h, np_ptr(a) if n else None, n))

    def tick(self, vehicle: Vehicle, what: int = PILOT_CONTROLLER | PILOT_DYNAMICS):
        self._ctx.check(lib().uavac_pilot_tick(self._h, C.byref(vehicle), int(what)))

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and _lib is not None and self._ctx._h.value:
            self.state = self.istate = None
            _lib.uavac_pilot_destroy(h)
            self._h = _P()

    __del__ = close


def np_ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(_P)


def as_f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a
