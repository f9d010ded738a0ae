"""Process-wide context for the single-UAV facade classes (B = 1 calls into libuavac.so)."""
from __future__ import annotations

from . import _native as nat

_ctx = None


def ctx() -> nat.Context:
    """Lazily created on first numeric call; raises UavacError when there is no GPU (no CPU fallback)."""
    global _ctx
    if _ctx is None:
        _ctx = nat.Context(-1)
    return _ctx


def vehicle_from(quad=None, g=None, dt_outer=None, **gains) -> nat.Vehicle:
    """uavac_vehicle from a (possibly duck-typed) reference-style quad object + explicit gains."""
    V = nat.Vehicle.default()
    if quad is not None:
        for dst, src in (("g", "g"), ("dt", "dt"), ("mass", "m"), ("arm", "l"), ("kf", "kf"), ("kappa", "kappa"),
                         ("min_thrust", "min_thrust"), ("max_thrust", "max_thrust"),
                         ("tau_rise", "motor_rise_time_constant"), ("tau_fall", "motor_fall_time_constant"),
                         ("max_ascent", "max_ascent_rate"), ("max_descent", "max_descent_rate"),
                         ("max_speed_xy", "max_speed_xy"), ("max_horiz_accel", "max_horiz_accel"),
                         ("max_tilt", "max_tilt_angle"), ("kp_xy", "kp_xy"), ("kd_xy", "kd_xy"), ("kp_z", "kp_z"),
                         ("kd_z", "kd_z"), ("ki_z", "ki_z"), ("kp_roll", "kp_roll"), ("kp_pitch", "kp_pitch"),
                         ("kp_yaw", "kp_yaw"), ("kp_p", "kp_p"), ("kp_q", "kp_q"), ("kp_r", "kp_r")):
            if hasattr(quad, src):
                setattr(V, dst, float(getattr(quad, src)))
        if all(hasattr(quad, a) for a in ("i_x", "i_y", "i_z")):
            V.inertia[:] = [float(quad.i_x), float(quad.i_y), float(quad.i_z)]
    if g is not None:
        V.g = float(g)
    V.dt_outer = float(dt_outer) if dt_outer is not None else V.dt * V.inner_per_outer
    for k, v in gains.items():
        setattr(V, k, float(v))
    return V
