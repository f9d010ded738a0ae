"""Pin SURVEY 8(a) row D2 (the free-body step) against a real MuJoCo -- for a machine that HAS `mujoco` installed.

The build container and the GPU boxes do not (no network), so this script has never been executed by the build: it is
the recipe a maintainer runs once (`pip install mujoco==3.11.0`, the version the reference pins in uv.lock) to turn the
"parity unpinned" note on D2 into a fixture:

    python3 tools/mujoco_pin.py            -> tests/golden/mujoco_trace.npz
    python3 -m pytest tests/test_oracle_control.py -k mujoco_trace     (skipped while the fixture is absent)

What it records: a free body with Table V's mass / inertia / rotor sites (no geoms: no contacts), MuJoCo's Euler
integrator at dt = 1 ms, driven for 2 000 steps by a seeded rotor-speed sequence around hover from a tilted, moving,
spinning start.  Rotor forces are applied the way the reference adapter applies them (uav_ac/simulation/mujoco_sim.py:232-251:
force kf w^2 along body +z at each rotor site, reaction torque spin * kappa * f about body z, `mj_applyFT` into
`qfrc_applied`), but with the kinematics of the CURRENT state (`mj_forward` first) -- the definition SURVEY D2 chose; the
reference's scripted loop uses the previous step's `xmat`, its viewer callback the current one.  States are stored in NED /
FRD through the build's own adapter (`mujoco_to_ned_state`, pinned by the reference's known answer).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "uav-autonomous-control_amd")]

try:
    import mujoco
except ImportError:
    sys.exit("mujoco is not installed here: run this on a machine that has it (pip install mujoco==3.11.0)")

from uav_ac.simulation.mujoco_sim import mujoco_to_ned_state

ARM, KF, KAPPA, MASS = 0.120208, 1.0, 0.016, 0.5
SPIN = (1.0, -1.0, 1.0, -1.0)
SITES = ((ARM, ARM), (ARM, -ARM), (-ARM, -ARM), (-ARM, ARM))            # rotor_0..3 in body FLU (front-left first)
XML = f"""
<mujoco model="d2_pin">
  <compiler angle="radian"/>
  <option timestep="0.001" gravity="0 0 -9.81" integrator="Euler"/>
  <worldbody>
    <body name="quadrotor" pos="1 -7 2">
      <freejoint/>
      <inertial pos="0 0 0" mass="{MASS}" diaginertia="0.0023 0.0023 0.0046"/>
      {''.join(f'<site name="rotor_{i}" pos="{x} {y} 0" size="0.008"/>' for i, (x, y) in enumerate(SITES))}
    </body>
  </worldbody>
</mujoco>
"""
K = 2000


def main():
    model = mujoco.MjModel.from_xml_string(XML)
    data = mujoco.MjData(model)
    body = mujoco.mj_name2id(model, mujoco.mjtObj.mjOBJ_BODY, "quadrotor")
    sites = [mujoco.mj_name2id(model, mujoco.mjtObj.mjOBJ_SITE, f"rotor_{i}") for i in range(4)]
    rng = np.random.default_rng(20260807)
    # start: 0.3 rad of tilt about a skew axis, 1.5 m/s of velocity, body rates of ~1 rad/s
    axis = np.array([0.6, -0.5, 0.62]); axis /= np.linalg.norm(axis)
    data.qpos[3:7] = np.concatenate([[np.cos(0.15)], np.sin(0.15) * axis])
    data.qvel[:] = [1.0, -0.8, 0.7, 0.9, -1.1, 0.6]
    hover = np.sqrt(MASS * 9.81 / (4 * KF))
    t = np.arange(K)[:, None] * 1e-3
    phase = rng.uniform(0, 2 * np.pi, (1, 4))
    omega = hover * (1.0 + 0.15 * np.sin(2 * np.pi * rng.uniform(0.5, 4.0, (1, 4)) * t + phase)
                     + 0.02 * rng.standard_normal((K, 4)))
    X = np.empty((K + 1, 13))
    X[0] = mujoco_to_ned_state(data.qpos[:3], data.qpos[3:7], data.qvel[:6])
    for k in range(K):
        mujoco.mj_forward(model, data)                      # xmat / site_xpos of the state the step starts from
        data.qfrc_applied[:] = 0.0
        R = data.xmat[body].reshape(3, 3)
        for i, s in enumerate(sites):
            f = KF * omega[k, i] ** 2
            mujoco.mj_applyFT(model, data, R @ np.array([0.0, 0.0, f]), R @ np.array([0.0, 0.0, SPIN[i] * KAPPA * f]),
                              data.site_xpos[s], body, data.qfrc_applied)
        mujoco.mj_step(model, data)
        X[k + 1] = mujoco_to_ned_state(data.qpos[:3], data.qpos[3:7], data.qvel[:6])
    out = os.path.join(ROOT, "tests", "golden", "mujoco_trace.npz")
    np.savez_compressed(out, omega=omega, X=X, dt=1e-3, mujoco_version=mujoco.__version__)
    print(f"wrote {out}: {K} steps, MuJoCo {mujoco.__version__}")


if __name__ == "__main__":
    main()
