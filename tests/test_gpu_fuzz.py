"""Randomised differential test: ragged mission batches x velocities x sample periods x vehicles, the HIP path against the
NumPy planner oracle (reference formulation, exact solve) and the scalar C control oracle.  A few draws by default;
UAVAC_FUZZ_ITERS=<n> for a soak."""
import os

import numpy as np
import pytest

from conftest import col_err

pytestmark = pytest.mark.gpu
ITERS = int(os.environ.get("UAVAC_FUZZ_ITERS", "6"))


@pytest.fixture(scope="module")
def eng():
    from uav_ac.fleet import Engine
    return Engine("cuda:0")


def _missions(rng, n):
    out = []
    for _ in range(n):
        m = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 17]))
        d = rng.standard_normal((m, 3)) * np.array([1, 1, rng.choice([0.0, 0.25, 1.0])])
        if rng.random() < 0.15:
            d[0, :2] = 0.0; d[0, 2] = -1.0                        # starts with a vertical leg: no heading for a while
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        L = rng.uniform(1.0, 6.0, (m, 1))
        w0 = np.array([rng.uniform(0, 24), rng.uniform(0, 14), -rng.uniform(1, 5)])
        out.append(np.concatenate([w0[None], w0 + np.cumsum(L * d, axis=0)]))
    return out


@pytest.mark.parametrize("it", range(ITERS))
def test_random_ragged_batch_against_the_oracles(eng, it):
    from uav_ac import _native as nat
    from oracle import c_oracle as cc
    from oracle import minsnap_oracle as mo
    rng = np.random.default_rng(9000 + it)
    B = int(rng.integers(1, 160))
    velocity, dt = float(rng.uniform(0.8, 4.0)), float(rng.choice([0.005, 0.01, 0.02]))
    missions = _missions(rng, B)
    rb = eng.plan_ragged(missions, velocity, dt)
    ro = rb.row_offsets.cpu().numpy()
    for b in rng.choice(B, size=min(B, 4), replace=False):
        ref = mo.plan(missions[b], velocity, dt, method="solve")
        got = rb.mission(int(b))
        assert got.shape == ref.shape, (it, b, got.shape, ref.shape)
        # the yaw column modulo 2 pi: where the horizontal velocity reverses through (almost) zero the heading steps by pi to
        # the last bit, and whether np.unwrap then adds 2 pi hangs on the rounding of two atan2 results (libm's there, the
        # GPU's here) -- the same direction either way, and the yaw law (controller.py:156-168) works modulo 2 pi
        yaw_err = np.abs(np.angle(np.exp(1j * (got[:, 9] - ref[:, 9]))))
        rest = [c for c in range(11) if c != 9]
        assert col_err(got[:, rest], ref[:, rest]) < 1e-5 and yaw_err.max() < 1e-5, (it, b)
        assert ro[b + 1] - ro[b] == len(ref)
    # fly it with a vehicle drawn around Table V, outer period = the sample period
    V, Vc = nat.Vehicle.default(), cc.Vehicle.default()
    F = max(1, int(round(dt / 0.001)))
    for name, val in (("inner_per_outer", F), ("dt_outer", 0.001 * F), ("mass", float(rng.uniform(0.4, 0.8))),
                      ("max_tilt", float(rng.uniform(0.5, 0.8)))):
        setattr(V, name, val); setattr(Vc, name, val)
    for name in ("kp_xy", "kd_xy", "kp_z", "kd_z", "kp_roll", "kp_pitch", "kp_yaw", "kp_p", "kp_q", "kp_r", "tau_rise", "tau_fall"):
        v = getattr(V, name) * float(rng.uniform(0.85, 1.2)); setattr(V, name, v); setattr(Vc, name, v)
    K = 800
    logs = {}
    for from_plan in (False, True):
        fleet = eng.fleet(rb, vehicle=V, from_plan=from_plan)
        logs[from_plan] = fleet.rollout(K, state_log=True, cmd_log=True)
    import torch
    assert torch.equal(logs[False][0], logs[True][0]) and torch.equal(logs[False][1], logs[True][1])
    for b in rng.choice(B, size=min(B, 2), replace=False):
        traj = rb.mission(int(b))
        state, istate = cc.initial_state(traj[0, 0:3], Vc)
        s_ref, c_ref = cc.rollout(traj, state, istate, K, Vc)
        s = logs[False][0][:, :, int(b)].cpu().numpy()
        if np.isfinite(s_ref).all() and np.abs(s_ref[:, 10:13]).max() < 40.0:      # a vehicle that has not tumbled away
            assert col_err(s, s_ref) < 1e-7, (it, b)
            assert col_err(logs[False][1][:, :, int(b)].cpu().numpy(), c_ref) < 1e-7, (it, b)


@pytest.mark.parametrize("it", range(ITERS))
def test_random_takeoffs_obstacles_and_log_combinations(eng, it):
    """Vehicles that start on the ground plane with stopped rotors (the reference's true start) below missions drawn at
    random, a handful of random cuboids tested on every tick, every combination of logs: the flags, the ground bookkeeping
    and the states do not depend on which logs are taken (the obstacle test moves between the waves with them), the flags
    equal a recomputation from the logged positions, and two lanes equal the C oracle."""
    import torch
    from uav_ac import _native as nat
    from oracle import c_oracle as cc
    rng = np.random.default_rng(7000 + it)
    B = int(rng.integers(1, 200))
    missions = _missions(rng, B)
    for w in missions:                                            # from the ground (z = -0.02: the body rests on the plane)
        w[0, 2] = -0.02
        w[1:, 2] = np.minimum(w[1:, 2], -1.0)
    velocity, dt = float(rng.uniform(0.8, 2.5)), 0.01
    rb = eng.plan_ragged(missions, velocity, dt)
    V, Vc = nat.Vehicle.default(), cc.Vehicle.default()
    V.ground = Vc.ground = 1
    n_obs = int(rng.integers(1, 12))
    lo = np.stack([rng.uniform(0, 20, n_obs), rng.uniform(0, 12, n_obs), -rng.uniform(2, 6, n_obs)], axis=1)
    size = rng.uniform(0.5, 4.0, (n_obs, 3))
    boxes = np.stack([lo[:, 0], lo[:, 0] + size[:, 0], lo[:, 1], lo[:, 1] + size[:, 1], lo[:, 2], lo[:, 2] + size[:, 2]], axis=1)
    K = 1500
    results = []
    for state_log, cmd_log in ((True, False), (False, False), (True, True), (False, True)):
        fleet = eng.fleet(rb, vehicle=V, hover=False, from_plan=bool(rng.integers(0, 2)))
        slog, clog = fleet.rollout(K, state_log=state_log or None, cmd_log=cmd_log or None, aabbs=boxes)
        results.append((fleet.state[:26].clone(), fleet.istate.clone(), slog))
    for st, ist, _ in results[1:]:
        assert torch.equal(st, results[0][0]) and torch.equal(ist, results[0][1])
    slog = results[0][2]
    inside = torch.zeros(B, dtype=torch.bool, device=slog.device)
    for c in boxes:
        inside |= ((slog[:, 0] >= c[0]) & (slog[:, 0] <= c[1]) & (slog[:, 1] >= c[2]) & (slog[:, 1] <= c[3]) &
                   (slog[:, 2] >= c[4]) & (slog[:, 2] <= c[5])).any(dim=0)
    assert torch.equal(results[0][1][2].bool(), inside)
    for b in rng.choice(B, size=min(B, 2), replace=False):
        traj = rb.mission(int(b))
        state, istate = cc.initial_state(traj[0, 0:3], Vc, hover=False)
        s_ref, _ = cc.rollout(traj, state, istate, K, Vc, aabbs=boxes)
        if np.isfinite(s_ref).all() and np.abs(s_ref[:, 10:13]).max() < 40.0:
            assert col_err(slog[:, :, int(b)].cpu().numpy(), s_ref) < 1e-7, (it, b)
            assert results[0][1][:, int(b)].cpu().tolist() == istate.tolist(), (it, b)
