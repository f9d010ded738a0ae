"""Pin the C oracle (oracle/uavac_oracle.c; the timed CPU baseline) to the reference's golden vectors."""
import numpy as np
import pytest

from conftest import col_err, load_golden
from oracle import c_oracle as cc

TOL = 1e-5


@pytest.mark.parametrize("m", [1, 2, 8, 12, 20])
def test_c_planner_matches_reference_golden(m):
    g = load_golden("synthetic_missions.npz")
    wps = g[f"m{m}_wp"]
    sub = []
    for i, wp in enumerate(wps):
        traj, coeffs, times = cc.plan(wp, 3.0, 0.01)
        assert np.array_equal(times, g[f"m{m}_times"][i])          # bit for bit (FMA dot product of np.linalg.norm)
        assert np.array_equal(np.bincount(traj[:, 10].astype(int), minlength=m), g[f"m{m}_rows_per_segment"][i])
        assert col_err(coeffs, g[f"m{m}_coeffs_solve"][i]) < 1e-9
        assert col_err(coeffs, g[f"m{m}_coeffs_lstsq"][i]) < TOL
        sub.append(traj[::16])
        if f"m{m}_traj{i}" in g:
            assert col_err(traj, g[f"m{m}_traj{i}"]) < TOL
    assert col_err(np.vstack(sub), g[f"m{m}_traj_every16"]) < TOL


def test_c_planner_config1_and_yaw():
    g = load_golden("fixed_missions.npz")
    traj, coeffs, _ = cc.plan(g["config1_wp"], 3.0, 0.01)
    assert traj.shape == (687, 11)
    assert col_err(traj, g["config1_traj"]) < TOL
    t0, _, _ = cc.plan(g["lab_wp"][:2], 3.0, 0.01)
    t1, _, _ = cc.plan(g["lab_wp"][1:], 3.0, 0.01)
    assert col_err(np.vstack((t0, t1)), g["lab_traj_free"]) < TOL


@pytest.mark.parametrize("name, K", [("config1", 8000), ("lab_v2", 17000)])
def test_c_closed_loop_matches_reference_golden(name, K):
    g = load_golden("closed_loop.npz")
    traj = g[name + "_traj"]
    state, istate = cc.initial_state(traj[0, 0:3])
    slog, clog = cc.rollout(traj, state, istate, K)
    assert col_err(slog[:200], g[name + "_state_first200"]) < 1e-10
    assert col_err(clog[:200], g[name + "_cmd_first200"]) < 1e-10
    assert col_err(slog[9::10], g[name + "_state_every10"]) < TOL
    assert col_err(clog[9::10], g[name + "_cmd_every10"]) < TOL
    assert istate[0] == len(traj) - 1 and istate[1] == K


def test_c_and_python_oracles_agree_with_aabb_flag():
    from oracle import control_oracle as co
    from oracle import minsnap_oracle as mo
    aabbs = load_golden("fixed_missions.npz")["lab_aabbs"]
    wp = mo.synthetic_missions(4, 8)[3]
    traj = mo.plan(wp, 3.0, 0.01, method="solve")
    tc, _, _ = cc.plan(wp, 3.0, 0.01)
    assert col_err(tc, traj) < 1e-9
    u = co.UAV(co.Vehicle(), position=traj[0, 0:3])
    s_py, c_py = co.rollout(u, traj, 1500, aabbs=aabbs)
    state, istate = cc.initial_state(traj[0, 0:3])
    s_c, c_c = cc.rollout(traj, state, istate, 1500, aabbs=aabbs)
    assert col_err(s_c, s_py) < 1e-9 and col_err(c_c, c_py) < 1e-9
    assert istate[2] == u.collided and istate[0] == u.traj_index


def test_cpu_baseline_leg_runs_bounded():
    from oracle import cpu_baseline as cb
    r = cb.run(segments=8, ticks=500, velocity=3.0, dt=0.01, budget_s=1.0, max_missions=3)
    assert r["kind"] == "port" and r["cores"] == 1 and r["value"] > 0 and r["unit"] == "UAV control-steps/s"
