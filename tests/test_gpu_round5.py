"""GPU tests of round 5: the self-check uavac_create runs (sampler heading == device library atan2), the device identity and the
clock probe behind the C ABI."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LAB_AABBS = np.array([[3.7, 4.3, 4.0, 10.0, -3.4, -2.8], [10.7, 11.3, 4.0, 10.0, -2.2, 0.0],
                      [13.3, 14.7, 6.3, 7.7, -6.0, 0.0], [20.2, 20.8, 4.0, 10.0, -3.3, -2.7]])


@pytest.fixture(scope="module")
def nat():
    from uav_ac import _native
    return _native


@pytest.fixture(scope="module")
def eng():
    from uav_ac.fleet import Engine
    return Engine("cuda:0")


def _missions(B, m):
    from oracle import minsnap_oracle as mo
    return mo.synthetic_missions(B, m)


def test_create_checks_the_heading_against_the_device_library(nat):
    """uavac_create runs the sampler's heading() and the device library's atan2 -- which the rollout's yaw scan calls -- over 2^16
    operand pairs and every special value and refuses a context when a bit differs (UAVAC_ETOOLCHAIN).  Here: it passes on this
    toolchain; the answer is a property of the build, so only the first context of a process asks (round-5 advice: no wall-clock
    assertion here)."""
    assert nat.ETOOLCHAIN == -7
    ctx = nat.Context(0)
    ident = ctx.device_identity()
    assert ident.startswith("uuid=") and ";pci=" in ident and "gfx950" in ident
    ctx.close()

    for _ in range(3):                                           # later contexts: created and destroyed without the check
        nat.Context(0).close()


def test_clock_probe_reads_a_plausible_shader_clock(eng):
    """uavac_clock_probe_dev: one wave stamps s_memtime / s_memrealtime a window apart; beside a busy GPU it reads between 1 and 2.6 GHz
    and its window is what was asked for."""
    import torch
    plan = eng.plan(_missions(16384, 8), 3.0, 0.01)
    fleet = eng.fleet(plan)
    side = torch.cuda.Stream(device=eng.device)
    probe = eng.clock_probe_begin(2000, stream=side)
    for _ in range(4):
        fleet.rollout(1000)
    torch.cuda.synchronize()
    c0, r0, c1, r1 = (int(v) for v in probe.cpu().tolist())
    assert 1.95e5 <= r1 - r0 <= 3.0e5                           # 2 ms at 100 MHz (plus the last sleep)
    assert 1.0 < eng.clock_probe_ghz(probe) < 2.6
    with pytest.raises(Exception):
        eng.clock_probe_begin(0)


@pytest.mark.parametrize("m", [1, 2, 3, 4, 7, 8, 9, 12, 20, 33])
def test_two_ended_solve_agrees_with_the_one_ended_solve_and_with_itself(eng, m):
    """The coefficient solve eliminates a mission's knots from BOTH ends with two lanes that meet at the middle knot (round 5,
    csrc/minsnap_solve_tw.hip; the tail lane runs the head's code on the time-reversed mission, J = diag(-1, 1, -1, -1) maps the knot
    unknowns back).  Another rounding than the one-ended kernel of rounds 1-4 (`solve_order` 0): the coefficients agree to 1e-12
    relative, both sit equally close to the reference formulation's dense pivoted solve, and every launch shape of the two-ended
    kernel gives the same bits -- missions per wave, blocks parked in HBM / LDS / registers -- for uniform and ragged batches
    (knot counts even and odd, m = 1 without any unknown, m = 2 with the junction alone); a ragged mission equals the mission alone."""
    import torch
    from oracle import minsnap_oracle as mo
    wps = mo.synthetic_missions(333, m)
    ragged = [w[: 2 + (i % m)] for i, w in enumerate(wps)]
    try:
        eng.ctx.set_option("solve_order", 0)
        p0 = eng.plan(wps, 3.0, 0.01)
        r0 = eng.plan_ragged(ragged, 3.0, 0.01)
        eng.ctx.set_option("solve_order", 1)
        got = {}
        for lanes, keep, park in ((64, 0, 0), (32, 0, 0), (16, 0, 0), (64, 1, 0), (-1, -1, -1), (64, 0, 1), (32, 0, 1)):
            eng.ctx.set_option("solve_lanes", lanes)
            eng.ctx.set_option("solve_keep", keep)
            eng.ctx.set_option("solve_park", park)
            p1 = eng.plan(wps, 3.0, 0.01)
            r1 = eng.plan_ragged(ragged, 3.0, 0.01)
            got[(lanes, keep, park)] = (p1.coeffs.clone(), r1.coeffs.clone(), p1.traj.clone(), r1.traj.clone())
    finally:
        for k, v in (("solve_order", 1), ("solve_lanes", -1), ("solve_keep", -1), ("solve_park", -1)):
            eng.ctx.set_option(k, v)
    base = got[(64, 0, 0)]
    for k, v in got.items():
        for x, y in zip(base, v):
            assert torch.equal(x, y), k
    scale = p0.coeffs.abs().max().clamp(min=1.0)
    assert float((base[0] - p0.coeffs).abs().max() / scale) < 1e-12 and float((base[1] - r0.coeffs).abs().max() / scale) < 1e-12
    so = r1.seg_offsets.cpu().numpy()
    for i in (0, 1, m - 1, 100, 332):                       # a ragged mission == the same waypoints planned alone
        k = len(ragged[i]) - 1
        alone = eng.plan(np.stack([ragged[i]] * 3), 3.0, 0.01)
        assert torch.equal(alone.coeffs[0], r1.coeffs[so[i]:so[i] + k].reshape(8 * k, 3)), i
    for b in (7, 200):                                      # SURVEY 8(c) metric against the reference formulation (NumPy oracle, dense solve)
        ref = mo.plan(wps[b], 3.0, 0.01, method="solve")
        for plan in (p0, p1):
            err = float(np.max(np.max(np.abs(plan.mission(b) - ref), axis=0) / np.maximum(1.0, np.max(np.abs(ref), axis=0))))
            assert err < 1e-8, (b, err)


@pytest.mark.parametrize("B", [1000, 20480, 35000])
def test_workgroups_per_cu_cap_changes_no_result(eng, B):
    """Below a full chip the launcher sizes a logged rollout's LDS so that no CU takes more workgroups than its even share (option
    "cu_balance", 55 / 41 KB of dynamic LDS for two / three workgroups per CU): a matter of where workgroups run, never
    of what they compute -- state log, command log, final state and flags are the same bit for bit with the cap on and off,
    plan-fed and row-fed."""
    import torch
    plan = eng.plan(_missions(B, 4), 3.0, 0.01)
    K = 130
    out = {}
    try:
        for cap in (0, 1):
            eng.ctx.set_option("cu_balance", cap)
            for feed in (True, False):
                f = eng.fleet(plan, from_plan=feed)
                s, c = f.rollout(K, state_log=True, cmd_log=True, aabbs=LAB_AABBS)
                out[(cap, feed)] = (s.clone(), c.clone(), f.state[:26].clone(), f.istate.clone())
    finally:
        eng.ctx.set_option("cu_balance", 1)
    for feed in (True, False):
        for x, y in zip(out[(0, feed)], out[(1, feed)]):
            assert torch.equal(x, y), feed
    for x, y in zip(out[(1, True)][:2], out[(1, False)][:2]):
        assert torch.equal(x, y)


def test_solve_order_is_zero_or_one(eng, nat):
    with pytest.raises(nat.UavacError):
        eng.ctx.set_option("solve_order", 2)
    eng.ctx.set_option("solve_order", 1)
