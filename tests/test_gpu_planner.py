"""GPU parity of the HIP planner (K0 row counts, K1 solve, K2 sampler + yaw scan) against the
reference's golden vectors and the CPU oracle.  Everything goes through the C ABI (ctypes)."""

import numpy as np
import pytest

from conftest import col_err, load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-5      # north_star: <= 1e-5 relative on trajectory samples (SURVEY.md 8(c) metric)


@pytest.fixture(scope="module")
def eng():
    from uav_ac.fleet import Engine
    return Engine("cuda:0")


@pytest.fixture(scope="module")
def ctx():
    from uav_ac import _native as nat
    return nat.Context(0)


def host_plan(ctx, wps, velocity, dt):
    """Plan through the host-pointer twins of the C ABI.  Returns coeffs, times, seg_rows, offsets, traj."""
    from uav_ac import _native as nat
    wps = nat.as_f64(wps)
    B, m = wps.shape[0], wps.shape[1] - 1
    times = np.empty((B, m))
    seg_rows = np.empty((B, m), dtype=np.int32)
    offs = np.empty(B + 1, dtype=np.int64)
    ctx.call("uavac_minsnap_row_counts", nat.np_ptr(wps), B, m, velocity, dt, nat.np_ptr(times), nat.np_ptr(seg_rows),
             nat.np_ptr(offs))
    coeffs = np.empty((B, 8 * m, 3))
    times2 = np.empty((B, m))
    ctx.call("uavac_minsnap_solve", nat.np_ptr(wps), B, m, velocity, nat.np_ptr(coeffs), nat.np_ptr(times2))
    assert np.array_equal(times, times2)
    traj = np.empty((int(offs[-1]), 11))
    ctx.call("uavac_minsnap_sample", nat.np_ptr(coeffs), nat.np_ptr(times), B, m, dt, nat.np_ptr(offs), nat.np_ptr(traj))
    return coeffs, times, seg_rows, offs, traj


@pytest.mark.parametrize("m", [2, 5, 8, 11, 12])
def test_solve_parked_in_lds_equals_solve_parked_in_hbm(eng, m):
    """Option "solve_park": the block-Thomas forward sweep keeps [Ut | rt] in the wave's LDS when (m - 1) x 14 KB fit (m <= 11)
    instead of the HBM workspace -- same arithmetic, same coefficients bit for bit, uniform and ragged batches; m = 12 does not
    fit and silently stays in HBM."""
    import torch
    from oracle import minsnap_oracle as mo
    wps = mo.synthetic_missions(200, m)
    ragged = [w[: 2 + (i % m)] for i, w in enumerate(wps)]
    got = {}
    try:
        for park in (0, 1):
            eng.ctx.set_option("solve_park", park)
            plan = eng.plan(wps, 3.0, 0.01)
            rb = eng.plan_ragged(ragged, 3.0, 0.01)
            got[park] = (plan.coeffs.clone(), plan.traj.clone(), rb.coeffs.clone(), rb.traj.clone())
    finally:
        eng.ctx.set_option("solve_park", -1)
    for x, y in zip(got[0], got[1]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("m", [1, 2, 3, 6, 7, 12, 20])
def test_solve_launch_shapes_give_the_same_coefficients_bit_for_bit(eng, m):
    """Options "solve_lanes" (64 / 32 / 16 lanes of a wave carry a mission: more waves for the same batch) and "solve_keep" (the
    first five knots' blocks of the forward sweep stay in registers instead of the workspace) choose how the solve is launched,
    never what it computes: uniform and ragged batches, sizes that are no multiple of 64, 32 or 16 lanes, fewer and more knots
    than the five that are kept."""
    import torch
    from oracle import minsnap_oracle as mo
    wps = mo.synthetic_missions(333, m)
    ragged = [w[: 2 + (i % m)] for i, w in enumerate(wps)]
    got = {}
    try:
        eng.ctx.set_option("solve_park", 0)
        for lanes, keep in ((64, 0), (32, 0), (16, 0), (64, 1), (-1, -1)):
            eng.ctx.set_option("solve_lanes", lanes)
            eng.ctx.set_option("solve_keep", keep)
            plan = eng.plan(wps, 3.0, 0.01)
            rb = eng.plan_ragged(ragged, 3.0, 0.01)
            assert torch.isfinite(plan.coeffs).all()
            got[(lanes, keep)] = (plan.coeffs.clone(), plan.traj.clone(), rb.coeffs.clone(), rb.traj.clone())
    finally:
        eng.ctx.set_option("solve_park", -1)
        eng.ctx.set_option("solve_lanes", -1)
        eng.ctx.set_option("solve_keep", -1)
    for k, v in got.items():
        for x, y in zip(got[(64, 0)], v):
            assert torch.equal(x, y), k


@pytest.mark.parametrize("B", [3, 9, 40])
def test_host_twins_move_one_piece_several_pieces_and_more_than_a_megabyte(ctx, eng, B):
    """The host-pointer twins stage pageable buffers through a pinned ping-pong buffer in 256 KiB pieces (two in flight) up to
    1 MiB and hand larger ones to the runtime: a row buffer of one piece (B = 3: 230 KB), of three or four pieces (B = 9:
    ~700 KB) and of more than 1 MiB (B = 40: 3 MB) must each equal the device path bit for bit (round-3 VERDICT 'weak' 8)."""
    from oracle import minsnap_oracle as mo
    wps = mo.synthetic_missions(B, 8)
    coeffs, times, seg_rows, offs, traj = host_plan(ctx, wps, 3.0, 0.01)
    plan = eng.plan(wps, 3.0, 0.01)
    nbytes = traj.nbytes
    assert (B == 3 and nbytes <= (256 << 10)) or (B == 9 and (512 << 10) < nbytes <= (1 << 20)) or (B == 40 and nbytes > (1 << 20))
    assert np.array_equal(traj, plan.traj.cpu().numpy()) and np.array_equal(coeffs, plan.coeffs.cpu().numpy())
    assert np.array_equal(offs, plan.row_offsets.cpu().numpy())


@pytest.mark.parametrize("m", [1, 2, 8, 12, 20])
def test_synthetic_missions_match_reference_golden(ctx, m):
    g = load_golden("synthetic_missions.npz")
    wps = g[f"m{m}_wp"]
    coeffs, times, seg_rows, offs, traj = host_plan(ctx, wps, 3.0, 0.01)
    assert np.array_equal(times, g[f"m{m}_times"])             # bit for bit: the FMA dot product of np.linalg.norm
    assert np.array_equal(seg_rows, g[f"m{m}_rows_per_segment"])          # row counts exact
    assert col_err(coeffs.reshape(-1, 3), g[f"m{m}_coeffs_lstsq"].reshape(-1, 3)) < TOL
    assert col_err(coeffs.reshape(-1, 3), g[f"m{m}_coeffs_solve"].reshape(-1, 3)) < 1e-9
    sub = []
    for b in range(len(wps)):
        tr = traj[offs[b]:offs[b + 1]]
        sub.append(tr[::16])
        key = f"m{m}_traj{b}"
        if key in g:
            assert tr.shape == g[key].shape
            assert np.array_equal(tr[:, 10], g[key][:, 10])               # spline ids exact
            assert col_err(tr, g[key]) < TOL
    assert col_err(np.vstack(sub), g[f"m{m}_traj_every16"]) < TOL


def test_config1_and_lab_missions_match_reference_golden(ctx):
    g = load_golden("fixed_missions.npz")
    coeffs, times, _, offs, traj = host_plan(ctx, g["config1_wp"][None], 3.0, 0.01)
    assert np.array_equal(times[0], g["config1_times"])
    assert traj.shape == (687, 11)
    assert col_err(coeffs[0], g["config1_coeffs"]) < TOL
    assert col_err(traj, g["config1_traj"]) < TOL
    # lab mission without obstacles = takeoff (1 spline) + course (7 splines), main.py:73-84
    lab = g["lab_wp"]
    _, _, _, _, t0 = host_plan(ctx, lab[None, :2], 3.0, 0.01)
    _, _, _, _, t1 = host_plan(ctx, lab[None, 1:], 3.0, 0.01)
    both = np.vstack((t0, t1))
    assert both.shape == g["lab_traj_free"].shape == (1076, 11)
    assert col_err(both, g["lab_traj_free"]) < TOL
    assert np.all(t0[:, 9] == 0.0)            # vertical takeoff: no valid heading -> zeros


def test_device_pointer_path_equals_host_pointer_path(eng, ctx):
    from oracle.minsnap_oracle import synthetic_missions
    wps = synthetic_missions(96, 8)
    plan = eng.plan(wps, 3.0, 0.01)
    eng.check(plan)
    coeffs, times, seg_rows, offs, traj = host_plan(ctx, wps, 3.0, 0.01)
    assert np.array_equal(plan.row_offsets.cpu().numpy(), offs)
    assert np.array_equal(plan.seg_rows.cpu().numpy(), seg_rows)
    assert np.array_equal(plan.coeffs.cpu().numpy(), coeffs)          # deterministic: bit-identical
    assert np.array_equal(plan.traj.cpu().numpy(), traj)
    # a second run into the same buffers is bit-identical too
    before = plan.traj.clone()
    eng.solve(plan)
    eng.sample(plan)
    assert bool((plan.traj == before).all())


@pytest.mark.parametrize("m, B", [(8, 512), (12, 256), (20, 128)])
def test_against_oracle_on_fresh_missions(eng, m, B):
    """Missions not in the golden set: HIP vs the oracle's exact (`solve`) KKT path."""
    from oracle import minsnap_oracle as mo
    wps = mo.synthetic_missions(B, m)[-8:]
    plan = eng.plan(wps, 3.0, 0.01)
    for b in range(len(wps)):
        ref = mo.plan(wps[b], 3.0, 0.01, method="solve")
        got = plan.mission(b)
        assert got.shape == ref.shape
        assert col_err(got, ref) < 1e-8


def test_stress_distribution_vs_exact_oracle(eng):
    """segment lengths U(1,6) m -- compare with the oracle's LU (`solve`) path, where the
    reference's own lstsq is no longer accurate to 1e-5."""
    from oracle import minsnap_oracle as mo
    wps = mo.synthetic_missions(24, 12, lo=1.0, hi=6.0)
    plan = eng.plan(wps, 3.0, 0.01)
    for b in range(0, 24, 3):
        ref = mo.plan(wps[b], 3.0, 0.01, method="solve")
        got = plan.mission(b)
        assert got.shape == ref.shape
        assert col_err(got, ref) < 1e-6


def _yaw_via_sampler(ctx, vel):
    """Feed crafted velocities through the HIP yaw scan: one 1-row segment per sample, c1 = velocity."""
    from uav_ac import _native as nat
    n = len(vel)
    coeffs = np.zeros((1, 8 * n, 3))
    coeffs[0, 1::8, :] = vel
    times = np.full((1, n), 0.005)
    offs = np.array([0, n], dtype=np.int64)
    traj = np.empty((n, 11))
    ctx.call("uavac_minsnap_sample", nat.np_ptr(coeffs), nat.np_ptr(times), 1, n, 0.01, nat.np_ptr(offs), nat.np_ptr(traj))
    assert np.array_equal(traj[:, 3:6], vel)
    return traj[:, 9]


@pytest.mark.parametrize("case", ["hold", "cross_pi", "none_valid", "exact_pi_steps", "leading_invalid"])
def test_yaw_scan_crafted_cases(ctx, case):
    g = load_golden("yaws.npz")
    vel = g[case + "_vel"]
    got = _yaw_via_sampler(ctx, vel)
    want = g[case + "_yaw"]
    if np.signbit(vel[vel == 0]).any():
        # a velocity of -0.0 cannot come out of a polynomial evaluation (0*t + -0.0 = +0.0), neither in the
        # reference's sampler nor here; the expected heading is the one of the +0.0 the sampler produces
        from oracle.minsnap_oracle import yaws_from_velocity
        want = yaws_from_velocity(vel + 0.0)
    assert np.allclose(got, want, rtol=0, atol=1e-12)


def test_yaw_scan_random_spin_windows(ctx):
    from oracle.minsnap_oracle import yaws_from_velocity
    vel = load_golden("yaws.npz")["random_spin_vel"]
    for lo in range(0, 400 - 64, 48):
        w = vel[lo:lo + 64]
        assert np.allclose(_yaw_via_sampler(ctx, w), yaws_from_velocity(w), rtol=0, atol=1e-11)


def test_yaw_scan_across_chunks_and_late_first_valid(eng):
    """A vertical first segment (> 256 rows: no usable heading for more than one chunk) followed by turns
    that wrap through +-pi: exercises the back-fill and the carry of the unwrap sum across chunks."""
    from oracle import minsnap_oracle as mo
    wp = np.array([[[0, 0, 0], [0, 0, -12], [-3, 0.2, -12], [-6, -0.2, -12], [-3, -3, -12], [0, 0, -12],
                    [-4, 0.1, -12], [-8, -0.1, -12]]], dtype=float)
    plan = eng.plan(wp, 3.0, 0.01)
    ref = mo.plan(wp[0], 3.0, 0.01, method="solve")
    got = plan.mission(0)
    assert got.shape == ref.shape and int(plan.seg_rows[0, 0]) > 256
    assert np.ptp(ref[:, 9]) > np.pi                       # the unwrap really accumulates
    assert np.allclose(got[:, 9], ref[:, 9], rtol=0, atol=1e-9)
    assert col_err(got, ref) < 1e-8


def test_yaw_backfill_when_first_heading_comes_after_several_chunks(ctx):
    """Crafted coefficients through the sampler: 400 rows of purely vertical motion (no usable heading for
    more than one 256-row chunk), then 900 rows whose heading winds several times: back-fill, hold and
    the unwrap carry across chunks, against the oracle's sampler + yaw scan."""
    from uav_ac import _native as nat
    from oracle import minsnap_oracle as mo
    coeffs = np.zeros((1, 24, 3))
    coeffs[0, 1, 2] = -1.0                                     # segment 0: z' = -1, x' = y' = 0 exactly
    t = np.arange(8)
    coeffs[0, 8:16, 0] = [0, -1.0, 0.9, 0.4, -0.31, 0.05, -0.002, 0]      # segment 1: winding heading
    coeffs[0, 8:16, 1] = [0, 0.02, -1.2, 0.5, 0.11, -0.06, 0.006, -0.0002]
    coeffs[0, 16:24, 0] = [1, 1e-4, 0, 0, 0, 0, 0, 0]                      # segment 2: below the speed threshold
    times = np.array([[4.0, 6.0, 3.0]])
    offs = np.array([0, 400 + 600 + 300], dtype=np.int64)
    traj = np.empty((1300, 11))
    ctx.call("uavac_minsnap_sample", nat.np_ptr(coeffs), nat.np_ptr(times), 1, 3, 0.01, nat.np_ptr(offs), nat.np_ptr(traj))
    pos, vel, acc, sid = mo.sample(coeffs[0], times[0], 0.01)
    yaw = mo.yaws_from_velocity(vel)
    assert np.ptp(yaw) > 2 * np.pi and np.all(yaw[:400] == yaw[400])   # winds; leading rows back-filled
    assert np.allclose(traj[:, 9], yaw, rtol=0, atol=1e-10)
    assert np.all(traj[1000:, 9] == traj[999, 9])                           # held through the slow tail
    assert col_err(traj[:, 0:9], np.hstack((pos, vel, acc))) < 1e-12
    assert np.array_equal(traj[:, 10], sid)


def test_edge_shapes_and_errors(ctx, eng):
    from uav_ac import _native as nat
    # ragged batch: missions of very different lengths share one trajectory buffer
    wps = np.array([[[0, 0, 0], [0.3, 0, 0], [0.6, 0.1, 0]], [[0, 0, -1], [30, 0, -1], [60, 5, -2]]], dtype=float)
    coeffs, times, seg_rows, offs, traj = host_plan(ctx, wps, 2.0, 0.01)
    assert seg_rows[0].sum() == offs[1] and offs[2] - offs[1] == seg_rows[1].sum() and seg_rows[0].sum() < 60
    from oracle import minsnap_oracle as mo
    for b in range(2):
        ref = mo.plan(wps[b], 2.0, 0.01, method="solve")
        assert col_err(traj[offs[b]:offs[b + 1]], ref) < 1e-6
    # invalid input -> error codes, like the adapter's ValueError (mujoco_sim.py:37-38)
    bad = wps.copy()
    bad[1, 1, 0] = np.nan
    with pytest.raises(nat.UavacError) as e:
        host_plan(ctx, bad, 2.0, 0.01)
    assert e.value.code == nat.ENONFINITE
    with pytest.raises(ValueError):
        eng.plan(bad, 2.0, 0.01)
    rep = wps.copy()
    rep[0, 1] = rep[0, 0]                                   # repeated waypoint: T = 0 -> singular
    with pytest.raises(nat.UavacError) as e:
        host_plan(ctx, rep, 2.0, 0.01)
    assert e.value.code in (nat.ESINGULAR, nat.ENONFINITE)
    with pytest.raises(nat.UavacError) as e:
        ctx.call("uavac_minsnap_solve", nat.np_ptr(wps), 2, 0, 2.0, nat.np_ptr(coeffs), None)
    assert e.value.code == nat.EINVAL
    # maximum segment count
    wmax = mo.synthetic_missions(2, nat.MAX_SEGMENTS)
    plan = eng.plan(wmax, 3.0, 0.01)
    eng.check(plan)
    assert col_err(plan.mission(1), mo.plan(wmax[1], 3.0, 0.01, method="solve")) < 1e-6


def test_full_size_properties_config2(eng):
    """BASELINE config 2 size (B = 4096, m = 8): size-independent properties instead of a full oracle run."""
    import torch
    from oracle import minsnap_oracle as mo
    B, m = 4096, 8
    wps = mo.synthetic_missions(B, m)
    plan = eng.plan(wps, 3.0, 0.01)
    eng.check(plan)
    traj, offs, seg = plan.traj, plan.row_offsets, plan.seg_rows.to(torch.int64)
    assert plan.total_rows == int(seg.sum()) and bool(torch.isfinite(traj).all())
    # first row of every segment sits on its waypoint with t = 0 (position = c0 exactly)
    seg_start = offs[:-1, None] + torch.cumsum(seg, 1) - seg
    wp_t = torch.as_tensor(wps, device=traj.device)
    assert bool((traj[seg_start.reshape(-1), 0:3].reshape(B, m, 3) == wp_t[:, :m]).all())
    # missions start and end at rest (v = a = 0 at t = 0; tiny at the last sample)
    assert float(traj[offs[:-1], 3:9].abs().max()) == 0.0
    assert float(traj[offs[1:] - 1, 3:6].abs().max()) < 1e-2
    # velocity is continuous across every knot (reference test_minimum_snap.py:139-151 bound 0.5)
    dv = (traj[1:, 3:6] - traj[:-1, 3:6]).norm(dim=1)
    same = torch.ones(plan.total_rows - 1, dtype=torch.bool, device=traj.device)
    same[offs[1:-1] - 1] = False
    assert float(dv[same].max()) < 0.5
    # spline ids are non-decreasing inside a mission and end at m-1
    assert bool((traj[offs[1:] - 1, 10] == m - 1).all())
    # spot-check 4 missions against the oracle
    for b in (0, 1365, 2730, 4095):
        assert col_err(plan.mission(b), mo.plan(wps[b], 3.0, 0.01, method="solve")) < 1e-8


@pytest.mark.parametrize("m, lo, hi, tol", [(1, 2.5, 3.5, 0.0), (2, 2.5, 3.5, 1e-11), (12, 2.5, 3.5, 1e-10),
                                           (20, 1.0, 6.0, 1e-8), (64, 2.5, 3.5, 1e-9)])
def test_two_device_solvers_agree(eng, m, lo, hi, tol):
    """The lane-per-mission block-Thomas solver (the one the API uses) against the independent
    wave-per-mission pivoted banded LU, on batches that are not a multiple of the wave size."""
    import ctypes
    import torch
    from oracle import minsnap_oracle as mo
    B = 203
    wps = mo.synthetic_missions(B, m, lo, hi)
    plan = eng.plan(wps, 3.0, 0.01)
    eng.check(plan)
    other = torch.empty_like(plan.coeffs)
    status = torch.ones((B,), dtype=torch.int32, device=other.device)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    eng._bind_stream()
    eng.ctx.call("uavac_minsnap_solve_banded_dev", P(plan.waypoints), P(plan.times), B, m, P(other), P(status))
    torch.cuda.synchronize()
    assert int(status.sum()) == 0
    a, b = plan.coeffs.cpu().numpy().reshape(-1, 3), other.cpu().numpy().reshape(-1, 3)
    # compare as trajectories at the knots' scale: coefficients of high powers are large, so normalise per mission
    assert col_err(a, b) <= tol * max(1.0, np.abs(b).max())
    ref = mo.plan(wps[B - 1], 3.0, 0.01, method="solve")
    assert col_err(plan.mission(B - 1), ref) < 1e-6


def test_batched_obstacle_replan_matches_reference_and_oracle(eng):
    """SURVEY.md 8(f) N1, batched: ragged missions through the lab course's four AABBs at once.  Missions 0/1
    are pinned by the reference's goldens (lab course, in-tree obstacle case); the others by the oracle's
    restatement of the same loop.  The collision scan runs inside the sampler (uavac_minsnap_sample_hits_dev)."""
    from oracle import minsnap_oracle as mo
    g = load_golden("fixed_missions.npz")
    aabbs = g["lab_aabbs"]
    # perturbed lab courses that need one midpoint each (found with the oracle; both of its KKT paths agree on
    # them -- where inserted midpoints make splines very short the reference's lstsq is rounding noise)
    extra = [np.array(w) for w in (
        [[0.306, 7.421, -1.929], [4.879, 7.283, -1.507], [8.514, 4.336, -2.852], [10.092, 7.381, -2.899],
         [14.786, 9.099, -1.907], [18.137, 10.603, -3.359], [19.445, 6.362, -1.644], [23.193, 7.388, -2.316]],
        [[0.053, 8.505, -1.008], [4.813, 6.709, -2.036], [8.411, 2.449, -3.085], [10.596, 7.133, -3.169],
         [14.558, 8.486, -2.093], [15.624, 10.454, -3.402], [19.668, 6.598, -1.978], [22.966, 7.643, -1.673]],
        [[1.162, 6.984, -1.939], [2.316, 6.934, -1.926], [7.956, 3.78, -2.785], [10.762, 8.079, -3.215],
         [15.071, 10.071, -2.944], [15.98, 9.633, -3.08], [20.045, 8.744, -1.429], [23.327, 7.061, -2.537]],
        [[0.781, 7.554, -0.895], [4.809, 6.475, -1.561], [7.306, 2.632, -2.75], [9.763, 6.745, -3.247],
         [14.376, 9.532, -2.248], [17.526, 8.708, -3.414], [20.748, 7.1, -1.595], [22.743, 8.334, -2.255]])]
    extra += [g["lab_wp"][1:] + np.array([0.0, 0.2, -0.1]), g["lab_wp"][1:6] + np.array([0.1, -0.1, 0.0])]
    missions = [g["lab_wp"][1:], g["lab_wp"][:2]] + extra
    rp = eng.plan_collision_free(missions, aabbs, 3.0, 0.01)
    assert rp.B == len(missions) and int(rp.row_offsets[-1]) == rp.total_rows
    assert np.array_equal(rp.final_waypoints[0], g["lab_course_final_wp"])
    ref_obs = g["lab_traj_obs"]                            # takeoff rows, then the course rows
    n_takeoff = len(rp.mission(1))
    assert col_err(rp.mission(1), ref_obs[:n_takeoff]) < TOL
    assert col_err(rp.mission(0), ref_obs[n_takeoff:]) < TOL
    grew = 0
    for b, wp in enumerate(missions):
        traj, final_wp = mo.plan_collision_free(wp, aabbs, 3.0, 0.01, method="solve")
        assert np.array_equal(rp.final_waypoints[b], final_wp)
        assert col_err(rp.mission(b), traj) < 1e-6
        grew += len(final_wp) > len(wp)
        p = rp.mission(b)[:, :3]
        c = aabbs[-1]                                      # the last cuboid is guaranteed clean (earlier ones are not re-checked)
        assert not np.any((p[:, 0] >= c[0]) & (p[:, 0] <= c[1]) & (p[:, 1] >= c[2]) & (p[:, 1] <= c[3]) &
                          (p[:, 2] >= c[4]) & (p[:, 2] <= c[5]))
    assert grew >= 4                                        # the case really inserts midpoints
    # the in-tree obstacle case of the reference (tests/unit/planning/test_minimum_snap.py:171-183)
    one = eng.plan_collision_free([g["obs_case_wp"]], g["obs_case_aabb"], 2.0, 0.01)
    assert np.array_equal(one.final_waypoints[0], g["obs_case_final_wp"])
    assert col_err(one.mission(0), g["obs_case_traj"]) < TOL
    # a ragged plan flies like any other
    fleet = eng.fleet(rp)
    fleet.rollout(500)
    assert bool((fleet.trajectory_index == 50).all())
    assert rp.converged.all()
    # an ill-posed mission (its straight leg crosses a bar squarely: midpoints never leave it) ends the strict call
    # but only marks itself in the lenient one; the others come out as before
    bar = np.array([[4.0, 5.0, -10.0, 10.0, -10.0, 10.0]])
    through = np.array([[0.0, 0.0, -2.0], [9.0, 0.0, -2.0], [9.0, 5.0, -2.0]])
    around = np.array([[0.0, 12.0, -2.0], [3.0, 14.0, -2.0], [9.0, 14.0, -2.5]])
    with pytest.raises(RuntimeError):
        eng.plan_collision_free([around, through], bar, 2.0, 0.01)
    lenient = eng.plan_collision_free([around, through], bar, 2.0, 0.01, strict=False)
    assert list(lenient.converged) == [True, False]
    assert col_err(lenient.mission(0), mo.plan(around, 2.0, 0.01, method="solve")) < 1e-7
    assert len(lenient.final_waypoints[1]) - 1 <= 64 and len(lenient.mission(1)) > 0


def test_recheck_passes_leave_no_row_inside_any_obstacle(eng):
    """Beyond the reference (which never re-checks an earlier obstacle): with recheck_passes the final rows of every
    converged mission are outside EVERY cuboid; without it the same missions show the reference's leftover conflicts
    or none -- never more rows than the re-checked plan fixes."""
    g = load_golden("fixed_missions.npz")
    aabbs = g["lab_aabbs"] if "lab_aabbs" in g.files else None
    if aabbs is None:
        aabbs = np.array([[3.7, 4.3, 4.0, 10.0, -3.4, -2.8], [10.7, 11.3, 4.0, 10.0, -2.2, 0.0],
                          [13.3, 14.7, 6.3, 7.7, -6.0, 0.0], [20.2, 20.8, 4.0, 10.0, -3.3, -2.7]])
    # long-legged courses that hug obstacle corners: thinned RRT* paths across the lab (deterministic seeds)
    from uav_ac.planning.rrt import RRTStar, draw_random_nodes_batch
    rng = np.random.default_rng(31)
    n = 128
    lw, up = np.array([0.0, 0.0, -6.0]), np.array([24.0, 14.0, 0.0])
    starts = np.round(np.array([1.0, 7.0, -1.3]) + rng.uniform(-0.5, 0.5, (n, 3)) * [1, 4, 0.5], 2)
    goals = np.round(np.array([23.0, 7.0, -2.0]) + rng.uniform(-0.5, 0.5, (n, 3)) * [1, 4, 0.5], 2)
    found = eng.rrt_star(starts, goals, 1.5, draw_random_nodes_batch(np.arange(n), lw, up, goals, 1200), aabbs).to_host()
    thin = RRTStar(np.stack([lw, up]), starts[0], goals[0], 1.5, 1, aabbs)
    missions = [thin.simplify_path(found.path(b)) for b in np.flatnonzero(found.status == 0)]
    assert len(missions) >= 100

    def rows_inside(rp):
        bad = np.zeros(rp.B, dtype=int)
        for b in range(rp.B):
            p = rp.mission(b)[:, :3]
            for c in aabbs:
                bad[b] += int(np.sum((p[:, 0] >= c[0]) & (p[:, 0] <= c[1]) & (p[:, 1] >= c[2]) & (p[:, 1] <= c[3]) &
                                     (p[:, 2] >= c[4]) & (p[:, 2] <= c[5])))
        return bad
    once = eng.plan_collision_free(missions, aabbs, 2.0, 0.01, strict=False)
    again = eng.plan_collision_free(missions, aabbs, 2.0, 0.01, strict=False, recheck_passes=6)
    left_once, left_again = rows_inside(once), rows_inside(again)
    assert again.converged.sum() >= 0.6 * len(missions)
    assert (left_once > 0).sum() >= 1                       # the reference's single pass does leave conflicts here
    assert not left_again[again.converged].any()
    assert left_once.sum() >= left_again.sum()
    same = [b for b in range(len(missions)) if left_once[b] == 0 and once.converged[b]]
    for b in same[:20]:                                     # nothing to fix => the extra passes change nothing
        assert np.array_equal(once.final_waypoints[b], again.final_waypoints[b])


def test_randomised_shapes_velocities_and_steps(eng):
    """40 random (B, m, leg lengths, velocity, dt) draws against the oracle's exact KKT path: row counts and
    spline ids exact, samples <= 1e-7."""
    from oracle import minsnap_oracle as mo
    rng = np.random.default_rng(2026)
    for _ in range(40):
        B, m = int(rng.integers(1, 6)), int(rng.integers(1, 11))
        lo = float(rng.uniform(0.8, 3.0)); hi = lo + float(rng.uniform(0.2, 3.0))
        vel, dt = float(rng.uniform(0.5, 4.0)), float(rng.choice([0.005, 0.01, 0.02, 0.013]))
        wps = mo.synthetic_missions(B, m, lo, hi) + rng.normal(0, 0.3, (B, m + 1, 3))
        plan = eng.plan(wps, vel, dt)
        eng.check(plan)
        for b in range(B):
            ref = mo.plan(wps[b], vel, dt, method="solve")
            got = plan.mission(b)
            assert got.shape == ref.shape, (B, m, vel, dt)
            assert np.array_equal(got[:, 10], ref[:, 10])
            assert col_err(got, ref) < 1e-7, (B, m, lo, hi, vel, dt)


def test_dense_yaw_column_equals_the_rows_yaw(eng):
    """Plan.yaw (written in groups of eight chunks, with the back-fill of leading rows patched in LDS or in HBM
    depending on whether the group had left) is column 9 of the rows, bit for bit -- including missions whose first
    usable heading comes after several chunks and missions shorter than one group."""
    import torch
    from oracle import minsnap_oracle as mo
    rng = np.random.default_rng(8)
    for B, m, vel, dt in ((200, 12, 3.0, 0.01), (64, 1, 3.0, 0.01), (77, 3, 0.4, 0.002), (50, 20, 6.0, 0.05)):
        wps = mo.synthetic_missions(B, m)
        wps[::3, 1:3, 0:2] = wps[::3, 0:1, 0:2]             # first legs vertical: no heading for hundreds of rows
        plan = eng.plan(wps + rng.normal(0, 1e-9, wps.shape) * 0, vel, dt, dense_yaw=True)
        assert torch.equal(plan.yaw, plan.traj[:, 9])
        eng.sample(plan)
        assert torch.equal(plan.yaw, plan.traj[:, 9])
        # the missions' first headings: what the rows before the first usable heading hold, i.e. row 0's yaw
        assert torch.equal(plan.first_yaw, plan.traj[plan.row_offsets[:-1], 9])


def test_sampler_heading_equals_the_device_library_atan2_bit_for_bit(eng):
    """csrc/minsnap_yaw.h heading() is the device library's atan2 with the instruction count cut (three-source fmas instead of
    fmac + a move per coefficient, no special-case tests): the same operations in the same order, so the same BITS -- over 2^24
    random operand pairs of every magnitude and every pairing of special values (zeros of both signs, infinities, NaNs, denormals,
    equal magnitudes, the axes)."""
    import ctypes as C
    import torch
    g = torch.Generator(device="cpu").manual_seed(11)
    n = 1 << 24
    mant = torch.rand(2, n, generator=g, dtype=torch.float64) * 2 - 1
    expo = torch.randint(-40, 40, (2, n), generator=g)
    expo[:, : n // 4] = torch.randint(-1000, 1000, (2, n // 4), generator=g)         # a quarter over the whole exponent range
    expo[:, n // 4: n // 2] = torch.randint(-3, 4, (2, n // 4), generator=g)         # a quarter at velocity-like magnitudes
    v = torch.ldexp(mant, expo)
    specials = torch.tensor([0.0, -0.0, 1.0, -1.0, float("inf"), float("-inf"), float("nan"), 5e-324, -5e-324, 2.2250738585072014e-308,
                             1.7976931348623157e308, -1.7976931348623157e308, 1e-3, -1e-3, 3.0, -3.0, 0.5, 2.0 ** -600, 2.0 ** 600], dtype=torch.float64)
    yy, xx = torch.meshgrid(specials, specials, indexing="ij")
    y = torch.cat([v[0], yy.reshape(-1), v[0, :4096], v[0, :4096]]).to(eng.device)
    x = torch.cat([v[1], xx.reshape(-1), v[0, :4096], -v[0, :4096]]).to(eng.device)
    a, b = torch.empty_like(y), torch.empty_like(y)
    eng.ctx.call("uavac_probe_heading_dev", y.data_ptr(), x.data_ptr(), C.c_int64(y.numel()), a.data_ptr(), b.data_ptr())
    torch.cuda.synchronize()
    same = (a.view(torch.int64) == b.view(torch.int64)) | (torch.isnan(a) & torch.isnan(b))
    assert bool(same.all()), (y[~same][:4].tolist(), x[~same][:4].tolist(), a[~same][:4].tolist(), b[~same][:4].tolist())
    ref = torch.atan2(y.cpu(), x.cpu())
    ok = torch.isfinite(ref)
    assert float((a.cpu()[ok] - ref[ok]).abs().max()) < 1e-14


def test_heading_threshold_is_numpys_unfused_sum_of_squares(ctx):
    """|v_xy| >= 1e-3 decides whether a row has a heading of its own (minimum_snap.py:128-129, np.linalg.norm = sqrt of a
    rounded sum of rounded squares).  Velocities for which a fused multiply-add in vx^2 + vy^2 lands on the other side of the
    threshold than NumPy's two rounded products (found by exact rational arithmetic): the row must take NumPy's side."""
    from oracle.minsnap_oracle import yaws_from_velocity
    pairs = [("0x1.c979106bbda29p-11", "0x1.001e375da1c15p-11"), ("0x1.11056ffcadd96p-11", "0x1.bf972ef14728bp-11"),
             ("-0x1.f64a69e912dcfp-11", "0x1.2c8bc98d6f686p-12"), ("-0x1.ce85b735d7881p-11", "0x1.edc3e06b5604fp-12"),
             ("-0x1.c40a0e4d6cf16p-11", "0x1.0997351d95b1bp-11"), ("0x1.7a6a13ec6710cp-11", "0x1.6ae0c76ec1f2cp-11"),
             ("-0x1.f68ad299fcf30p-11", "0x1.2adbe40a08995p-12"), ("0x1.1556a9e8e62dep-11", "0x1.bceda915709e0p-11"),
             ("-0x1.9b28de405ba7dp-11", "0x1.454efd7f47760p-11"), ("-0x1.2431c52b6b0dap-13", "0x1.03964ae034407p-10"),
             ("-0x1.9b842123f91b6p-13", "0x1.010c2d80d11fbp-10"), ("0x1.ef32ec9f4beafp-11", "0x1.586affb58c88ep-12")]
    rows = [[1.0, 0.25, 0.0]]
    for i, (a, b) in enumerate(pairs):
        rows += [[float.fromhex(a), float.fromhex(b), 0.1], [float.fromhex(b), float.fromhex(a), 0.0], [np.cos(0.4 * i), -np.sin(0.4 * i), 0.0]]
    vel = np.array(rows)
    h = np.sqrt(vel[:, 0] ** 2 + vel[:, 1] ** 2)
    assert (h[1::3] < 1e-3).any() and (h[1::3] >= 1e-3).any()          # both sides of the threshold occur
    got, want = _yaw_via_sampler(ctx, vel), yaws_from_velocity(vel)
    assert np.allclose(got, want, rtol=0, atol=1e-12), np.flatnonzero(np.abs(got - want) > 1e-12)
