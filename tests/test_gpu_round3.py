"""GPU tests of round 3: the gather of the PLAN (root re-samples the peers' rows, bit-identical to the row gather), the
atomic refusal of a re-plan that does not fit, the yaw scan of a fleet whose plan is re-solved in flight, row offsets from
row counts, and the RCCL version guard."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


def test_row_offsets_from_row_counts(eng, nat):
    import torch
    plan = eng.plan(_missions(1000, 5), 3.0, 0.01)
    ro = torch.full_like(plan.row_offsets, -1)
    eng._bind_stream()
    eng.ctx.call("uavac_minsnap_row_offsets_dev", C.c_void_p(plan.seg_rows.data_ptr()), plan.B, plan.m, C.c_void_p(ro.data_ptr()))
    assert torch.equal(ro, plan.row_offsets)
    with pytest.raises(nat.UavacError):
        eng.ctx.call("uavac_minsnap_row_offsets_dev", C.c_void_p(plan.seg_rows.data_ptr()), plan.B, plan.m, None)


def test_plan_from_parts_resamples_bit_identical_rows_at_any_alignment(eng):
    """What the root of the plan gather does: rows from (coefficients, rows per spline, dt) alone.  The row buffer of the
    re-sampled plan starts elsewhere (other 128-byte phase of every mission): the sampler's store pattern follows the
    address, the values do not."""
    import torch
    plan = eng.plan(_missions(777, 8), 3.0, 0.01)
    again = eng.plan_from_parts(plan.coeffs.clone(), plan.times.clone(), plan.seg_rows.clone(), 8, 3.0, 0.01)
    assert again.total_rows == plan.total_rows and torch.equal(again.row_offsets, plan.row_offsets)
    assert torch.equal(again.traj, plan.traj) and torch.equal(again.first_yaw, plan.first_yaw)
    # a tail of the batch, into a preallocated buffer, rows landing at another line phase
    k = 5
    big = torch.full((plan.total_rows + 3, 11), -1.0, dtype=torch.float64, device=eng.device)
    tail = eng.plan_from_parts(plan.coeffs[k:], None, plan.seg_rows[k:], 8, 3.0, 0.01, traj=big[3:])
    r0 = int(plan.row_offsets[k])
    assert tail.total_rows == plan.total_rows - r0
    assert torch.equal(tail.traj, plan.traj[r0:]) and bool((big[:3] == -1.0).all())
    # a fleet flies the re-assembled plan like the original (start positions come from the coefficients)
    f0, f1 = eng.fleet(plan, from_plan=True), eng.fleet(again, from_plan=True)
    f0.rollout(700)
    f1.rollout(700)
    assert torch.equal(f0.state, f1.state) and torch.equal(f0.istate, f1.istate)


def test_plan_gather_world1_equals_row_gather(eng, nat):
    """The RCCL calls of the plan gather at world 1 (the root's own block) + the root's re-sampling: the rows equal what the
    row gather delivers, bit for bit."""
    import torch
    from uav_ac.fleet import RcclComm
    buf = C.create_string_buffer(nat.COMM_ID_BYTES)
    eng.ctx.call("uavac_comm_unique_id", buf)
    comm = RcclComm(eng, unique_id=bytes(buf.raw), world=1, rank=0)
    try:
        plan = eng.plan(_missions(3000, 8), 3.0, 0.01)
        rows, counts = comm.gather_rows(plan.traj, dst=0)
        gp, pcounts = comm.gather_plan(plan, dst=0)
        assert pcounts == counts == [plan.total_rows]
        assert gp.coeffs.data_ptr() != plan.coeffs.data_ptr()
        assert torch.equal(gp.traj, rows) and torch.equal(gp.coeffs, plan.coeffs) and torch.equal(gp.times, plan.times)
        assert torch.equal(gp.seg_rows, plan.seg_rows) and torch.equal(gp.row_offsets, plan.row_offsets)
        assert torch.equal(gp.first_yaw, plan.first_yaw)
        # beside other work on a second stream
        side = torch.cuda.Stream(device=eng.device)
        ticket = comm.gather_plan_begin(plan, dst=0, stream=side)
        fleet = eng.fleet(plan)
        fleet.rollout(200)
        gp2, _ = comm.gather_finish(ticket)
        torch.cuda.synchronize()
        assert torch.equal(gp2.traj, plan.traj)
        # argument checking happens before anything is enqueued
        cnt = (C.c_int64 * 1)(plan.B * plan.m)
        p = lambda t: C.c_void_p(t.data_ptr())     # noqa: E731
        with pytest.raises(nat.UavacError) as e:   # counts[rank] must be this rank's segment count
            eng.ctx.call("uavac_gather_plan_dev", comm._h, p(plan.coeffs), p(plan.times), p(plan.seg_rows), 5, cnt, 0,
                         p(gp.coeffs), p(gp.times), p(gp.seg_rows))
        assert e.value.code == nat.EINVAL
        with pytest.raises(nat.UavacError) as e:   # times on one side only
            eng.ctx.call("uavac_gather_plan_dev", comm._h, p(plan.coeffs), None, p(plan.seg_rows), plan.B * plan.m, cnt, 0,
                         p(gp.coeffs), p(gp.times), p(gp.seg_rows))
        assert e.value.code == nat.EINVAL
        with pytest.raises(nat.UavacError):
            eng.ctx.call("uavac_gather_plan_dev", comm._h, p(plan.coeffs), p(plan.times), p(plan.seg_rows), plan.B * plan.m, cnt, 2,
                         p(gp.coeffs), p(gp.times), p(gp.seg_rows))
    finally:
        comm.close()


def test_rccl_versions_are_reported_and_compatible(nat):
    built, rt = C.c_int(0), C.c_int(0)
    assert nat.lib().uavac_comm_versions(C.byref(built), C.byref(rt)) == 0
    assert built.value >= 21000 and rt.value >= 21000          # 2.10+
    assert built.value // 10000 == rt.value // 10000            # same major: what uavac_comm_init_rank insists on


def test_refused_replan_leaves_the_whole_plan_untouched_and_flyable(eng):
    """round-2 ADVICE (medium): a re-plan that needs more rows than plan.traj holds used to overwrite times, row counts,
    offsets and coefficients before the sampler refused -- a row-fed rollout then indexed past the row buffer."""
    import torch
    wps = _missions(500, 8)
    plan = eng.plan(wps, 3.0, 0.01)
    fleet = eng.fleet(plan, from_plan=False)
    ref_fleet = eng.fleet(plan, from_plan=False)
    keep = {k: getattr(plan, k).clone() for k in ("times", "seg_rows", "row_offsets", "coeffs", "traj", "first_yaw", "status")}
    # a slower cruise needs more rows than the buffer holds
    old_v, plan.velocity = plan.velocity, 1.5
    eng.replan(plan)
    assert eng.take_flags() == [0, 0, 1, 0]
    for k, v in keep.items():
        assert torch.equal(getattr(plan, k), v), k
    plan.velocity = old_v
    fleet.rollout(1500)
    ref_fleet.rollout(1500)
    assert torch.equal(fleet.state, ref_fleet.state)
    # and a plan that fits again goes through (same waypoints, faster cruise: fewer rows)
    plan.velocity = 3.5
    eng.replan(plan)
    assert eng.take_flags() == [0, 0, 0, 0]
    fresh = eng.plan(wps, 3.5, 0.01)
    n = fresh.total_rows
    assert torch.equal(plan.row_offsets, fresh.row_offsets) and torch.equal(plan.traj[:n], fresh.traj)
    assert torch.equal(plan.coeffs, fresh.coeffs) and torch.equal(plan.times, fresh.times)


def test_replan_under_a_flying_fleet_rebuilds_the_carried_yaw_scan(eng):
    """round-2 ADVICE (low): the yaw scan a plan-fed vehicle carries (state rows 26-29) belongs to the coefficients it was
    built from.  After Engine.replan without Fleet.reset() the plan-fed fleet must see the NEW plan's yaw, like the row-fed
    fleet that reads the new rows."""
    import torch
    wps = _missions(256, 6)
    plan = eng.plan(wps, 3.0, 0.01)
    fed, rows = eng.fleet(plan, from_plan=True), eng.fleet(plan, from_plan=False)
    fed.rollout(2000)
    rows.rollout(2000)
    assert torch.equal(fed.state[:26], rows.state[:26])
    # same row counts (same leg lengths), other headings: mirror the missions in y about their start
    w2 = torch.as_tensor(wps, device=eng.device).clone()
    w2[:, :, 1] = 2 * w2[:, :1, 1] - w2[:, :, 1]
    plan.waypoints.copy_(w2)
    eng.replan(plan)
    assert eng.take_flags() == [0, 0, 0, 0]
    fed.rollout(1500)
    rows.rollout(1500)
    assert torch.equal(fed.state[:26], rows.state[:26]) and torch.equal(fed.istate, rows.istate)


@pytest.mark.parametrize("B", [1000, 4097, 70001])
def test_pitched_logs_equal_dense_logs_and_leave_the_padding_alone(eng, B):
    """round-2 VERDICT 4a: log rows `pitch` doubles apart (uavac_set_option "log_pitch"; Fleet picks a multiple of 16).
    Same values as the dense [K][13][B] layout, padding columns never written.  B = 70 001 also walks the persistent-tile
    loop (more 64-UAV tiles than SIMDs) with a ragged last tile."""
    import torch
    plan = eng.plan(_missions(B, 3), 3.0, 0.01)
    K = 130
    f0, f1 = eng.fleet(plan, from_plan=False), eng.fleet(plan, from_plan=False)
    dense_s = torch.empty((K, 13, B), dtype=torch.float64, device=eng.device)
    dense_c = torch.empty((K, 12, B), dtype=torch.float64, device=eng.device)
    f0.rollout(K, state_log=dense_s, cmd_log=dense_c)
    P = -(-B // 16) * 16 + 48
    ps = torch.full((K, 13, P), -5.0, dtype=torch.float64, device=eng.device)
    pc = torch.full((K, 12, P), -5.0, dtype=torch.float64, device=eng.device)
    s_view, c_view = f1.rollout(K, state_log=ps, cmd_log=pc, log_pitch=P)      # pitched caller buffers are opt-in
    assert s_view.data_ptr() == ps.data_ptr() and s_view.shape == (K, 13, B) and c_view.shape == (K, 12, B)
    assert torch.equal(ps[:, :, :B], dense_s) and torch.equal(pc[:, :, :B], dense_c)
    assert bool((ps[:, :, B:] == -5.0).all()) and bool((pc[:, :, B:] == -5.0).all())
    assert torch.equal(f0.state, f1.state) and torch.equal(f0.istate, f1.istate)
    # logs allocated by the fleet come back as (K, rows, B) views of pitched buffers
    f2 = eng.fleet(plan, from_plan=False)
    s2, c2 = f2.rollout(K, state_log=True, cmd_log=True)
    assert s2.shape == (K, 13, B) and s2.stride(1) % 16 == 0
    assert torch.equal(s2, dense_s) and torch.equal(c2, dense_c)
    # round-3 ADVICE: the pitch is never inferred from a caller's tensor -- a flat buffer is written densely; round-4 ADVICE: a 3-D
    # tensor whose rows are NOT the pitch that will be written is refused (indexing it as (K, 13, P) afterwards would read
    # scrambled data without any error), ...
    f3 = eng.fleet(plan, from_plan=False)
    wide = torch.full((K, 13, P), -5.0, dtype=torch.float64, device=eng.device)
    with pytest.raises(ValueError, match="log_pitch"):
        f3.rollout(K, state_log=wide)
    assert bool((wide == -5.0).all())                                  # refused before anything was launched
    flat = wide.reshape(-1)
    got, _ = f3.rollout(K, state_log=flat)
    assert got is flat and torch.equal(flat[:K * 13 * B].view(K, 13, B), dense_s) and bool((flat[K * 13 * B:] == -5.0).all())
    # ... a log allocated beside a caller's dense one takes the same pitch instead of being refused (B % 16 != 0 too), ...
    f4 = eng.fleet(plan, from_plan=False)
    mine = torch.empty((K, 12, B), dtype=torch.float64, device=eng.device)
    s4, c4 = f4.rollout(K, state_log=True, cmd_log=mine)
    assert s4.shape == (K, 13, B) and s4.is_contiguous() and c4 is mine
    assert torch.equal(s4, dense_s) and torch.equal(mine, dense_c)
    # ... and beside a pitched one as well
    f5 = eng.fleet(plan, from_plan=False)
    pc2 = torch.full((K, 12, P), -5.0, dtype=torch.float64, device=eng.device)
    s5, c5 = f5.rollout(K, state_log=True, cmd_log=pc2, log_pitch=P)
    assert s5.shape == (K, 13, B) and s5.stride(1) == P and torch.equal(s5, dense_s) and torch.equal(c5, dense_c)
    with pytest.raises(ValueError):
        f5.rollout(K, state_log=dense_s, log_pitch=B + 16)          # too small for that pitch
    with pytest.raises(ValueError):
        f5.rollout(K, state_log=True, log_pitch=B - 1)
    # a pitch below B is refused
    eng.ctx.set_option("log_pitch", B - 1)
    try:
        with pytest.raises(Exception):
            f2._launch_rollout(1, dense_s, None, None)
    finally:
        eng.ctx.set_option("log_pitch", 0)


def test_placeholder_wave_between_compute_and_store_wave_changes_no_bit(eng):
    """Option "idle_waves": a third wave between the compute and the store wave of every workgroup that ends at once, so that
    two workgroups on a CU get a SIMD each for all four of their working waves (the launcher's choice for 16 385 .. 32 768
    UAVs).  The workgroup's barriers go on without the wave that has ended; state log, command log, collision flags and
    final state equal those of the two-wave launch, for both feeds."""
    import torch
    B, K = 3000, 400
    plan = eng.plan(_missions(B, 8), 3.0, 0.01)
    boxes = np.array([[3.7, 4.3, 4, 10, -3.4, -2.8], [10.7, 11.3, 4, 10, -2.2, 0], [13.3, 14.7, 6.3, 7.7, -6, 0],
                      [20.2, 20.8, 4, 10, -3.3, -2.7], [0, 30, 0, 20, -3.2, -3.1], [5, 6, 5, 6, -4, -2], [7, 9, 1, 3, -5, -1],
                      [1, 2, 1, 2, -4, -2], [11, 12, 3, 9, -4, -2], [15, 16, 2, 4, -4, -2]])
    for feed in (False, True):
        for n_obs, logs in ((0, (True, False)), (10, (True, True)), (4, (False, True)), (8, (True, False)), (3, (False, False))):
            out = []
            for idle in (0, 1):
                eng.ctx.set_option("idle_waves", idle)
                try:
                    f = eng.fleet(plan, from_plan=feed)
                    s, c = f.rollout(K, state_log=logs[0] or None, cmd_log=logs[1] or None,
                                     aabbs=boxes[:n_obs] if n_obs else None)
                    torch.cuda.synchronize()
                finally:
                    eng.ctx.set_option("idle_waves", -1)
                out.append((s, c, f.state.clone(), f.istate.clone()))
            for a, b in zip(*out):
                assert (a is None and b is None) or torch.equal(a, b), (feed, n_obs, logs)
            if n_obs:
                assert int(out[0][3][2].sum()) > 0             # somebody did fly into a cuboid
    # the launcher's own choice at a half-full chip equals the explicit two-wave launch too
    plan = eng.plan(_missions(20000, 2), 3.0, 0.01)
    logs = []
    for idle in (0, -1):
        eng.ctx.set_option("idle_waves", idle)
        f = eng.fleet(plan, from_plan=False)
        logs.append(f.rollout(60, state_log=True)[0])
    eng.ctx.set_option("idle_waves", -1)
    assert torch.equal(logs[0], logs[1])


def _sampler_cases(seed):
    """Missions that exercise every path of the sampler's yaw scan and row layout: the 8(d) missions, one-spline missions,
    a fine time step, long vertical climbs (no heading for 1 800-2 700 rows: placeholders patched later), circles (headings
    wrap in many chunks: the ordered path of the scan) and legs of 1-3 rows."""
    rng = np.random.default_rng(seed)
    cases = [(_missions(300, 12), 3.0, 0.01), (_missions(64, 1), 3.0, 0.01), (_missions(50, 20), 0.9, 0.004)]
    climb = []
    for i in range(40):
        h = rng.uniform(8.0, 12.0)
        p0 = np.array([rng.uniform(0, 5), rng.uniform(0, 5), -1.0])
        legs = np.cumsum(rng.uniform(-3, 3, (5, 3)) * np.array([1, 1, 0.1]), axis=0)
        climb.append(np.vstack([p0, p0 + [0, 0, -h], p0 + [0, 0, -h] + legs]))
    cases.append((np.stack(climb), 1.0, 0.01))
    circ = []
    for i in range(40):
        n, sense = 29, (1 if i % 2 else -1)
        th = sense * np.linspace(0, 7 * np.pi, n) + rng.uniform(0, 2 * np.pi)
        rad = rng.uniform(1.5, 4.0)
        circ.append(np.stack([10 + rad * np.cos(th), 10 + rad * np.sin(th), -3 + 0.1 * np.sin(3 * th)], axis=1))
    cases.append((np.stack(circ), 2.0, 0.01))
    cases.append((_missions(100, 6) * 0.02, 3.0, 0.01))
    return cases


@pytest.mark.parametrize("waves,group", [(4, 1), (2, 1), (8, 1), (16, 1), (8, 2), (8, 5), (4, 3), (2, 6), (16, 7)])
def test_streaming_sampler_equals_the_one_wave_sampler_bit_for_bit(eng, waves, group):
    """minsnap_sample_stream.hip (workgroups of W wavefronts stream the 64-row chunks of G consecutive missions in address
    order, the yaw scan's carry handed from wave to wave through LDS) against minsnap_sample.hip (one wavefront walks a
    mission): rows and first headings bit for bit, for every W and G, and whatever the row buffer's alignment (the chunk
    grid follows the buffer's address: three offsets of the same rows)."""
    import torch
    for wps, v, dt in _sampler_cases(77 + waves):
        got = {}
        for sw in (1, waves):
            eng.ctx.set_option("sampler_waves", sw)
            eng.ctx.set_option("sampler_group", group)
            try:
                plan = eng.plan(wps, v, dt, placement_trials=1)
                plan.traj.fill_(float("nan"))
                plan.first_yaw.fill_(float("nan"))
                eng.replan(plan)                               # the one-call chain takes the same sampler
                assert eng.take_flags() == [0, 0, 0, 0]
                got[sw] = (plan.traj.clone(), plan.first_yaw.clone(), plan.row_offsets.clone())
                if sw != 1:
                    for off in (1, 6, 11):                     # the same rows into a buffer that starts `off` doubles later
                        big = torch.full((plan.total_rows * 11 + 32,), float("nan"), dtype=torch.float64, device=eng.device)
                        keep = plan.traj
                        plan.traj = big[off:off + plan.total_rows * 11].view(plan.total_rows, 11)
                        eng.sample(plan)
                        assert torch.equal(plan.traj, got[sw][0]), off
                        assert bool(torch.isnan(big[:off]).all()) and bool(torch.isnan(big[off + plan.total_rows * 11:]).all())
                        plan.traj = keep
            finally:
                eng.ctx.set_option("sampler_waves", 4)
                eng.ctx.set_option("sampler_group", 1)
        assert torch.equal(got[1][2], got[waves][2])
        assert not bool(torch.isnan(got[waves][0]).any())
        assert torch.equal(got[1][0], got[waves][0]) and torch.equal(got[1][1], got[waves][1])


def test_streaming_sampler_edge_shapes(eng):
    """The streaming sampler where its item arithmetic is thin: ONE mission of tens of thousands of rows (hundreds of items per
    wave, the mailbox sequence numbers run long), a batch of one-row and two-row missions (several missions per 64-row chunk:
    every item partial), more missions per workgroup than the batch has, and the oracle as referee for one of each."""
    import torch
    from oracle import minsnap_oracle as mo
    long_one = np.array([[[0.0, 0.0, -1.0], [120.0, 40.0, -3.0], [60.0, 160.0, -2.0], [-50.0, 90.0, -4.0]]])    # ~45 000 rows at v = 1, dt = 0.01
    tiny = mo.synthetic_missions(333, 2) * 0.004                                  # legs of ~1 cm: one or two rows per spline
    for wps, v, dt in ((long_one, 1.0, 0.01), (tiny, 3.0, 0.01), (mo.synthetic_missions(3, 5), 3.0, 0.01)):
        got = {}
        for sw, g in ((1, 1), (4, 1), (4, 7), (2, 64), (16, 3)):
            eng.ctx.set_option("sampler_waves", sw)
            eng.ctx.set_option("sampler_group", g)
            try:
                plan = eng.plan(wps, v, dt)
                assert eng.take_flags() == [0, 0, 0, 0]
                got[(sw, g)] = (plan.traj.clone(), plan.first_yaw.clone())
            finally:
                eng.ctx.set_option("sampler_waves", 4)
                eng.ctx.set_option("sampler_group", 1)
        for key, val in got.items():
            assert torch.equal(val[0], got[(1, 1)][0]) and torch.equal(val[1], got[(1, 1)][1]), key
        ref = mo.plan(wps[0], v, dt, method="solve")
        mine = plan.mission(0)
        assert mine.shape == ref.shape and np.array_equal(mine[:, 10], ref[:, 10])
        err = np.max(np.abs(mine - ref), axis=0) / np.maximum(1.0, np.max(np.abs(ref), axis=0))
        assert err.max() < 1e-5, err
    assert long_one.shape[0] == 1 and plan is not None


def test_streaming_sampler_variants_equal_the_one_wave_sampler(eng):
    """The streaming sampler's other outputs -- dense yaw column, jerk / snap, hit flags of a cuboid, ragged batches --
    against the one-wave-per-mission kernel, bit for bit."""
    import torch
    lab = np.array([[3.7, 4.3, 4.0, 10.0, -3.4, -2.8], [10.7, 11.3, 4.0, 10.0, -2.2, 0.0]])
    for wps, v, dt in _sampler_cases(5)[:5]:
        out = {}
        for sw, g in ((1, 1), (4, 1), (8, 3)):
            eng.ctx.set_option("sampler_waves", sw)
            eng.ctx.set_option("sampler_group", g)
            try:
                plan = eng.plan(wps, v, dt, dense_yaw=True, placement_trials=1)
                jerk, snap = eng.sample_derivatives(plan)
                rag = eng.plan_ragged([w[: 2 + (i % (len(w) - 1))] for i, w in enumerate(wps)], v, dt, cuboid=lab[0])
                out[(sw, g)] = (plan.traj.clone(), plan.yaw.clone(), jerk.clone(), snap.clone(), rag.traj.clone(), rag.hit.clone(),
                                rag.first_yaw.clone())
            finally:
                eng.ctx.set_option("sampler_waves", 4)
                eng.ctx.set_option("sampler_group", 1)
        ref = out[(1, 1)]
        assert torch.equal(ref[1], ref[0][:, 9])
        for key, val in out.items():
            for x, y in zip(ref, val):
                assert torch.equal(x, y), key


def test_obstacle_loop_on_the_device_equals_the_host_side_loop(eng):
    """round-2 VERDICT 7: the midpoint insertion of the obstacle loop (minimum_snap.py:359-391 inside :63-95) as a kernel,
    a round = one C call.  Same final waypoints, same rows and same give-up flags as round 2's loop that inserted the
    midpoints with NumPy -- on 8(d) missions through the lab cuboids (a few of which run the bounded loop to its end), with
    and without the extra re-check passes."""
    import torch
    lab = np.array([[3.7, 4.3, 4.0, 10.0, -3.4, -2.8], [10.7, 11.3, 4.0, 10.0, -2.2, 0.0],
                    [13.3, 14.7, 6.3, 7.7, -6.0, 0.0], [20.2, 20.8, 4.0, 10.0, -3.3, -2.7]])
    wps = _missions(700, 8)
    ragged = [w[:rng_n] for w, rng_n in zip(wps, np.random.default_rng(5).integers(3, 10, len(wps)))]
    for passes in (0, 2):
        host = eng.plan_collision_free(ragged, lab, 3.0, 0.01, strict=False, recheck_passes=passes, max_iterations=12,
                                       device_loop=False)
        dev = eng.plan_collision_free(ragged, lab, 3.0, 0.01, strict=False, recheck_passes=passes, max_iterations=12)
        assert np.array_equal(host.converged, dev.converged)
        assert (~dev.converged).sum() >= 1 and sum(len(w) > len(r) for w, r in zip(dev.final_waypoints, ragged)) >= 20
        for b in np.flatnonzero(dev.converged):
            assert np.array_equal(host.final_waypoints[b], dev.final_waypoints[b]), b
        ok = torch.as_tensor(np.flatnonzero(dev.converged), device=eng.device)
        n_h = (host.row_offsets[1:] - host.row_offsets[:-1])[ok]
        n_d = (dev.row_offsets[1:] - dev.row_offsets[:-1])[ok]
        assert torch.equal(n_h, n_d)
        for b in np.flatnonzero(dev.converged)[:200]:
            assert np.array_equal(host.mission(b), dev.mission(b)), b
    # strict mode raises like the host-side loop
    bar = np.array([[4.0, 5.0, -10.0, 10.0, -10.0, 10.0]])
    through = np.array([[0.0, 0.0, -2.0], [9.0, 0.0, -2.0], [9.0, 5.0, -2.0]])
    with pytest.raises(RuntimeError):
        eng.plan_collision_free([through], bar, 2.0, 0.01)
    # no obstacles at all: plain ragged planning
    none = eng.plan_collision_free(ragged[:50], None, 3.0, 0.01)
    rb = eng.plan_ragged(ragged[:50], 3.0, 0.01)
    assert torch.equal(none.traj, rb.traj) and none.converged.all()


def test_yaw_column_never_forks_from_numpy_on_the_baseline_distribution(eng):
    """round-2 VERDICT 9: the fuzz test compares the yaw modulo 2 pi because np.unwrap can fork by 2 pi where a heading
    reverses through zero (a vertical leg followed by a horizontal one: |delta| = pi to the last bit, decided by the rounding
    of two atan2 results).  On the 8(d) distribution -- and on U(1, 6) m -- that does not occur: NumPy's own yaw scan on the
    GPU rows' velocities gives the GPU's yaw column, not merely modulo 2 pi."""
    from oracle import minsnap_oracle as mo
    for lo, hi in ((2.5, 3.5), (1.0, 6.0)):
        wps = mo.synthetic_missions(400, 12, lo, hi)
        plan = eng.plan(wps, 3.0, 0.01)
        rows, ro = plan.traj.cpu().numpy(), plan.row_offsets.cpu().numpy()
        worst = 0.0
        for b in range(plan.B):
            r = rows[ro[b]:ro[b + 1]]
            worst = max(worst, float(np.abs(r[:, 9] - mo.yaws_from_velocity(r[:, 3:6])).max()))
        assert worst < 1e-9, (lo, hi, worst)


def test_ragged_plan_gather_world1_and_plan_fed_flight_of_an_obstacle_corrected_plan(eng, nat):
    """The plan gather for ragged batches (what the obstacle loop produces): splines per mission travel as one more column,
    the root re-samples with the ragged sampler -- rows bit-identical to the row gather.  And the RaggedPlan of
    plan_collision_free now carries its coefficients: a Fleet flies it from them exactly as from its rows."""
    import torch
    from uav_ac.fleet import RcclComm
    lab = np.array([[3.7, 4.3, 4.0, 10.0, -3.4, -2.8], [10.7, 11.3, 4.0, 10.0, -2.2, 0.0],
                    [13.3, 14.7, 6.3, 7.7, -6.0, 0.0], [20.2, 20.8, 4.0, 10.0, -3.3, -2.7]])
    wps = _missions(900, 8)
    ragged = [w[:n] for w, n in zip(wps, np.random.default_rng(11).integers(2, 10, len(wps)))]
    rp = eng.plan_collision_free(ragged, lab, 3.0, 0.01, strict=False, max_iterations=10)
    assert rp.batch is not None and rp.coeffs.shape[0] == int(rp.seg_offsets_host[-1])
    buf = C.create_string_buffer(nat.COMM_ID_BYTES)
    eng.ctx.call("uavac_comm_unique_id", buf)
    comm = RcclComm(eng, unique_id=bytes(buf.raw), world=1, rank=0)
    try:
        rows, counts = comm.gather_rows(rp.traj, dst=0)
        for plan in (rp, rp.batch):
            gp, pcounts = comm.gather_plan(plan, dst=0)
            assert pcounts == counts and torch.equal(gp.traj, rows)
            assert torch.equal(gp.seg_offsets, rp.seg_offsets) and torch.equal(gp.coeffs, rp.coeffs)
            assert torch.equal(gp.row_offsets, rp.row_offsets) and torch.equal(gp.first_yaw, rp.first_yaw)
            assert torch.equal(gp.start_positions, rp.start_positions)
    finally:
        comm.close()
    flights = []
    for feed in (False, True):
        f = eng.fleet(rp, from_plan=feed)
        log, _ = f.rollout(900, state_log=True)
        flights.append((log, f.state[:26].clone(), f.istate.clone()))
    for a, b in zip(*flights):
        assert torch.equal(a, b)
    # the gathered batch flies too (its start positions come from the coefficients)
    f = eng.fleet(gp, from_plan=True)
    log, _ = f.rollout(900, state_log=True)
    assert torch.equal(log, flights[0][0])


def test_host_pointer_obstacle_loop_equals_the_engine(eng, nat):
    """uavac_minsnap_obstacle_waypoints (plain host buffers, the whole obstacle loop inside one C call) against
    Engine.plan_collision_free: same final waypoint lists and give-up flags, with and without re-check sweeps; then the rows
    through uavac_minsnap_plan_ragged equal the engine's."""
    lab = np.array([[3.7, 4.3, 4.0, 10.0, -3.4, -2.8], [10.7, 11.3, 4.0, 10.0, -2.2, 0.0],
                    [13.3, 14.7, 6.3, 7.7, -6.0, 0.0], [20.2, 20.8, 4.0, 10.0, -3.3, -2.7]])
    wps = _missions(400, 8)
    ragged = [np.ascontiguousarray(w[:n]) for w, n in zip(wps, np.random.default_rng(3).integers(2, 10, len(wps)))]
    B = len(ragged)
    so = np.zeros(B + 1, dtype=np.int64)
    np.cumsum([len(w) - 1 for w in ragged], out=so[1:])
    flat = np.ascontiguousarray(np.concatenate(ragged))
    for passes in (0, 2):
        ref = eng.plan_collision_free(ragged, lab, 3.0, 0.01, strict=False, max_iterations=10, recheck_passes=passes)
        cap = B * (nat.MAX_SEGMENTS + 1)
        wp_out, so_out, ok = np.empty((cap, 3)), np.empty(B + 1, dtype=np.int64), np.zeros(B, dtype=np.int32)
        eng.ctx.call("uavac_minsnap_obstacle_waypoints", nat.np_ptr(flat), nat.np_ptr(so), B, 3.0, 0.01, nat.np_ptr(lab), len(lab), 10,
                     passes, nat.np_ptr(wp_out), cap, nat.np_ptr(so_out), nat.np_ptr(ok))
        assert np.array_equal(ok.astype(bool), ref.converged) and (~ref.converged).sum() >= 1
        for b in range(B):
            got = wp_out[so_out[b] + b:so_out[b + 1] + b + 1]
            assert np.array_equal(got, ref.final_waypoints[b]), b
        # rows of the final waypoints through the host-pointer ragged planner
        S = int(so_out[-1])
        ro = np.empty(B + 1, dtype=np.int64)
        final = np.ascontiguousarray(wp_out[:S + B])
        eng.ctx.call("uavac_minsnap_plan_ragged", nat.np_ptr(final), nat.np_ptr(so_out), B, 3.0, 0.01, None, nat.np_ptr(ro), None, None, 0)
        rows = np.empty((int(ro[-1]), 11))
        eng.ctx.call("uavac_minsnap_plan_ragged", nat.np_ptr(final), nat.np_ptr(so_out), B, 3.0, 0.01, None, nat.np_ptr(ro), None,
                     nat.np_ptr(rows), len(rows))
        assert np.array_equal(ro, ref.row_offsets.cpu().numpy()) and np.array_equal(rows, ref.traj.cpu().numpy())
    # capacity too small: refused with the needed size reported
    with pytest.raises(nat.UavacError):
        eng.ctx.call("uavac_minsnap_obstacle_waypoints", nat.np_ptr(flat), nat.np_ptr(so), B, 3.0, 0.01, nat.np_ptr(lab), len(lab), 10, 0,
                     nat.np_ptr(wp_out), 5, nat.np_ptr(so_out), nat.np_ptr(ok))
    assert so_out[-1] >= so[-1]


def test_bench_line_schema_with_extras():
    """`python bench.py` as the driver runs it (fewer steps): ONE JSON line with the contract's keys, `roofline` and `minsnap` priced
    both by algorithmic and by counter bytes when profiles/hbm_traffic.json matches the sources, the planning chain as median and
    spread over the timed steps with the shader clock of a second, probed leg beside it (round-4 VERDICT 4: no placement search in
    the driver's bench any more), per-launch times, exit code 0."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-config4"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "minsnap", "build"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["value"] > 1e10
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and 0.5 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert len(rf["per_launch_ms_one_step"]) == 10 and rf["kernel_vgprs"] <= 256
    m = d["minsnap"]
    # round 6 (round-5 advice): the old keys are on the MEAN of the timed steps again, the median has its own names
    assert 0.45 < m["roofline"]["frac"] < 0.85 and abs(m["roofline"]["frac"] * m["ms_mean"] - m["roofline"]["frac_median"] * m["ms_median"]) < 1e-9
    assert m["ms_solve_plus_sample"] == m["ms_mean"] and abs(m["value"] * m["ms_mean"] - m["value_median"] * m["ms_median"]) < 1e-3 * m["value"]
    assert "row_buffer_searched_once" not in m and "MEAN" in m["statistic"]
    assert m["ms_min"] <= m["ms_median"] <= m["ms_max"] and len(m["ms_per_timed_step"]) == 3 and 0.45 < m["frac_median"] < 0.85
    assert '"early": true' in r.stderr                                  # the headline went to stderr as soon as it existed
    ck = m["clocked_steps"]
    assert "error" not in ck and len(ck["planning_ms"]) == 8 and all(1.0 < g < 2.6 for g in ck["planning_clock_ghz"])
    assert m["clock_ghz"] == ck["clock_ghz"] and 0.9 < m["leg_to_leg"] < 1.1
    if rf["traffic"] is not None:                                        # profiles/hbm_traffic.json was measured on these sources
        assert 0.85 < rf["traffic"] / rf["algorithmic_bytes_per_launch"] < 1.05 and rf["frac_counter_bytes"] < rf["frac"]


def test_full_size_planning_chain_is_the_same_under_every_launch_shape(eng):
    """BASELINE configs[2] at its full size (65 536 missions of 12 segments, 85.6 M rows): the coefficients of the solve as the
    launcher shapes it there (five knots in registers, one in LDS, the last never parked) equal those of the plain 64-lane
    workspace form bit for bit, and the rows of the chunk-streaming sampler equal the one-wave sampler's -- checked through a
    64-bit checksum per mission of both (size-independent: no second 7.5 GB buffer on the host)."""
    import torch
    from bench import missions
    wps = missions(65536, 12, 0, 65536)
    plan = eng.plan(wps, 3.0, 0.01)

    def digest():
        torch.cuda.synchronize()
        c = plan.coeffs.view(torch.int64).reshape(plan.B, -1).sum(dim=1)
        rows = plan.traj.view(torch.int64).sum(dim=1)                           # wrap-around int64 sums: a checksum, order-free per row
        per_mission = torch.zeros(plan.B, dtype=torch.int64, device=rows.device)
        ids = torch.repeat_interleave(torch.arange(plan.B, device=rows.device), (plan.row_offsets[1:] - plan.row_offsets[:-1]))
        per_mission.index_add_(0, ids, rows * (torch.arange(rows.numel(), device=rows.device) % 1000003 + 1))
        return c.clone(), per_mission

    ref_c, ref_r = digest()
    try:
        for opts in ({"solve_keep": 0, "solve_lanes": 64, "solve_park": 0, "sampler_waves": 1}, {"solve_keep": 1, "sampler_waves": 2},
                     {"solve_lanes": 32, "solve_keep": 0, "sampler_waves": 8, "sampler_group": 2}):
            for k, v in opts.items():
                eng.ctx.set_option(k, v)
            plan.coeffs.fill_(float("nan")); plan.traj.fill_(float("nan"))
            eng.replan(plan)
            c, r = digest()
            assert torch.equal(c, ref_c) and torch.equal(r, ref_r), opts
            for k in opts:
                eng.ctx.set_option(k, 4 if k == "sampler_waves" else 1 if k == "sampler_group" else -1)
    finally:
        for k, v in (("solve_keep", -1), ("solve_lanes", -1), ("solve_park", -1), ("sampler_waves", 4), ("sampler_group", 1)):
            eng.ctx.set_option(k, v)
