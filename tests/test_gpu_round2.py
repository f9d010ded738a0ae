"""GPU tests of what round 2 added behind the C ABI: the RCCL gather (world-1 self-test of the very calls the 8-GPU
gather makes), the one-call planning chain, sticky-flag handling, the stand-alone yaw scan, jerk / snap outputs, the
device guard, the rollout's launch shapes, and the per-rank shard of BASELINE config 4 at full size."""
import ctypes as C

import numpy as np
import pytest

from conftest import col_err, load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def nat():
    from uav_ac import _native
    return _native


@pytest.fixture(scope="module")
def eng():
    from uav_ac.fleet import Engine
    return Engine("cuda:0")


# ------------------------------------------------------------------------------------------- RCCL behind the C ABI
def test_bench_multi_gpu_path_on_real_rccl_at_world_1():
    """bench.py's N > 1 code with WORLD_SIZE = 1 (UAVAC_BENCH_FORCE_DIST=1): torch's RCCL process group with a device id,
    barrier and max-reduction on GPU tensors, the unique id broadcast over that group, ncclCommInitRank behind the C ABI,
    the config-4 leg at its full 262 144 UAVs with the gather timed and verified, orderly shut-down, exit code 0."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UAVAC_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "gather_error" not in line and "FORCED_DIST" in line and line["n_gpus"] == 1
    c4 = line["config4"]
    assert c4["batch_total"] == 262144 and c4["batch_per_gpu"] == 262144 and c4["gather_verified"] is True
    assert c4["gather_rows_total"] == c4["rows_rank0"] and c4["gather_ms"] > 0
    # round 3: the gather of the plan + re-sampling on the root, verified bit for bit against the gathered rows
    assert c4["plan_gather_verified"] is True and c4["plan_gather_ms"] > 0 and c4["steps_per_s_with_plan_gather"] > 0
    assert c4["plan_overlapped_verified"] is True and "overlap_error" not in c4
    # round 6: rows-free plans, the pipelined gather (parts on a transfer stream), the rows sampled once -- verified against the
    # literal row gather, and the job's end-to-end form
    assert c4["pipelined_plan_gather_verified"] is True and c4["rows_free_overlapped_verified"] is True
    assert c4["end_to_end"]["form"].startswith("every rank plans ROWS-FREE") and c4["compute_rows_free_ms"] < c4["compute_ms"]
    assert c4["rccl_versions"]["built_with"] > 0 and c4["rccl_versions"]["runtime"] > 0
    assert "early" in r.stderr and '"value"' in r.stderr          # the headline on stderr before any collective of config 4
    # round 5: what RCCL and HIP themselves saw -- one rank in the communicator, one named device
    assert c4["rccl_ranks"] == 1 and len(c4["devices"]) == 1 and c4["distinct_devices"] == 1 and c4["devices"][0].startswith("uuid=")
    assert c4["log_pitch"] == 262144 and c4["tick_table"] is None and c4["root_share"] is None      # one rank: nothing to balance
    # ... and ONE wrong value among the gathered rows ends in exit status 3 with the line still printed (round-4 VERDICT 3)
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--no-extras"], env=dict(env, UAVAC_BENCH_CORRUPT_GATHER="1", MASTER_PORT="29534"),
                         capture_output=True, text=True, timeout=900)
    assert bad.returncode == 3, (bad.returncode, bad.stderr[-1500:])
    line = json.loads([l for l in bad.stdout.splitlines() if l.startswith("{")][-1])
    assert line["gather_error_kind"] == "mismatch" and line["config4"]["gather_verified"] is False and line["value"] > 0
    assert "end_to_end" not in line["config4"]


def test_rccl_world1_gather_and_loopback(eng, nat):
    """ncclCommInitRank / ncclAllGather / ncclSend + ncclRecv through libuavac.so on one GPU: the calls of the
    8-GPU gather with world = 1 (the root's own block) and with this rank as its own peer (loopback)."""
    import torch
    from uav_ac.fleet import RcclComm
    buf = C.create_string_buffer(nat.COMM_ID_BYTES)
    eng.ctx.call("uavac_comm_unique_id", buf)
    comm = RcclComm(eng, unique_id=bytes(buf.raw), world=1, rank=0)
    try:
        w, r = C.c_int(-1), C.c_int(-1)
        eng.ctx.call("uavac_comm_shape", comm._h, C.byref(w), C.byref(r))
        assert (w.value, r.value) == (1, 0) == comm.shape()
        ident = eng.ctx.device_identity()
        assert ident.startswith("uuid=") and ";pci=" in ident and "gfx950" in ident
        assert comm.counts(12345) == [12345]
        rows = torch.randn((70001, 11), dtype=torch.float64, device=eng.device)
        out, counts = comm.gather_rows(rows, dst=0)
        assert counts == [70001] and out.data_ptr() != rows.data_ptr() and torch.equal(out, rows)
        empty, c0 = comm.gather_rows(rows[:0], dst=0)
        assert c0 == [0] and empty.shape == (0, 11)
        # the transport itself: 96 MB through ncclSend -> ncclRecv, bit for bit
        src = torch.randn((12_000_000,), dtype=torch.float64, device=eng.device)
        assert torch.equal(comm.loopback(src), src)
        # argument checking happens before anything is enqueued
        cnt = (C.c_int64 * 1)(5)
        with pytest.raises(nat.UavacError) as e:
            eng.ctx.call("uavac_gather_rows_dev", comm._h, C.c_void_p(rows.data_ptr()), 5, 11, cnt, 3, C.c_void_p(out.data_ptr()))
        assert e.value.code == nat.EINVAL
        with pytest.raises(nat.UavacError) as e:          # counts[rank] must be this rank's n_rows
            eng.ctx.call("uavac_gather_rows_dev", comm._h, C.c_void_p(rows.data_ptr()), 6, 11, cnt, 0, C.c_void_p(out.data_ptr()))
        assert e.value.code == nat.EINVAL
        with pytest.raises(ValueError):
            comm.gather_rows(rows.float(), dst=0)
    finally:
        comm.close()


def test_gather_rows_refuses_gpu_rows_without_a_communicator(eng):
    import torch
    from uav_ac.comm_host import gather_rows
    with pytest.raises(ValueError):
        gather_rows(torch.zeros((4, 11), dtype=torch.float64, device=eng.device))


# ------------------------------------------------------------------------------------------- one-call planning chain
def test_replan_is_bit_identical_to_the_separate_calls_and_checks_capacity(eng, nat):
    import torch
    from oracle import minsnap_oracle as mo
    wps = mo.synthetic_missions(300, 8)
    plan = eng.plan(wps, 3.0, 0.01)
    ref = {k: getattr(plan, k).clone() for k in ("times", "seg_rows", "row_offsets", "coeffs", "traj", "first_yaw")}
    for k in ("times", "coeffs", "traj", "first_yaw"):
        getattr(plan, k).fill_(float("nan"))
    plan.seg_rows.zero_(); plan.row_offsets.zero_()
    eng.replan(plan)
    assert eng.take_flags() == [0, 0, 0, 0]
    for k, v in ref.items():
        assert torch.equal(getattr(plan, k), v), k
    # a row buffer that is one row short: nothing is written, flag 2 is raised, and reading the flags clears them
    short = plan.traj[:-1].clone()
    short.fill_(-7.0)
    full, plan.traj = plan.traj, short
    eng.replan(plan)
    assert eng.take_flags() == [0, 0, 1, 0]
    assert bool((short == -7.0).all())
    plan.traj = full
    assert eng.take_flags() == [0, 0, 0, 0]


def test_sticky_flags_do_not_leak_into_host_twins_and_plan_raises_on_singular(eng, nat):
    from oracle import minsnap_oracle as mo
    wps = mo.synthetic_missions(4, 3)
    bad = wps.copy()
    bad[2, 2] = bad[2, 1]                         # a repeated waypoint: zero-length segment -> singular knot system
    with pytest.raises(nat.UavacError) as e:
        eng.plan(bad, 3.0, 0.01)
    assert e.value.code == nat.ESINGULAR
    plan = eng.plan(bad, 3.0, 0.01, strict=False)        # the device flag is now set and nobody has read it
    assert plan.status.cpu().tolist() == [0, 0, 1, 0]
    # a host twin on the same context with valid input must not report the stale flag
    coeffs = np.empty((4, 24, 3)); times = np.empty((4, 3))
    eng._bind_stream()
    eng.ctx.call("uavac_minsnap_solve", nat.np_ptr(wps), 4, 3, 3.0, nat.np_ptr(coeffs), nat.np_ptr(times))
    assert np.isfinite(coeffs).all()
    with pytest.raises(nat.UavacError) as e:
        eng.ctx.call("uavac_minsnap_solve", nat.np_ptr(bad), 4, 3, 3.0, nat.np_ptr(coeffs), nat.np_ptr(times))
    assert e.value.code == nat.ESINGULAR
    assert eng.take_flags() == [0, 0, 0, 0]


# ------------------------------------------------------------------------------------------- yaw scan on its own
def test_yaw_scan_matches_reference_on_crafted_and_long_sequences(eng):
    from uav_ac.planning.minimum_snap import MinimumSnap
    g = load_golden("yaws.npz")
    for name in ("hold", "cross_pi", "none_valid", "exact_pi_steps", "leading_invalid", "random_spin"):
        got = MinimumSnap._calculate_yaws(g[name + "_vel"])
        assert got.shape == g[name + "_yaw"].shape
        assert np.allclose(got, g[name + "_yaw"], rtol=0, atol=1e-12), name
    gl = load_golden("yaws_long.npz")
    for name in ("n5000", "n20000"):                      # far beyond the 64 samples the round-1 facade accepted
        got = MinimumSnap._calculate_yaws(gl[name + "_vel"])
        assert np.max(np.abs(got - gl[name + "_yaw"])) < 1e-9 * max(1.0, np.max(np.abs(gl[name + "_yaw"]))), name
    assert MinimumSnap._calculate_yaws(np.zeros((0, 3))).shape == (0,)
    # batched, device resident: several sequences in one launch == each on its own
    vel = np.vstack([gl["n5000_vel"], g["random_spin_vel"], gl["n20000_vel"][:777]])
    offs = np.cumsum([0, 5000, 400, 777])
    y = eng.yaw_scan(vel, offs).cpu().numpy()
    assert np.array_equal(y[:5000], MinimumSnap._calculate_yaws(gl["n5000_vel"]))
    assert np.array_equal(y[5000:5400], MinimumSnap._calculate_yaws(g["random_spin_vel"]))
    assert np.array_equal(y[5400:], MinimumSnap._calculate_yaws(gl["n20000_vel"][:777]))


def test_sampler_yaw_equals_standalone_scan_of_its_own_velocities(eng):
    from oracle import minsnap_oracle as mo
    plan = eng.plan(mo.synthetic_missions(64, 12), 3.0, 0.01, dense_yaw=True)
    y = eng.yaw_scan(plan.traj[:, 3:6].contiguous(), plan.row_offsets)
    import torch
    assert torch.equal(y, plan.traj[:, 9].contiguous()) and torch.equal(y, plan.yaw)


# ------------------------------------------------------------------------------------------- jerk / snap
def test_jerk_and_snap_match_reference_polynom_products(eng):
    g = load_golden("derivatives.npz")
    import torch
    for name, m in (("config1", 4), ("m12_0", 12), ("m12_1", 12), ("m12_2", 12)):
        plan = eng.plan(g[name + "_wp"][None], 3.0, 0.01)
        before = plan.traj.clone()
        jerk, snap = eng.sample_derivatives(plan)
        assert torch.equal(plan.traj, before)                       # the (N, 11) rows are untouched by the extra outputs
        assert jerk.shape == (plan.total_rows, 3) == tuple(g[name + "_jerk"].shape)
        assert col_err(jerk.cpu().numpy(), g[name + "_jerk"]) < TOL, name
        assert col_err(snap.cpu().numpy(), g[name + "_snap"]) < TOL, name
    # against the oracle on the device's own coefficients: rounding-level agreement
    from oracle import minsnap_oracle as mo
    wps = mo.synthetic_missions(3, 8)
    plan = eng.plan(wps, 2.0, 0.02)
    jerk, snap = eng.sample_derivatives(plan)
    for b in range(3):
        lo, hi = int(plan.row_offsets[b]), int(plan.row_offsets[b + 1])
        j_ref, s_ref = mo.sample_jerk_snap(plan.coeffs[b].cpu().numpy(), plan.times[b].cpu().numpy(), 0.02)
        assert col_err(jerk[lo:hi].cpu().numpy(), j_ref) < 1e-10 and col_err(snap[lo:hi].cpu().numpy(), s_ref) < 1e-10


# ------------------------------------------------------------------------------------------- device guard
def test_engine_resolves_device_and_restores_the_callers(nat):
    import torch
    from uav_ac.fleet import Engine
    e = Engine("cuda")                               # no index: the current device (round-1 ADVICE: raised TypeError)
    assert e.device.index == torch.cuda.current_device()
    lib = nat.lib()
    assert lib.uavac_device(e.ctx._h) == e.device.index
    with pytest.raises(nat.UavacError):
        Engine("cpu")
    if torch.cuda.device_count() > 1:               # only on multi-GPU boxes: a ctx for GPU 1 while GPU 0 is current
        from oracle import minsnap_oracle as mo
        torch.cuda.set_device(0)
        e1 = Engine("cuda:1")
        plan = e1.plan(mo.synthetic_missions(8, 3), 3.0, 0.01)
        assert plan.traj.device.index == 1 and torch.cuda.current_device() == 0
        ref = Engine("cuda:0").plan(mo.synthetic_missions(8, 3), 3.0, 0.01)
        assert torch.equal(plan.traj.cpu(), ref.traj.cpu())


# ------------------------------------------------------------------------------------------- aligner launch
def test_rollout_aligner_launch_does_not_change_a_bit(nat):
    """The empty 2-wave kernel in front of a logged rollout launch only changes where the hardware puts the waves:
    logs and final state are bit-identical with and without it."""
    import torch
    from uav_ac.fleet import Engine
    from bench import missions
    B, K = 65536, 300
    wps = missions(B, 4, 0, B)
    logs = []
    for align in (1, 0):
        e = Engine("cuda:0")
        e.ctx.set_option("rollout_align", align)
        plan = e.plan(wps, 3.0, 0.01)
        fleet = e.fleet(plan)
        slog, _ = fleet.rollout(K, state_log=True)
        # (a full chip: the compute wave evaluates the target rows, coefficients through registers -- no ninth argument)
        assert e.ctx.last_rollout_kernel() == "control_rollout_kernel<1, 1, true, false, false, true, false, true, 0>"
        logs.append((slog, fleet.state.clone(), fleet.istate.clone()))
        del fleet, plan
    assert torch.equal(logs[1][0], logs[0][0]) and torch.equal(logs[1][1], logs[0][1]) and torch.equal(logs[1][2], logs[0][2])
    with pytest.raises(nat.UavacError):
        Engine("cuda:0").ctx.set_option("no_such_option", 1)


# ------------------------------------------------------------------------------------------- N3: ground plane
def test_ground_takeoff_matches_oracle_and_free_flight_is_untouched(eng, nat):
    """The lab mission from the reference's true start (on the ground, rotors stopped) with the build-defined ground
    contact: HIP rollout == C oracle (1e-9 over the take-off), bookkeeping bits equal; and with the plane out of reach
    the ground-enabled kernel reproduces the free-flight kernel bit for bit."""
    import torch
    from oracle import c_oracle as cc
    from oracle import minsnap_oracle as mo
    g = load_golden("fixed_missions.npz")
    traj = mo.mission_trajectory(g["lab_wp"], None, 2.0, 0.01)
    K = 6000
    V = nat.Vehicle.default()
    V.ground = 1
    rows = torch.as_tensor(traj, device=eng.device)
    offs = torch.tensor([0, len(traj)], dtype=torch.int64, device=eng.device)
    P = lambda t: C.c_void_p(t.data_ptr())          # noqa: E731
    pos = torch.as_tensor(traj[0:1, 0:3].copy(), device=eng.device)
    state = torch.empty((nat.STATE_ROWS, 1), dtype=torch.float64, device=eng.device)
    istate = torch.empty((nat.ISTATE_ROWS, 1), dtype=torch.int32, device=eng.device)
    log = torch.empty((K, 13, 1), dtype=torch.float64, device=eng.device)
    eng._bind_stream()
    eng.ctx.call("uavac_state_init_dev", C.byref(V), P(pos), 1, 0, P(state), P(istate))
    eng.ctx.call("uavac_control_rollout_dev", C.byref(V), P(rows), P(offs), P(state), P(istate), 1, K, P(log), None, None, 0)
    assert eng.ctx.last_rollout_kernel() == "control_rollout_kernel<1, 1, true, false, false, false, true, false, 0>"
    Vc = cc.Vehicle.default()
    Vc.ground = 1
    s0, i0 = cc.initial_state(traj[0, 0:3], Vc, hover=False)
    s_ref, _ = cc.rollout(traj, s0, i0, K, Vc)
    got = log[:, :, 0].cpu().numpy()
    assert col_err(got, s_ref) < 1e-9
    assert istate[:, 0].cpu().tolist() == i0.tolist() and i0[3] == nat.GROUND_TAKEN_OFF
    assert got[:, 2].max() > -0.0205 and got[-1, 2] < -1.0          # touched down while the rotors spun up, then left
    # plane far below: the ground-enabled kernel must be the free-flight kernel, bit for bit
    wps = mo.synthetic_missions(256, 4)
    plan = eng.plan(wps, 3.0, 0.01)
    Vg = nat.Vehicle.default()
    Vg.ground, Vg.ground_z = 1, 1000.0
    a, b = eng.fleet(plan), eng.fleet(plan, vehicle=Vg)
    la, _ = a.rollout(700, state_log=True)
    lb, _ = b.rollout(700, state_log=True)
    assert torch.equal(la, lb) and torch.equal(a.state[:26], b.state[:26]) and torch.equal(a.istate[:3], b.istate[:3])
    assert bool((b.istate[3] == nat.GROUND_TAKEN_OFF).all()) and bool((a.istate[3] == 0).all())


def test_scene_simulation_ground_invariants_of_the_reference_tests():
    """tests/unit/simulation/test_mujoco_sim.py:123-174 against the scene-reading simulation (HIP dynamics step)."""
    from uav_ac.simulation.mujoco_sim import MujocoSimulation
    start = np.array([1.0, 7.0, -0.021])
    sim = MujocoSimulation()
    assert sim.has_collision is False and sim.ground_z == 0.0 and sim.ground_clearance == 0.02
    for _ in range(50):                                                    # :123-131
        sim.step()
    assert sim.has_collision is True and sim.collision_detected is False
    sim = MujocoSimulation()                                               # :150-160
    z0 = sim.quad.z
    for _ in range(20):
        sim.step()
    assert sim.quad.z > z0 and sim.quad.z_vel > 0.0
    sim = MujocoSimulation()                                               # :163-174
    q = sim.quad
    q.omega[:] = np.sqrt(q.m * q.g / (4 * q.kf))
    for _ in range(100):
        sim.step()
    assert np.allclose(q.position, start, atol=1e-6) and np.allclose(q.velocity, 0.0, atol=1e-6)
    sim = MujocoSimulation()                                               # :134-147
    sim.quad.X[2] = -0.2
    sim.step()
    sim.quad.X[2] = -0.019
    sim.quad.X[7:13] = 0.0
    sim.step()
    assert sim.collision_detected is True


def test_lab_mission_from_the_ground_meets_reference_integration_bounds():
    """tests/integration/test_mujoco_trajectory_tracking.py:34-36 from the reference's true start state: final distance
    and mean tracking error < 0.5 m, no collision (ground contact before take-off does not count, as upstream)."""
    from uav_ac.main import fly_mission
    from uav_ac.simulation.mujoco_sim import DEFAULT_SCENE_PATH
    from uav_ac import _native as nat
    out = fly_mission(DEFAULT_SCENE_PATH, velocity=2.0, frequency=10, settle_ticks=0)
    assert out["distance_to_goal"] < 0.5 and out["mean_tracking_error"] < 0.5 and out["collision_detected"] is False
    assert out["ground_bits"] == nat.GROUND_TAKEN_OFF
    assert out["states"][:, 2].max() > -0.0205                           # it really stood on the plane first
    free = fly_mission(DEFAULT_SCENE_PATH, velocity=2.0, frequency=10, settle_ticks=0, ground=False)
    assert free["distance_to_goal"] < 0.5 and free["ground_bits"] == 0


# ------------------------------------------------------------------------------------------- config 4, one rank's shard
def test_full_size_properties_config4_rank_shard(eng):
    """BASELINE configs[3], the share of one of 8 ranks: 32 768 UAVs, 8-segment missions, 5 000 ticks (5 launches x
    1 000 with the state log).  Size-independent properties + spot lanes against the C oracle."""
    import torch
    from bench import missions
    from oracle import c_oracle as co
    from uav_ac.fleet import shard_bounds
    world, rank = 8, 5
    lo, hi = shard_bounds(262144, rank, world)
    assert hi - lo == 32768
    wps = missions(262144, 8, lo, hi)                    # this rank's contiguous block of the 262 144 missions
    plan = eng.plan(wps, 3.0, 0.01)
    fleet = eng.fleet(plan)
    B, K, CH = 32768, 5000, 1000
    log = torch.empty((CH, 13, B), dtype=torch.float64, device=eng.device)
    spots = [0, 1, 9999, 20000, B - 1]
    kept = []
    sums = []
    for c in range(K // CH):
        fleet.rollout(CH, state_log=log)
        kept.append(log[:, :, spots].clone())
        sums.append(log.sum(dim=(1, 2)))
    slog = torch.cat(kept).cpu().numpy()
    nrows = (plan.row_offsets[1:] - plan.row_offsets[:-1])
    idx = fleet.trajectory_index.long()
    assert bool((idx == torch.clamp(torch.full_like(nrows, K // 10), max=nrows - 1)).all())
    assert bool((fleet.istate[1] == K).all())
    q = fleet.X[3:7]
    finite = torch.isfinite(fleet.state).all(dim=0)
    assert float(finite.double().mean()) == 1.0
    assert float(((q * q).sum(dim=0) - 1).abs().max()) < 1e-9           # attitude stays unit
    # the same flight in one launch without a log ends in the same state, bit for bit
    f2 = eng.fleet(plan)
    f2.rollout(K)
    assert torch.equal(f2.state, fleet.state) and torch.equal(f2.istate, fleet.istate)
    # spot lanes against the scalar C oracle (plan + 5 000 ticks)
    for j, b in enumerate(spots):
        traj, _, _ = co.plan(wps[b], 3.0, 0.01)
        got = plan.mission(b)
        assert got.shape == traj.shape and col_err(got, traj) < TOL
        state, istate = co.initial_state(traj[0, 0:3])
        s_ref, _ = co.rollout(traj, state, istate, K, log_cmd=False)
        assert col_err(slog[:, :, j], s_ref) < TOL, b


# ------------------------------------------------------------------------------------------- resident session
def test_pilot_ticks_equal_the_fused_rollout_bit_for_bit(nat):
    """`uavac_pilot_tick` (controller + vehicle half per call, state in pinned mapped memory) for K calls == one fused
    `uavac_control_rollout` of K ticks on the same three missions, including the obstacle flag; editing the pinned state
    between calls is honoured."""
    from oracle import minsnap_oracle as mo
    ctx = nat.Context(0)
    wps = mo.synthetic_missions(3, 4)
    trajs = [mo.plan(w, 3.0, 0.01, method="solve") for w in wps]
    rows = np.vstack(trajs)
    offs = np.concatenate([[0], np.cumsum([len(t) for t in trajs])]).astype(np.int64)
    V = nat.Vehicle.default()
    K, B = 700, 3
    aabbs = np.array([[-100.0, 100.0, -100.0, 100.0, -3.2, -3.03]])      # a slab just below the start altitude (z = -3)
    state = np.zeros((nat.STATE_ROWS, B)); istate = np.zeros((nat.ISTATE_ROWS, B), np.int32)
    ctx.call("uavac_state_init", C.byref(V), nat.np_ptr(np.ascontiguousarray(wps[:, 0, :])), B, 1, nat.np_ptr(state), nat.np_ptr(istate))
    ref_s, ref_i = state.copy(), istate.copy()
    log = np.empty((K, 13, B))
    ctx.call("uavac_control_rollout", C.byref(V), nat.np_ptr(rows), nat.np_ptr(offs), nat.np_ptr(ref_s), nat.np_ptr(ref_i), B, K,
             nat.np_ptr(log), None, nat.np_ptr(aabbs), 1)
    p = nat.Pilot(ctx, rows, offs)
    p.set_obstacles(aabbs)
    p.state[:] = state
    p.istate[:] = istate
    for k in range(K):
        if k % 2:
            p.tick(V)                                                      # both halves in one call
        else:
            p.tick(V, nat.PILOT_CONTROLLER); p.tick(V, nat.PILOT_DYNAMICS)
        assert np.array_equal(p.state[0:13], log[k])
    assert np.array_equal(p.state, ref_s) and np.array_equal(p.istate, ref_i)
    assert ref_i[2].any()                                                  # somebody did cross the slab
    p.state[2, 1] += 0.25                                                  # the host moves a vehicle: the next tick starts there
    z = p.state[2, 1]
    p.tick(V, nat.PILOT_DYNAMICS)
    assert abs(p.state[2, 1] - z) < 0.01 and p.state[2, 1] != z
    p.close()
    with pytest.raises(nat.UavacError):
        nat.Pilot(ctx, rows, np.array([0, 5, 3], dtype=np.int64))          # offsets must not decrease


def test_plan_chain_edge_sizes_and_optional_outputs(eng, nat):
    """One segment and UAVAC_MAX_SEGMENTS through the one-call chain; the yaw column is optional; jerk / snap of a single
    segment equal the closed form of the rest-to-rest septic."""
    import torch
    from oracle import minsnap_oracle as mo
    for m in (1, nat.MAX_SEGMENTS):
        wps = mo.synthetic_missions(5, m)
        plan = eng.plan(wps, 3.0, 0.01)
        ref = plan.traj.clone()
        plan.traj.zero_()
        eng._bind_stream()
        eng.ctx.call("uavac_minsnap_plan_dev", C.c_void_p(plan.waypoints.data_ptr()), plan.B, plan.m, plan.velocity, plan.dt,
                     C.c_void_p(plan.times.data_ptr()), C.c_void_p(plan.seg_rows.data_ptr()), C.c_void_p(plan.row_offsets.data_ptr()),
                     C.c_void_p(plan.coeffs.data_ptr()), C.c_void_p(plan.status.data_ptr()), C.c_void_p(plan.traj.data_ptr()),
                     int(plan.traj.shape[0]), None, None)
        assert torch.equal(plan.traj, ref) and eng.take_flags() == [0, 0, 0, 0]
        assert col_err(plan.mission(4), mo.plan(wps[4], 3.0, 0.01, method="solve")) < 1e-6
    plan = eng.plan(np.array([[[0.0, 0.0, -1.0], [3.0, 0.0, -1.0]]]), 1.0, 0.01)           # T = 3 * 1.5 = 4.5 s, along x
    jerk, snap = eng.sample_derivatives(plan)
    T, d = 4.5, 3.0
    t = np.arange(0.0, T, 0.01) / T
    # x(tau) = d (35 tau^4 - 84 tau^5 + 70 tau^6 - 20 tau^7): rest-to-rest minimum snap
    j = d / T ** 3 * (840 * t - 5040 * t ** 2 + 8400 * t ** 3 - 4200 * t ** 4)
    s4 = d / T ** 4 * (840 - 10080 * t + 25200 * t ** 2 - 16800 * t ** 3)
    assert np.allclose(jerk[:, 0].cpu().numpy(), j, rtol=1e-9, atol=1e-9) and np.allclose(snap[:, 0].cpu().numpy(), s4, rtol=1e-9, atol=1e-9)
    assert float(jerk[:, 1:].abs().max()) < 1e-12 and float(snap[:, 1:].abs().max()) < 1e-12


# ------------------------------------------------------------------------------------- many obstacles per tick
@pytest.mark.parametrize("n_obs", [1, 8, 9, 10, 11, 14])
def test_obstacle_flags_with_more_obstacles_than_registers_hold(eng, n_obs):
    """The store wave keeps 8 obstacles in registers, the compute wave 10 in the lanes of one register pair; the rest take
    the scalar-load path.  Only the LAST obstacle of the list is where vehicles fly (the others are far away), so a flag
    can only come from the last slot of whichever path holds it: logged, unlogged and a recomputation from the log agree."""
    import torch
    from oracle import minsnap_oracle as mo
    B, K = 333, 1500
    wps = mo.synthetic_missions(B, 4)
    boxes = np.array([[100.0 + o, 101.0 + o, 100.0, 101.0, -50.0, -49.0] for o in range(n_obs)])
    boxes[-1] = [0.0, 30.0, 0.0, 7.0, -3.5, -2.5]
    logged, unlogged = (eng.fleet(eng.plan(wps, 3.0, 0.01)) for _ in range(2))
    slog, _ = logged.rollout(K, state_log=True, aabbs=boxes)
    unlogged.rollout(K, aabbs=boxes)
    c = torch.as_tensor(boxes[-1], device=slog.device)
    inside = ((slog[:, 0] >= c[0]) & (slog[:, 0] <= c[1]) & (slog[:, 1] >= c[2]) & (slog[:, 1] <= c[3]) &
              (slog[:, 2] >= c[4]) & (slog[:, 2] <= c[5])).any(dim=0)
    assert torch.equal(logged.collided.bool(), inside) and torch.equal(unlogged.collided.bool(), inside)
    assert 0 < int(inside.sum()) < B
    assert torch.equal(logged.state, unlogged.state)


# ------------------------------------------------------------------------------------- ragged batches
def test_ragged_batch_equals_uniform_plans_bit_for_bit(eng, nat):
    """Missions of different lengths in one call (uavac_minsnap_*_ragged_dev): every mission's durations, row counts,
    coefficients, rows, first heading and status equal those of the uniform entry points on a batch of its own length --
    same kernels, same arithmetic.  Lengths are mixed inside every wave of the lane-per-mission solver (1 .. 64 segments),
    the batch ends in a ragged tail, and the per-spline hit flags equal a recomputation from the rows."""
    import torch
    from oracle import minsnap_oracle as mo
    lengths = [1, 2, 3, 5, 8, 12, 20, 33, 64]
    rng = np.random.default_rng(7)
    ms = rng.choice(lengths, size=203)
    ms[:9] = lengths                                             # each length at least once
    pools = {m: mo.synthetic_missions(int((ms == m).sum()), m) for m in lengths}
    taken = {m: 0 for m in lengths}
    missions, where = [], []
    for m in ms:
        missions.append(pools[m][taken[m]]); where.append((int(m), taken[m])); taken[m] += 1
    cub = np.array([0.0, 30.0, 0.0, 7.0, -3.5, -2.5])
    rb = eng.plan_ragged(missions, 3.0, 0.01, cuboid=cub)
    assert rb.B == 203 and rb.max_m == 64 and int(rb.seg_offsets_host[-1]) == int(ms.sum())
    uniform = {m: eng.plan(pools[m], 3.0, 0.01) for m in lengths}
    ro = rb.row_offsets.cpu().numpy()
    hit = rb.hit.cpu().numpy()
    for b, (m, j) in enumerate(where):
        u = uniform[m]
        s0, s1 = int(rb.seg_offsets_host[b]), int(rb.seg_offsets_host[b + 1])
        assert s1 - s0 == m
        assert torch.equal(rb.times[s0:s1], u.times[j]) and torch.equal(rb.seg_rows[s0:s1], u.seg_rows[j])
        assert torch.equal(rb.coeffs[s0:s1].reshape(-1, 3), u.coeffs[j])
        uo = u.row_offsets[j:j + 2].cpu().numpy()
        assert ro[b + 1] - ro[b] == uo[1] - uo[0]
        rows = rb.traj[ro[b]:ro[b + 1]]
        assert torch.equal(rows, u.traj[uo[0]:uo[1]])
        assert float(rb.first_yaw[b]) == float(u.first_yaw[j])
        inside = ((rows[:, 0] >= cub[0]) & (rows[:, 0] <= cub[1]) & (rows[:, 1] >= cub[2]) & (rows[:, 1] <= cub[3]) &
                  (rows[:, 2] >= cub[4]) & (rows[:, 2] <= cub[5]))
        want = np.zeros(m, dtype=bool)
        want[np.unique(rows[inside][:, 10].cpu().numpy().astype(int))] = True
        assert np.array_equal(hit[s0:s1].astype(bool), want), b
    assert int(rb.status.sum()) == 0 and 0 < hit.sum() < len(hit)
    assert eng.take_flags() == [0, 0, 0, 0]


def test_ragged_entry_points_reject_bad_arguments(eng, nat):
    import torch
    from oracle import minsnap_oracle as mo
    with pytest.raises(ValueError):
        eng.plan_ragged([], 3.0, 0.01)
    with pytest.raises(ValueError):
        eng.plan_ragged([np.zeros((1, 3))], 3.0, 0.01)             # a path needs two waypoints
    with pytest.raises(ValueError):
        eng.plan_ragged([np.zeros((nat.MAX_SEGMENTS + 2, 3))], 3.0, 0.01)
    rb = eng.plan_ragged([mo.synthetic_missions(1, 3)[0], mo.synthetic_missions(1, 5)[0]], 3.0, 0.01)
    P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())
    def code(fn, *a):
        with pytest.raises(nat.UavacError) as e:
            eng.ctx.call(fn, *a)
        return e.value.code
    assert code("uavac_minsnap_row_counts_ragged_dev", P(rb.waypoints), P(None), 2, 5, 3.0, 0.01, P(rb.times), P(rb.seg_rows),
                P(rb.row_offsets)) == nat.EINVAL
    assert code("uavac_minsnap_row_counts_ragged_dev", P(rb.waypoints), P(rb.seg_offsets), 2, 65, 3.0, 0.01, P(rb.times),
                P(rb.seg_rows), P(rb.row_offsets)) == nat.EINVAL
    assert code("uavac_minsnap_solve_ragged_dev", P(rb.waypoints), P(rb.times), P(None), 2, 5, P(rb.coeffs), P(None)) == nat.EINVAL
    args = [P(rb.coeffs), P(rb.seg_rows), P(rb.seg_offsets), P(rb.row_offsets), 2, 5, 8, 0.01, P(rb.traj), rb.total_rows]
    assert code("uavac_minsnap_sample_ragged_dev", *args, P(rb.first_yaw), P(None), P(None)) == nat.EINVAL     # cuboid without flags
    args[6] = 1                                                                                               # fewer segments than missions
    assert code("uavac_minsnap_sample_ragged_dev", *args, P(None), P(None), P(None)) == nat.EINVAL
    # a segment count beyond max_m is clamped and reported by the sticky flag, never followed out of bounds
    eng.take_flags()
    eng.ctx.call("uavac_minsnap_row_counts_ragged_dev", P(rb.waypoints), P(rb.seg_offsets), 2, 4, 3.0, 0.01, P(rb.times),
                 P(rb.seg_rows), P(rb.row_offsets))
    assert eng.take_flags()[0] == 1
    # a row buffer that is too small: nothing is written, flag 2
    small = torch.zeros((10, nat.TRAJ_COLS), dtype=torch.float64, device=eng.device)
    rb2 = eng.plan_ragged([mo.synthetic_missions(1, 3)[0]], 3.0, 0.01)
    eng.ctx.call("uavac_minsnap_sample_ragged_dev", P(rb2.coeffs), P(rb2.seg_rows), P(rb2.seg_offsets), P(rb2.row_offsets), 1, 3,
                 3, 0.01, P(small), 10, P(None), P(None), P(None))
    assert eng.take_flags()[2] == 1 and float(small.abs().sum()) == 0.0


def test_fleet_flies_a_ragged_batch_like_uniform_plans(eng):
    """A RaggedBatch feeds the rollout through its rows (the default below 40 960 vehicles) or through its coefficients
    (plan-fed, uavac_control_rollout_plan_ragged_dev): every vehicle's log equals, bit for bit, the one it gets in a uniform
    batch of its own length; split launches and single ticks included."""
    import torch
    from oracle import minsnap_oracle as mo
    a, b = mo.synthetic_missions(40, 3), mo.synthetic_missions(30, 6)
    missions = [a[i // 2] if i % 2 == 0 else b[i // 2] for i in range(60)]          # interleaved: 30 of each
    rb = eng.plan_ragged(missions, 3.0, 0.01)
    fl = eng.fleet(rb)
    assert not fl.from_plan
    log, _ = fl.rollout(1200, state_log=True)
    fed = eng.fleet(rb, from_plan=True)
    flog1, _ = fed.rollout(450, state_log=True)
    fed.rollout(1)
    flog2, _ = fed.rollout(749, state_log=True)
    assert torch.equal(flog1, log[:450]) and torch.equal(flog2, log[451:])
    assert torch.equal(fed.state[:26], fl.state[:26]) and torch.equal(fed.istate, fl.istate)
    with pytest.raises(ValueError):
        eng.fleet(rb, from_plan=True, yaw_from="column").rollout(1)
    la, _ = eng.fleet(eng.plan(a[:30], 3.0, 0.01), from_plan=False).rollout(1200, state_log=True)
    lb, _ = eng.fleet(eng.plan(b[:30], 3.0, 0.01), from_plan=False).rollout(1200, state_log=True)
    assert torch.equal(log[:, :, 0::2], la) and torch.equal(log[:, :, 1::2], lb)


def test_row_buffer_placement_trials_keep_the_rows(eng):
    """Engine.plan(..., placement_trials=3) times the sampler on three candidate row buffers and keeps one: same rows as a
    plain plan, two or three recorded times (it stops at the first clearly faster candidate), the kept buffer is the fastest."""
    import torch
    from oracle import minsnap_oracle as mo
    wps = mo.synthetic_missions(500, 6)
    plain = eng.plan(wps, 3.0, 0.01)
    placed = eng.plan(wps, 3.0, 0.01, placement_trials=3)
    assert plain.placement_ms is None and 2 <= len(placed.placement_ms) <= 3 and min(placed.placement_ms) > 0
    assert torch.equal(plain.traj, placed.traj) and torch.equal(plain.first_yaw, placed.first_yaw)
    eng.replan(placed)
    assert torch.equal(plain.traj, placed.traj)


def test_row_buffer_search_holds_two_buffers_at_most_and_the_pool_pays_it_once():
    """Round-3 VERDICT 5.  `placement_trials` draws row buffers ONE AFTER THE OTHER (best so far + one candidate alive: peak 2x the
    row memory, not trials x); `pool=True` keeps the buffer that was found in the Engine and hands it to every later pooled plan
    of at most that many rows without searching again; a larger pooled plan replaces it.  Rows as a plain plan's, always."""
    import torch
    from oracle import minsnap_oracle as mo
    from uav_ac.fleet import Engine
    eng = Engine("cuda:0")
    wps = mo.synthetic_missions(12000, 8)
    plain = eng.plan(wps, 3.0, 0.01)
    row_bytes = plain.traj.numel() * 8
    assert row_bytes > 5e8
    plain_rows = plain.traj.clone().cpu()
    del plain
    torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    before = torch.cuda.memory_allocated()
    eng.FAST_ROW_BUFFER_FRACTION_OF_PEAK = 2.0               # unreachable: every draw is taken
    first = eng.plan(wps, 3.0, 0.01, placement_trials=5, pool=True)
    peak = torch.cuda.max_memory_allocated() - before
    assert len(first.placement_ms) == 5 and first.pooled
    assert peak < 2.0 * row_bytes + 0.25 * row_bytes, (peak, row_bytes)      # two row buffers + the plan's small arrays
    assert torch.equal(first.traj.cpu(), plain_rows)
    # the same fleet planned again, and a smaller one: the pooled buffer, no search
    again = eng.plan(wps, 3.0, 0.01, placement_trials=5, pool=True)
    small = eng.plan(wps[:5000], 3.0, 0.01, placement_trials=5, pool=True)
    assert again.placement_ms is None and small.placement_ms is None
    assert again.traj.data_ptr() == first.traj.data_ptr() == small.traj.data_ptr()
    assert torch.equal(small.traj.cpu(), plain_rows[: small.total_rows])
    eng.sample(again)
    assert torch.equal(again.traj.cpu(), plain_rows)
    # a plan that does not fit the pool gets (and becomes) a new one; unpooled plans never touch it
    big = eng.plan(mo.synthetic_missions(15000, 8), 3.0, 0.01, pool=True)
    other = eng.plan(wps, 3.0, 0.01)
    assert big.traj.data_ptr() != first.traj.data_ptr() and eng._row_pool.data_ptr() == big.traj.data_ptr()
    assert other.traj.data_ptr() not in (big.traj.data_ptr(), first.traj.data_ptr()) and not other.pooled


def test_host_pointer_ragged_plan_equals_the_device_path(eng, nat):
    """uavac_minsnap_plan_ragged (host buffers, one call): sizing call with traj = NULL, then the full call; equals
    Engine.plan_ragged on the same missions bit for bit; bad segment tables and a short row buffer are refused."""
    from oracle import minsnap_oracle as mo
    missions = [mo.synthetic_missions(1, m)[0] for m in (3, 1, 7, 12, 2)]
    rb = eng.plan_ragged(missions, 3.0, 0.01)
    wp = np.ascontiguousarray(np.concatenate(missions))
    so = np.ascontiguousarray(rb.seg_offsets_host)
    S, B = int(so[-1]), len(missions)
    ro = np.zeros(B + 1, np.int64)
    null = C.c_void_p(0)
    eng.ctx.call("uavac_minsnap_plan_ragged", nat.np_ptr(wp), nat.np_ptr(so), B, 3.0, 0.01, null, nat.np_ptr(ro), null, null, 0)
    assert np.array_equal(ro, rb.row_offsets.cpu().numpy())
    times, coeffs, traj = np.empty(S), np.empty((S, 8, 3)), np.empty((int(ro[-1]), nat.TRAJ_COLS))
    eng.ctx.call("uavac_minsnap_plan_ragged", nat.np_ptr(wp), nat.np_ptr(so), B, 3.0, 0.01, nat.np_ptr(times), nat.np_ptr(ro),
                 nat.np_ptr(coeffs), nat.np_ptr(traj), len(traj))
    assert np.array_equal(times, rb.times.cpu().numpy()) and np.array_equal(coeffs, rb.coeffs.cpu().numpy())
    assert np.array_equal(traj, rb.traj.cpu().numpy())
    with pytest.raises(nat.UavacError) as e:                        # a row buffer one row short
        eng.ctx.call("uavac_minsnap_plan_ragged", nat.np_ptr(wp), nat.np_ptr(so), B, 3.0, 0.01, null, nat.np_ptr(ro), null,
                     nat.np_ptr(traj), len(traj) - 1)
    assert e.value.code == nat.EINVAL
    bad = so.copy(); bad[2] = bad[1]                                # a mission without a segment
    with pytest.raises(nat.UavacError) as e:
        eng.ctx.call("uavac_minsnap_plan_ragged", nat.np_ptr(wp), nat.np_ptr(bad), B, 3.0, 0.01, null, nat.np_ptr(ro), null, null, 0)
    assert e.value.code == nat.EINVAL
    wp2 = wp.copy(); wp2[3, 1] = np.nan
    with pytest.raises(nat.UavacError) as e:
        eng.ctx.call("uavac_minsnap_plan_ragged", nat.np_ptr(wp2), nat.np_ptr(so), B, 3.0, 0.01, null, nat.np_ptr(ro), null, null, 0)
    assert e.value.code == nat.ENONFINITE


# ------------------------------------------------------------------------------------- vehicles other than Table V
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_rollout_with_another_vehicle_matches_the_c_oracle(eng, nat, seed):
    """Every field of the vehicle (mass, inertia, arm, rotor constants, thrust limits, motor time constants, flight limits,
    all eleven gains, tick lengths) drawn away from Table V: the fused kernel -- which keeps these constants in a mixture of
    scalar and vector registers and folds several of them into reciprocals and products -- against the scalar C oracle,
    row-fed and plan-fed, with both logs."""
    import torch
    from oracle import c_oracle as cc
    from oracle import minsnap_oracle as mo
    rng = np.random.default_rng(seed)
    V, Vc = nat.Vehicle.default(), cc.Vehicle.default()
    def both(name, value):
        setattr(V, name, value); setattr(Vc, name, value)
    F = int(rng.choice([5, 8, 10]))
    both("dt", float(rng.choice([0.001, 0.002]))); both("inner_per_outer", F); both("dt_outer", V.dt * F)
    both("mass", float(rng.uniform(0.35, 0.9)))
    for i, s in enumerate(rng.uniform(0.7, 1.6, 3)):
        V.inertia[i] *= s; Vc.inertia[i] = V.inertia[i]
    for name in ("arm", "kf", "kappa", "tau_rise", "tau_fall", "max_ascent", "max_descent", "max_speed_xy", "max_horiz_accel",
                 "kp_xy", "kd_xy", "kp_z", "kd_z", "ki_z", "kp_roll", "kp_pitch", "kp_yaw", "kp_p", "kp_q", "kp_r"):
        both(name, getattr(V, name) * float(rng.uniform(0.8, 1.25)))
    both("max_tilt", float(rng.uniform(0.5, 0.8)))
    both("min_thrust", float(rng.uniform(0.05, 0.15))); both("max_thrust", V.mass * 9.81 / 4 * float(rng.uniform(2.5, 4.0)))
    plan = eng.plan(mo.synthetic_missions(96, 6), 2.0, V.dt_outer)
    K = 1500
    for from_plan in (False, True):
        fleet = eng.fleet(plan, vehicle=V, from_plan=from_plan)
        slog, clog = fleet.rollout(K, state_log=True, cmd_log=True)
        assert bool(torch.isfinite(slog).all())
        for b in (0, 41, 95):
            traj = plan.mission(b)
            state, istate = cc.initial_state(traj[0, 0:3], Vc)
            s_ref, c_ref = cc.rollout(traj, state, istate, K, Vc)
            assert col_err(slog[:, :, b].cpu().numpy(), s_ref) < 1e-9, (from_plan, b)
            assert col_err(clog[:, :, b].cpu().numpy(), c_ref) < 1e-9, (from_plan, b)
            assert int(fleet.trajectory_index[b]) == istate[0]


def test_gather_beside_the_rollout_world_1(eng, nat):
    """gather_rows_begin / gather_finish: the transfer of the (final) trajectories on a second stream while the vehicles
    fly on the first -- same gathered rows as the synchronous call, and the flight is not disturbed."""
    import torch
    from oracle import minsnap_oracle as mo
    from uav_ac.fleet import RcclComm
    buf = C.create_string_buffer(nat.COMM_ID_BYTES)
    eng.ctx.call("uavac_comm_unique_id", buf)
    comm = RcclComm(eng, unique_id=bytes(buf.raw), world=1, rank=0)
    try:
        plan = eng.plan(mo.synthetic_missions(3000, 8), 3.0, 0.01)
        ref_fleet, fleet = eng.fleet(plan), eng.fleet(plan)
        ref_log, _ = ref_fleet.rollout(600, state_log=True)
        want, counts = comm.gather_rows(plan.traj, dst=0)
        side = torch.cuda.Stream(device=eng.device)
        eng.replan(plan)                                           # rows rewritten on the main stream ...
        ticket = comm.gather_rows_begin(plan.traj, dst=0, stream=side)      # ... and gathered once that is done
        log, _ = fleet.rollout(600, state_log=True)                # enqueued behind the replan, beside the gather
        got, counts2 = comm.gather_finish(ticket)
        torch.cuda.synchronize()
        assert counts2 == counts == [plan.total_rows] and torch.equal(got, want) and got.data_ptr() != plan.traj.data_ptr()
        assert torch.equal(log, ref_log)
    finally:
        comm.close()
