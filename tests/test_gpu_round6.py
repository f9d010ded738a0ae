"""GPU tests of round 6: the ROWS-FREE planning chain (uavac_minsnap_plan_dev with traj = NULL) and the kernel behind it,
uavac_minsnap_first_yaw_dev -- the heading a mission's leading rows take (MinimumSnap._calculate_yaws, minimum_snap.py:126-136
upstream), computed from coefficients and row counts alone, bit for bit what the sampler writes beside its rows."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
FUZZ_ITERS = max(4, int(os.environ.get("UAVAC_FUZZ_ITERS", "4")))      # a few draws by default; UAVAC_FUZZ_ITERS=<n> for a soak


@pytest.fixture(scope="module")
def eng():
    from uav_ac.fleet import Engine
    return Engine("cuda:0")


def _fuzz_missions(rng, n, segments=(1, 2, 3, 4, 6, 8, 12, 17), vertical=0.15):
    """tests/test_gpu_fuzz.py's generator with more ways not to have a heading: vertical first legs (several in a row), purely
    vertical missions (no heading at all: first_yaw = 0)."""
    out = []
    for _ in range(n):
        m = int(rng.choice(segments))
        d = rng.standard_normal((m, 3)) * np.array([1, 1, rng.choice([0.0, 0.25, 1.0])])
        u = rng.random()
        if u < vertical:
            k = int(rng.integers(1, m + 1))                       # the first k legs are vertical: no heading for k legs' rows
            d[:k, :2] = 0.0
            d[:k, 2] = rng.choice([-1.0, 1.0], size=k)
        elif u < vertical + 0.03:
            d[:, :2] = 0.0                                        # never a heading
            d[:, 2] = -1.0
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        L = rng.uniform(1.0, 6.0, (m, 1))
        w0 = np.array([rng.uniform(0, 24), rng.uniform(0, 14), -rng.uniform(1, 5)])
        out.append(np.concatenate([w0[None], w0 + np.cumsum(L * d, axis=0)]))
    return out


def _same_plan(a, b, torch):
    for k in ("times", "seg_rows", "row_offsets", "coeffs", "status", "first_yaw"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert a.total_rows == b.total_rows


@pytest.mark.parametrize("m", [1, 2, 8, 12, 20])
def test_rows_free_plan_equals_the_full_plan_on_the_reference_goldens(eng, m):
    """Golden missions of the reference (tests/golden/synthetic_missions.npz): the rows-free chain returns the full chain's
    durations, row counts, offsets, coefficients and first headings bit for bit, and the first heading is the yaw the reference
    gives the mission's first row (rows before the first valid sample take its heading) to 1e-5."""
    import torch
    g = load_golden("synthetic_missions.npz")
    wps = g[f"m{m}_wp"]
    full = eng.plan(wps, 3.0, 0.01)
    free = eng.plan(wps, 3.0, 0.01, rows=False)
    assert free.traj is None and free.yaw is None
    _same_plan(full, free, torch)
    assert np.array_equal(free.seg_rows.cpu().numpy(), g[f"m{m}_rows_per_segment"])
    fy = free.first_yaw.cpu().numpy()
    for b in range(len(wps)):
        key = f"m{m}_traj{b}"
        if key in g:
            assert abs(fy[b] - g[key][0, 9]) < 1e-5, (m, b)
    assert torch.equal(eng.first_yaw(full), full.first_yaw)     # the kernel on its own, from a full plan's coefficients
    # the rows arrive later, the very same ones
    eng.sample_rows(free)
    assert torch.equal(free.traj, full.traj)


def test_rows_free_plan_on_the_fixed_missions(eng):
    """config 1 (5 hard-coded waypoints), the lab course's take-off (vertical: no heading at all -> 0) and its course."""
    import torch
    g = load_golden("fixed_missions.npz")
    for wp, ref_rows in ((g["config1_wp"], g["config1_traj"]), (g["lab_wp"][:2], g["lab_traj_free"][:1]), (g["lab_wp"][1:], None)):
        full = eng.plan(wp[None], 3.0, 0.01)
        free = eng.plan(wp[None], 3.0, 0.01, rows=False)
        _same_plan(full, free, torch)
        if ref_rows is not None:
            assert abs(float(free.first_yaw[0]) - ref_rows[0, 9]) < 1e-5
    take_off = eng.plan(g["lab_wp"][None, :2], 3.0, 0.01, rows=False)
    assert float(take_off.first_yaw[0]) == 0.0


@pytest.mark.parametrize("it", range(FUZZ_ITERS))
def test_first_yaw_kernel_on_fuzzed_ragged_batches(eng, it):
    """Ragged batches with vertical first legs (the first valid sample lies whole 64-row items into the mission), purely
    vertical missions, sample periods from 1 ms (thousands of rows before a heading) to 20 ms: the kernel's first headings equal
    the sampler's bit for bit, and NumPy's on the sampled velocities (first row with |v_xy| >= 1e-3) to 1e-12."""
    import torch
    rng = np.random.default_rng(6100 + it)
    dt = (0.001, 0.005, 0.01, 0.02)[it % 4]
    B = int(rng.integers(40, 120 if it % 4 == 0 else 400))      # (1 ms samples: up to 7 500 rows per spline)
    velocity = float(rng.uniform(0.8, 4.0))
    missions = _fuzz_missions(rng, B, vertical=0.3)
    full = eng.plan_ragged(missions, velocity, dt)
    free = eng.plan_ragged(missions, velocity, dt, rows=False)
    assert free.traj is None and free.total_rows == full.total_rows
    for k in ("times", "seg_rows", "row_offsets", "coeffs", "first_yaw"):
        assert torch.equal(getattr(full, k), getattr(free, k)), k
    assert torch.equal(eng.first_yaw(full), full.first_yaw)
    rows = full.traj.cpu().numpy()
    ro = full.row_offsets.cpu().numpy()
    fy = free.first_yaw.cpu().numpy()
    late = 0
    for b in range(B):
        v = rows[ro[b]:ro[b + 1], 3:5]
        valid = np.flatnonzero(np.sqrt(v[:, 0] ** 2 + v[:, 1] ** 2) >= 1e-3)
        want = np.arctan2(v[valid[0], 1], v[valid[0], 0]) if len(valid) else 0.0
        assert abs(fy[b] - want) < 1e-12, (it, b)
        late += int(len(valid) > 0 and valid[0] >= 64)
        assert fy[b] == rows[ro[b], 9]                          # what the mission's first row holds
    if it == 0:
        assert late > 0                                         # 1 ms samples: the walk past the first 64-row item was exercised


def test_first_yaw_kernel_on_a_full_batch_and_edge_shapes(eng):
    """65 536 synthetic missions of the bench's shape (m = 12) and 32 768 of config 4's (m = 8): kernel == sampler for every
    mission; B not a multiple of the workgroup's four missions; the longest missions the ABI takes (64 segments)."""
    import torch
    from oracle import minsnap_oracle as mo
    for B, m in ((65536, 12), (32768, 8), (1, 1), (5, 3), (1001, 64)):
        full = eng.plan(mo.synthetic_missions(B, m), 3.0, 0.01)
        assert torch.equal(eng.first_yaw(full), full.first_yaw), (B, m)
        assert int((full.first_yaw != 0).sum()) > 0.99 * B
        del full


def test_rows_free_plans_fly_and_replan_like_full_plans(eng):
    """State logs of a fleet on a rows-free plan == logs of the same fleet on the full plan (plan-fed and row-fed), through a
    re-plan under the flying fleet as well; single ticks, the row sampler and mission() refuse a plan without rows; a dense
    yaw column cannot be asked for without rows."""
    import torch
    from uav_ac import _native as nat
    from oracle import minsnap_oracle as mo
    wps, wps2 = mo.synthetic_missions(700, 8), mo.synthetic_missions(1400, 8)[700:]
    full = eng.plan(wps, 3.0, 0.01)
    bigger = eng.plan(wps2, 3.0, 0.01)
    if bigger.total_rows > full.total_rows:                     # the re-plan below must fit the full plan's row buffer
        wps, wps2, full = wps2, wps, bigger
    del bigger
    free = eng.plan(wps, 3.0, 0.01, rows=False)
    K = 1200
    logs = []
    for plan, from_plan in ((full, False), (full, True), (free, None)):
        fleet = eng.fleet(plan, from_plan=from_plan)
        assert fleet.from_plan == (from_plan is not False)
        s, c = fleet.rollout(K, state_log=True, cmd_log=True)
        logs.append((s.clone(), c.clone(), fleet.state[:26].clone(), fleet.istate.clone()))
    for other in logs[1:]:
        for x, y in zip(logs[0], other):
            assert torch.equal(x, y)
    # re-plan in place (other waypoints), fleets keep flying: the carried yaw scan is rebuilt, both plans agree again
    f_full, f_free = eng.fleet(full, from_plan=True), eng.fleet(free)
    for f in (f_full, f_free):
        f.rollout(300)
    full.waypoints.copy_(torch.as_tensor(wps2, device=eng.device))
    free.waypoints.copy_(torch.as_tensor(wps2, device=eng.device))
    eng.replan(full)
    eng.replan(free)
    assert eng.take_flags() == [0, 0, 0, 0]
    _same_plan(full, free, torch)
    for f in (f_full, f_free):
        f.reset()
    a, _ = f_full.rollout(500, state_log=True)
    b, _ = f_free.rollout(500, state_log=True)
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        f_free.step()
    with pytest.raises(ValueError):
        eng.fleet(free, from_plan=False)
    with pytest.raises(ValueError):
        eng.sample(free)
    with pytest.raises(ValueError):
        free.mission(0)
    with pytest.raises(ValueError):
        eng.plan(wps, 3.0, 0.01, rows=False, dense_yaw=True)
    # the C ABI itself: a yaw column without rows is EINVAL; a NULL first_yaw is fine (times / counts / coefficients only)
    yaw = torch.empty((free.total_rows,), dtype=torch.float64, device=eng.device)
    args = [C.c_void_p(free.waypoints.data_ptr()), free.B, free.m, 3.0, 0.01] + \
           [C.c_void_p(getattr(free, k).data_ptr()) for k in ("times", "seg_rows", "row_offsets", "coeffs", "status")]
    rc = nat.lib().uavac_minsnap_plan_dev(eng.ctx._h, *args, None, 0, C.c_void_p(yaw.data_ptr()), None)
    assert rc == nat.EINVAL
    before = free.coeffs.clone()
    free.coeffs.zero_()
    rc = nat.lib().uavac_minsnap_plan_dev(eng.ctx._h, *args, None, 0, None, None)
    assert rc == nat.OK
    torch.cuda.synchronize()
    assert torch.equal(free.coeffs, before)


def test_a_singular_mission_has_no_first_heading(eng):
    """A repeated waypoint makes the knot system singular: NaN coefficients, no sample has a heading -> first_yaw 0 from the
    kernel as from the sampler, status 1 either way."""
    import torch
    from oracle import minsnap_oracle as mo
    wps = mo.synthetic_missions(9, 4)
    wps[3, 2] = wps[3, 1]
    full = eng.plan(wps, 3.0, 0.01, strict=False)
    free = eng.plan(wps, 3.0, 0.01, strict=False, rows=False)
    assert int(free.status[3]) == 1 and float(free.first_yaw[3]) == 0.0
    assert torch.equal(full.first_yaw, free.first_yaw) and torch.equal(full.status, free.status)
    eng.take_flags()


def test_the_loaded_library_is_the_one_the_build_checks_saw(eng):
    """The output-only operands of the row prefetch (control_rollout.hip row_issue) and of the sampler's heading prefetch
    (minsnap_yaw.h HeadingFromLds) are only as safe as the disassembly checks of uav_ac/_buildcheck.py.  Those run wherever the
    library is built -- __graft_entry__.build(), the autobuild of uav_ac._native.lib() -- and leave lib/libuavac.buildcheck.json:
    here, ON THE GPU BOX, the library this process loaded must be the very file they passed on (sha256), built by the compiler
    they name; when the object files travelled with it, the checks run once more against this box's LLVM tools."""
    import os
    from conftest import PKG
    from uav_ac import _buildcheck
    from uav_ac import _native as nat
    stamp = _buildcheck.read_stamp()
    assert stamp is not None, "lib/libuavac.buildcheck.json is missing: this library never passed its build checks"
    assert stamp["library_sha256"] == _buildcheck.library_sha256(nat.LIB_PATH), "libuavac.so is not the file the build checks passed on"
    info = nat.lib().uavac_build_info().decode()
    assert stamp["build_info"] == info
    compiler = info.split("; ")[-1].lower().replace("amd clang", "").strip()       # "22.0.0git (https://... roc-7.2.0 ...)"
    assert compiler.split(" ")[0] in stamp["checked_with"].lower(), (info, stamp["checked_with"])
    assert stamp["checks"]["row_prefetch"] == 16 and stamp["checks"]["heading_prefetch"] >= 20
    if os.path.exists(os.path.join(PKG, "build", "control_rollout.o")) and os.path.exists(os.path.join(_buildcheck.LLVM_BIN, "llvm-objdump")):
        again = _buildcheck.run_all(write_stamp=False)
        assert again["library_sha256"] == stamp["library_sha256"] and again["checks"] == stamp["checks"]


def test_sampling_by_ranges_in_any_order_leaves_the_same_rows(eng):
    """Engine.sample_range: missions [b0, b1) of a plan through the same sampler on offset pointers (row offsets are absolute).
    Any cover of the batch, in any order, with empty ranges in between, writes the rows, the first headings and a dense yaw
    column of one sample(plan) bit for bit -- what the root of a pipelined plan gather relies on."""
    import torch
    from oracle import minsnap_oracle as mo
    rng = np.random.default_rng(61)
    for B, m, dense in ((3000, 8, False), (777, 12, True), (5, 2, False)):
        plan = eng.plan(mo.synthetic_missions(B, m), 3.0, 0.01, dense_yaw=dense)
        want = {k: getattr(plan, k).clone() for k in ("traj", "first_yaw") + (("yaw",) if dense else ())}
        for waves in (4, 1):                                      # the chunk-streaming sampler and the one-wave-per-mission one
            eng.ctx.set_option("sampler_waves", waves)
            for k in want:
                getattr(plan, k).fill_(float("nan"))
            cuts = sorted({0, B} | {int(x) for x in rng.integers(0, B + 1, size=6)})
            ranges = list(zip(cuts, cuts[1:])) + [(cuts[1], cuts[1])]
            for i in rng.permutation(len(ranges)):
                eng.sample_range(plan, *ranges[int(i)])
            for k, v in want.items():
                assert torch.equal(getattr(plan, k), v), (B, m, waves, k)
        eng.ctx.set_option("sampler_waves", 4)
    with pytest.raises(ValueError):
        eng.sample_range(plan, 3, 2)
    with pytest.raises(ValueError):
        eng.sample_range(eng.plan(mo.synthetic_missions(4, 2), 3.0, 0.01, rows=False), 0, 4)


def test_pipelined_plan_gather_world1_equals_the_one_shot_gather(eng):
    """RcclComm.gather_plan(parts=...) at world 1 over RCCL behind the C ABI: durations and row counts first, the coefficients in
    parts on a transfer stream (uavac_gather_plan_part_dev: here the root's own block, device-to-device), every part sampled
    behind its event.  Same Plan as the one-shot gather, bit for bit, from a full and from a rows-free plan, with the default
    shares and with others; part arguments are validated before anything is enqueued."""
    import torch
    from uav_ac import _native as nat
    from uav_ac.fleet import RcclComm
    from uav_ac.sharding import PIPELINE_SHARES, part_bounds
    from oracle import minsnap_oracle as mo
    buf = C.create_string_buffer(nat.COMM_ID_BYTES)
    eng.ctx.call("uavac_comm_unique_id", buf)
    comm = RcclComm(eng, unique_id=bytes(buf.raw), world=1, rank=0)
    try:
        wps = mo.synthetic_missions(2500, 8)
        full = eng.plan(wps, 3.0, 0.01)
        free = eng.plan(wps, 3.0, 0.01, rows=False)
        ref, counts = comm.gather_plan(full, dst=0)
        assert counts == [full.total_rows] and torch.equal(ref.traj, full.traj)
        side = torch.cuda.Stream(device=eng.device)
        known = comm.plan_counts(free)
        assert known == ([2500 * 8], [full.total_rows])
        for plan, parts, stream, kc in ((full, True, None, None), (free, True, side, known), (free, (0.5, 1.0), None, None),
                                        (free, (0.001, 0.002, 1.0), side, known)):
            got, cnt = comm.gather_finish(comm.gather_plan_begin(plan, dst=0, stream=stream, parts=parts, known_counts=kc))
            torch.cuda.synchronize()
            assert cnt == counts
            for k in ("traj", "coeffs", "times", "seg_rows", "row_offsets", "first_yaw"):
                assert torch.equal(getattr(got, k), getattr(ref, k)), (parts, k)
        assert part_bounds(2500, PIPELINE_SHARES) == [0, 156, 625, 1250, 2500]
        one = (C.c_int64 * 1)
        args = lambda first, count: ("uavac_gather_plan_part_dev", comm._h, C.c_void_p(free.coeffs.data_ptr()), None, None,      # noqa: E731
                                     one(20000), one(first), one(count), 0, C.c_void_p(ref.coeffs.data_ptr()), None, None)
        for first, count in ((19999, 2), (-1, 1)):
            with pytest.raises(nat.UavacError) as e:
                eng.ctx.call(*args(first, count))
            assert e.value.code == nat.EINVAL
        with pytest.raises(nat.UavacError):                       # the root's own arrays and its outputs must name the same arrays
            eng.ctx.call("uavac_gather_plan_part_dev", comm._h, C.c_void_p(free.coeffs.data_ptr()), None, None, one(20000), one(0), one(8),
                         0, None, C.c_void_p(ref.times.data_ptr()), None)
        with pytest.raises(ValueError):
            comm.gather_plan(eng.plan_ragged([w[:3] for w in wps[:4]], 3.0, 0.01), parts=True)
    finally:
        comm.close()


@pytest.mark.parametrize("m", [1, 2, 8, 12, 20])
def test_both_elimination_orders_against_the_reference_solve_golden(eng, m):
    """`solve_order` is not a tuning knob: 1 (two-ended, the default at every launch), 0 (one-ended), -1 (opt-in: the faster one
    by (m, B)) round differently (5e-14).  Each of them sits within 1e-9 (relative, SURVEY 8(c) metric) of the reference's
    `np.linalg.solve` coefficients on the golden missions, and within 1e-5 of its lstsq ones; -1 is bitwise one of the two, and
    on a batch past its threshold (m <= 8, 48 missions per SIMD) it is the one-ended kernel."""
    import torch
    from conftest import col_err
    from oracle import minsnap_oracle as mo
    g = load_golden("synthetic_missions.npz")
    wps = g[f"m{m}_wp"]
    got = {}
    try:
        for order in (1, 0, -1):
            eng.ctx.set_option("solve_order", order)
            p = eng.plan(wps, 3.0, 0.01)
            got[order] = p.coeffs.clone()
            co = p.coeffs.cpu().numpy().reshape(-1, 3)
            assert col_err(co, g[f"m{m}_coeffs_solve"].reshape(-1, 3)) < 1e-9, (m, order)
            assert col_err(co, g[f"m{m}_coeffs_lstsq"].reshape(-1, 3)) < 1e-5, (m, order)
        assert torch.equal(got[-1], got[1])                       # 16-32 missions: far below the threshold
        if m == 8:
            big = mo.synthetic_missions(49152, 8)
            full = {}
            for order in (1, 0, -1):
                eng.ctx.set_option("solve_order", order)
                full[order] = eng.plan(big, 3.0, 0.01, rows=False).coeffs
            assert torch.equal(full[-1], full[0]) and not torch.equal(full[0], full[1])
            scale = full[1].abs().max()
            assert float((full[0] - full[1]).abs().max() / scale) < 1e-12
    finally:
        eng.ctx.set_option("solve_order", 1)
    with pytest.raises(Exception):
        eng.ctx.set_option("solve_order", 2)


def test_rows_free_ragged_batches_fly_and_travel_like_full_ones(eng):
    """A ragged batch planned rows-free (plan_ragged(rows=False)): a fleet on it logs what the fleet on the full batch logs, and the
    plan gather (world 1, RCCL) turns it into the full batch's rows on the root -- the rows were never sampled where it was planned."""
    import torch
    from uav_ac import _native as nat
    from uav_ac.fleet import RcclComm
    rng = np.random.default_rng(66)
    missions = _fuzz_missions(rng, 300, vertical=0.2)
    full = eng.plan_ragged(missions, 2.5, 0.01)
    free = eng.plan_ragged(missions, 2.5, 0.01, rows=False)
    a, _ = eng.fleet(full, from_plan=True).rollout(900, state_log=True)
    b, _ = eng.fleet(free).rollout(900, state_log=True)
    c, _ = eng.fleet(full, from_plan=False).rollout(900, state_log=True)
    assert torch.equal(a, b) and torch.equal(a, c)
    with pytest.raises(ValueError):
        free.mission(0)
    buf = C.create_string_buffer(nat.COMM_ID_BYTES)
    eng.ctx.call("uavac_comm_unique_id", buf)
    comm = RcclComm(eng, unique_id=bytes(buf.raw), world=1, rank=0)
    try:
        got, counts = comm.gather_plan(free, dst=0)
        assert counts == [full.total_rows] and torch.equal(got.traj, full.traj) and torch.equal(got.first_yaw, full.first_yaw)
        assert torch.equal(got.row_offsets, full.row_offsets) and torch.equal(got.coeffs, full.coeffs)
    finally:
        comm.close()
