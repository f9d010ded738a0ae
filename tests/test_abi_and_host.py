"""CPU-side checks: the C-ABI library loads and exports every symbol include/uavac.h declares, the
ctypes mirror of uavac_vehicle matches the header, the engine refuses to run without a GPU (no CPU
fallback), and the product package never imports the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import PKG, REPO


def header_functions():
    text = open(os.path.join(REPO, "include", "uavac.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(uavac_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from uav_ac import _native as nat
    lib = nat.lib()
    declared = header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/uavac.h but not exported"
    assert sorted(nat.exported_symbols()) == declared       # the ctypes table covers the header exactly
    assert lib.uavac_version() == nat.VERSION == 310


def test_vehicle_struct_layout_and_defaults():
    from uav_ac import _native as nat
    text = open(os.path.join(REPO, "include", "uavac.h")).read()
    body = text[text.index("typedef struct uavac_vehicle {"):text.index("} uavac_vehicle;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    n_doubles = 0
    for decl in re.findall(r"double\s+([^;]+);", body):
        for item in decl.split(","):
            mm = re.search(r"\[(\d+)\]", item)
            n_doubles += int(mm.group(1)) if mm else 1
    assert C.sizeof(nat.Vehicle) == 8 * n_doubles + 8
    V = nat.Vehicle.default()                              # pure host function: works without a GPU
    assert (V.g, V.dt, V.mass, V.kappa, V.inner_per_outer) == (9.81, 0.001, 0.5, 0.016, 10)
    assert list(V.inertia) == [0.0023, 0.0023, 0.0046]
    assert V.kp_z == 1 / 0.2 ** 2 and V.kp_r == 1 / 0.09     # quad.py:53-73
    # free flight by default; the reference scene's plane / body box / MuJoCo's default solref when switched on
    assert (V.ground, V.ground_z, V.ground_clearance, V.ground_timeconst) == (0, 0.0, 0.02, 0.02)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from uav_ac import _native as nat
    from uav_ac.fleet import Engine
    with pytest.raises(nat.UavacError):
        nat.Context(0)
    with pytest.raises(nat.UavacError):
        Engine()
    from uav_ac.planning.minimum_snap import MinimumSnap
    with pytest.raises(nat.UavacError):
        MinimumSnap(np.array([[0., 0, 0], [1, 0, 0]]), None, 1.0, 0.01).get_trajectory()


def test_product_never_imports_the_oracle():
    bad = []
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                src = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "liboracle" in src or "oracle/" in src:
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_shard_bounds_partition_the_batch():
    from uav_ac.fleet import balanced_root_share, shard_bounds, shard_sizes
    for B in (1, 7, 65536, 262144):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
            # a smaller block for the gather's root (round-3 VERDICT 2), any root: still a partition in rank order
            for root in {0, world - 1}:
                for share in (0.0, 0.05, 1.0 / world, 0.9):
                    ws = shard_sizes(B, world, share, root)
                    spans = [shard_bounds(B, r, world, share, root) for r in range(world)]
                    assert sum(ws) == B and [hi - lo for lo, hi in spans] == ws and spans[0][0] == 0 and spans[-1][1] == B
                    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
                    if world > 1 and B >= world:
                        peers = [n for r, n in enumerate(ws) if r != root]
                        floor = 0 if share == 0.0 else 1             # an exact 0: the root only assembles the trajectories
                        assert max(peers) - min(peers) <= 1 and min(peers) >= 1 and ws[root] >= floor
                        assert ws[root] == min(max(int(round(share * B)), floor), B - (world - 1))
    # BASELINE configs[3] on 8 GPUs.  Round 6: every rank plans rows-free and the root samples everybody's rows beside its own
    # flight -- it keeps 1-3 % of the missions (an equal block is 12.5 %); the round-5 form (peers sample too, root re-samples):
    # ~6 %
    # ... and, since a flight beside its own sampler runs a quarter slower (measured), NOTHING: 0.0, which `shard_sizes` honours
    s8 = balanced_root_share(262144, 8, 5000, 8)
    assert s8 == 0.0 and shard_sizes(262144, 8, s8, 0) == [0, 37450] + [37449] * 6 and balanced_root_share(262144, 1, 5000, 8) == 1.0
    assert 0.005 < balanced_root_share(262144, 8, 5000, 8, flight_beside_sampler=1.0, hbm_write_bytes_per_s=5.4e12) < 0.04
    assert shard_sizes(10, 4, 0.0, 3) == [4, 3, 3, 0] and shard_sizes(10, 4, 0.01, 0) == [1, 3, 3, 3]
    assert 0.04 < balanced_root_share(262144, 8, 5000, 8, rows_free=False, hbm_write_bytes_per_s=5.8e12) < 0.09
    assert balanced_root_share(262144, 2, 5000, 8) <= 0.5 and balanced_root_share(262144, 4, 5000, 8) <= 0.25
    # the cost table is an argument (bench.py measures it on the GPUs at hand); the cut follows the table; three-column tables
    # (rounds 4 / 5: no rows-free planning column) are still taken
    from uav_ac.fleet import DEFAULT_TICK_TABLE, candidate_shard_sizes
    assert balanced_root_share(262144, 8, 5000, 8, tick_table=DEFAULT_TICK_TABLE) == s8
    flat = [(n, 1.0, 0.018) for n in (4096, 16384, 32768, 37450)]                 # a tick that costs the same at every size:
    slow_big = [(4096, 0.8, 0.018), (16384, 0.8, 0.018), (32768, 1.2, 0.018), (37450, 1.5, 0.018)]    # ... against one that grows with the shard
    assert balanced_root_share(262144, 8, 5000, 8, tick_table=slow_big) > balanced_root_share(262144, 8, 5000, 8, tick_table=flat)
    cheap_plans = [row + (0.002,) for row in slow_big]                            # rows-free planning: the peers finish earlier, the root takes less
    assert balanced_root_share(262144, 8, 5000, 8, tick_table=cheap_plans) < balanced_root_share(262144, 8, 5000, 8, tick_table=slow_big)
    assert candidate_shard_sizes(262144, 8) == [4096, 16384, 32768, 37450] and candidate_shard_sizes(262144, 2) == [16384, 65536, 131072, 262144]
    with pytest.raises(ValueError):
        balanced_root_share(262144, 8, 5000, 8, tick_table=[(0, 1.0, 1.0)])
    with pytest.raises(ValueError):
        shard_sizes(10, 2, 1.5)


def test_pipeline_part_bounds():
    """sharding.part_bounds: the mission boundaries of a pipelined plan gather's parts -- pure arithmetic every rank repeats for
    every rank: monotone, from 0 to n, small blocks degenerate to empty parts, shares are validated."""
    from uav_ac.sharding import PIPELINE_SHARES, part_bounds
    assert PIPELINE_SHARES[-1] == 1.0 and part_bounds(36864) == [0, 2304, 9216, 18432, 36864]
    for n in (0, 1, 2, 5, 15, 16, 17, 8191, 262144):
        b = part_bounds(n)
        assert b[0] == 0 and b[-1] == n and len(b) == len(PIPELINE_SHARES) + 1 and all(x <= y for x, y in zip(b, b[1:]))
    assert part_bounds(10, (0.5, 1.0)) == [0, 5, 10] and part_bounds(7, (1.0,)) == [0, 7]
    for bad in ((), (0.5,), (0.5, 0.5, 1.0), (0.7, 0.2, 1.0), (0.0, 1.0)):
        with pytest.raises(ValueError):
            part_bounds(10, bad)


def _build_c_demo(tmp_path):
    import subprocess
    from conftest import PKG, REPO
    exe = str(tmp_path / "c_abi_demo")
    lib_dir = os.path.join(PKG, "lib")
    subprocess.run(["gcc", os.path.join(REPO, "examples", "c_abi_demo.c"), "-I" + os.path.join(REPO, "include"), "-L" + lib_dir,
                    "-luavac", "-Wl,-rpath," + lib_dir, "-lm", "-Wall", "-Werror", "-o", exe], check=True, capture_output=True)
    return exe


def test_c_program_links_against_the_abi(tmp_path):
    """examples/c_abi_demo.c -- plain C, only include/uavac.h and libuavac.so -- compiles without warnings and links."""
    from uav_ac import _native as nat
    nat.lib()                                          # builds libuavac.so if it is not there yet
    assert os.path.exists(_build_c_demo(tmp_path))


@pytest.mark.gpu
def test_c_program_plans_and_flies_a_mission(tmp_path):
    import subprocess
    # the first process to touch the GPU on a fresh box pages the HIP runtime and RCCL in from a cold image: minutes, not seconds
    out = subprocess.run([_build_c_demo(tmp_path)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "worst tracking error" in out.stdout


def test_rollout_kernels_keep_two_waves_per_simd():
    """Every variant of control_rollout_kernel must fit TWO waves on a SIMD (<= 256 of its 512 VGPRs, no spills): the check
    __graft_entry__.build() runs (uav_ac/_buildcheck.py), here as a test; the library also says which compiler built it."""
    from uav_ac import _buildcheck
    from uav_ac import _native as nat
    nat.lib()                                          # builds the library (and build/*.o) if it is not there yet
    counts = _buildcheck.check_rollout_registers()
    if counts is None:
        pytest.skip("no build directory / LLVM tools: library was built elsewhere")
    assert len(counts) >= 40 and max(v for _, v, _ in counts) <= 256 and all(s == 0 for _, _, s in counts)
    info = nat.lib().uavac_build_info().decode()
    assert info.startswith("libuavac %d; gfx950; HIP " % nat.VERSION) and "clang" in info.lower(), info


def test_planning_kernels_keep_their_register_budgets():
    """The streaming sampler at six waves per SIMD (<= 80 VGPRs; four with jerk / snap), the solve at two (one for the variant that
    keeps five knots in registers), none with spills: the second check of __graft_entry__.build()."""
    from uav_ac import _buildcheck
    from uav_ac import _native as nat
    nat.lib()
    counts = _buildcheck.check_planning_registers()
    if counts is None:
        pytest.skip("no build directory / LLVM tools: library was built elsewhere")
    assert len(counts) >= 40 and all(s == 0 for _, _, s in counts)
    for kern in ("minsnap_solve_bt_kernel", "minsnap_solve_tw_kernel"):                   # one-ended and two-ended solve
        assert any(v > 256 for n, v, _ in counts if kern in n), kern                     # the five-block variant really uses the whole file


def test_row_prefetch_lands_in_the_carried_registers_and_no_diagnostics_ship():
    """The row-fed rollout kernels issue their trajectory-row prefetch from inline asm the compiler cannot see in flight; that is
    right only while the loads land in the very registers row_wait() hands on and nothing touches them in between.  Checked in
    the disassembly of every row-fed variant (the third check of __graft_entry__.build()), together with: the shipped library
    exports no uavac_diag_* symbol, the product kernels carry no UAVAC_DIAG block, and the Makefile turns inline-asm warnings
    (a clobber the compiler will not honour) into errors."""
    from uav_ac import _buildcheck
    if not os.path.exists(os.path.join(PKG, "build", "control_rollout.o")):
        pytest.skip("no object files here (library built elsewhere)")
    assert _buildcheck.check_row_prefetch() == 16
    assert _buildcheck.check_heading_prefetch() >= 20          # the sampler's LDS prefetch of the heading coefficients, same rule
    assert _buildcheck.check_no_diagnostics() is True
    for name in os.listdir(os.path.join(PKG, "csrc")):
        with open(os.path.join(PKG, "csrc", name)) as f:
            assert "UAVAC_DIAG" not in f.read(), name
    with open(os.path.join(PKG, "Makefile")) as f:
        assert "-Werror=inline-asm" in f.read()


def _row_fed_kernel():
    from uav_ac import _buildcheck
    kernels = _buildcheck._disassemble_cfg(os.path.join(PKG, "build", "control_rollout.o"))
    if kernels is None:
        pytest.skip("no object file / LLVM tools here")
    name, ins = next((n, i) for n, i in kernels.items() if "control_rollout_kernel" in n and "ELb0ELb0ELb0ELi0E" in n)
    at = max(j for j, x in enumerate(ins) if _buildcheck._ROW_LOAD.match(x) and "offset:64" in x)
    return kernels, name, ins, at, _buildcheck._ROW_LOAD.match(ins[at]).group(1)


def _check_with(kernels):
    from uav_ac import _buildcheck
    real = _buildcheck._disassemble_cfg
    _buildcheck._disassemble_cfg = lambda obj: kernels
    try:
        return _buildcheck.check_row_prefetch()
    finally:
        _buildcheck._disassemble_cfg = real


def test_row_prefetch_check_catches_a_copy_made_too_early():
    """The disassembly check fails when an instruction reads a row register between the loads and their wait."""
    from uav_ac._buildcheck import Ins
    kernels, name, ins, at, reg = _row_fed_kernel()
    assert _check_with(kernels) == 16                           # the build as it is passes
    tampered = dict(kernels)
    tampered[name] = ins[:at + 1] + [Ins(f"v_mov_b32_e32 v255, v{reg}")] + ins[at + 1:]
    with pytest.raises(RuntimeError, match="touches a row register"):
        _check_with(tampered)


def test_row_prefetch_check_follows_branches_round_a_wait():
    """Round-5 advice: a `s_waitcnt vmcnt(0)` that the executed path BRANCHES OVER proves nothing.  Behind the in-loop row loads:
    a conditional branch over a wait, then an instruction that touches a row register.  In linear layout a wait precedes the
    instruction (the round-5 check accepted that); on the control-flow graph the loads reach it without one."""
    from uav_ac._buildcheck import Ins
    kernels, name, ins, at, reg = _row_fed_kernel()
    landing = 0x7fff0000                                        # an address no real instruction has
    tampered = dict(kernels)
    tampered[name] = (ins[:at + 1] + [Ins("s_cbranch_scc1 1", None, landing), Ins("s_waitcnt vmcnt(0)"),
                                     Ins(f"v_mov_b32_e32 v255, v{reg}", landing)] + ins[at + 1:])
    with pytest.raises(RuntimeError, match="touches a row register"):
        _check_with(tampered)
    # the same instruction behind a wait that every path crosses is fine
    tampered[name] = ins[:at + 1] + [Ins("s_waitcnt vmcnt(0)"), Ins(f"v_mov_b32_e32 v255, v{reg}")] + ins[at + 1:]
    assert _check_with(tampered) == 16
    # ... and control flow the check cannot follow is refused, not waved through
    tampered[name] = ins[:at + 1] + [Ins("s_setpc_b64 s[0:1]")] + ins[at + 1:]
    with pytest.raises(RuntimeError, match="indirect control flow"):
        _check_with(tampered)


def test_a_build_check_that_cannot_run_fails_the_build(monkeypatch):
    """`run_all` -- what __graft_entry__.build() and the autobuild of uav_ac._native.lib() call -- treats a check that cannot run
    (no LLVM tools, no object files) as a failure, and leaves a stamp naming library and compiler when everything passed."""
    from uav_ac import _buildcheck
    from uav_ac import _native as nat
    nat.lib()
    if not os.path.exists(os.path.join(PKG, "build", "control_rollout.o")):
        pytest.skip("no object files here (library built elsewhere)")
    stamp = _buildcheck.run_all()
    assert stamp == _buildcheck.read_stamp()
    assert stamp["library_sha256"] == _buildcheck.library_sha256() and stamp["build_info"] == nat.lib().uavac_build_info().decode()
    assert stamp["checks"]["row_prefetch"] == 16 and "clang" in stamp["checked_with"].lower()
    monkeypatch.setattr(_buildcheck, "LLVM_BIN", "/nonexistent")
    with pytest.raises(RuntimeError, match="could not run"):
        _buildcheck.run_all(write_stamp=False)


def test_autobuild_runs_the_build_checks_and_sets_a_failing_library_aside(tmp_path, monkeypatch):
    """uav_ac._native.lib() with UAVAC_AUTOBUILD=1 and no library: `make`, then the SAME checks as __graft_entry__.build(); a
    build that fails them is not loaded and not left in place for the next process."""
    import shutil
    from uav_ac import _buildcheck
    from uav_ac import _native as nat
    real_lib = nat.lib()
    pkg = tmp_path / "pkg"
    (pkg / "lib").mkdir(parents=True)
    (pkg / "Makefile").write_text(f".PHONY: lib\nlib:\n\tcp {nat.LIB_PATH} {pkg}/lib/libuavac.so\n")
    target = str(pkg / "lib" / "libuavac.so")
    monkeypatch.setattr(nat, "LIB_PATH", target)
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setenv("UAVAC_AUTOBUILD", "1")
    calls = []

    def failing(*a, **k):
        calls.append(1)
        raise RuntimeError("'v_mov_b32_e32 v255, v2' touches a row register while the row loads may be in flight")
    monkeypatch.setattr(_buildcheck, "run_all", failing)
    with pytest.raises(nat.UavacError, match="FAILED its build checks"):
        nat.lib()
    assert calls and not os.path.exists(target) and os.path.exists(target + ".failed-buildcheck")
    # with passing checks the freshly built library is loaded
    monkeypatch.setattr(_buildcheck, "run_all", lambda *a, **k: calls.append(2) or {})
    assert nat.lib() is not None and calls[-1] == 2 and os.path.exists(target)
    monkeypatch.setattr(nat, "_lib", real_lib)
    shutil.rmtree(pkg)
