"""Build-time check of the rollout kernels' register budget (used by __graft_entry__.build() and by the CPU tests).

Every variant of `control_rollout_kernel` must fit TWO wavefronts on a SIMD: at most 256 of its 512 vector registers, no
vector-register spills.  A logged launch puts a compute and a store wave of the same kernel on every SIMD; at 258 registers
(a 6-register 'optimisation' tried in round 3) the bench launch took 1.96 ms instead of 1.27 ms -- a cliff that no functional
test sees and that a toolchain bump can cross silently.  Read from the code object inside build/control_rollout.o."""
import os
import re
import shutil
import subprocess
import tempfile

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
VGPR_LIMIT = 256


def kernel_register_counts(obj: str, kernel: str):
    """[(kernel name, vgpr_count, vgpr_spill_count)] of every kernel of the object file whose name contains `kernel`, or None
    when the object file or the LLVM tools are not there (a library that was built elsewhere)."""
    objdump, readelf = os.path.join(LLVM_BIN, "llvm-objdump"), os.path.join(LLVM_BIN, "llvm-readelf")
    if not (os.path.exists(obj) and os.path.exists(objdump) and os.path.exists(readelf)):
        return None
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(obj, os.path.join(tmp, "o.o"))
        subprocess.run([objdump, "--offloading", "o.o"], cwd=tmp, check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
        notes = subprocess.run([readelf, "--notes", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp)
    found = re.findall(r"\.name:\s+(\S*" + re.escape(kernel) + r"\S*).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", notes, flags=re.S)
    return [(n, int(v), int(s)) for n, v, s in found]


def rollout_register_counts(obj: str = None):
    """The same for every control_rollout_kernel variant."""
    return kernel_register_counts(obj or os.path.join(PKG, "build", "control_rollout.o"), "control_rollout_kernel")


def check_planning_registers():
    """The planning kernels' budgets: the chunk-streaming sampler without jerk / snap runs six waves per SIMD (<= 80 vector
    registers; with jerk / snap four: <= 128), the solve one or two (the variant that keeps five knots in registers: <= 512, the
    others <= 256); none may spill (a spill in the sampler showed as 3 % more HBM writes and no other symptom, NOTES R4-2).
    Returns the counts, None when they cannot be read."""
    s = kernel_register_counts(os.path.join(PKG, "build", "minsnap_sample_stream.o"), "minsnap_sample_stream_kernel")
    k = kernel_register_counts(os.path.join(PKG, "build", "minsnap_solve_bt.o"), "minsnap_solve_bt_kernel")
    k2 = kernel_register_counts(os.path.join(PKG, "build", "minsnap_solve_tw.o"), "minsnap_solve_tw_kernel")
    if s is None or k is None or k2 is None:
        return None
    k = k + k2                                        # (the two-ended solve: same budgets, same naming of the last template argument)
    bad = []
    for n, v, sp in s:
        args = re.search(r"minsnap_sample_stream_kernelILi(\d+)ELb([01])ELb([01])ELb([01])E", n)
        waves, derivs = int(args.group(1)), args.group(3) == "1"
        limit = 128 if (derivs or waves == 16) else 80
        if v > limit or sp:
            bad.append((n[:70], v, sp, limit))
    for n, v, sp in k:
        limit = 512 if n.rstrip("E").endswith("Li5") or "Li5EE" in n else 256
        if v > limit or sp:
            bad.append((n[:70], v, sp, limit))
    if len(s) < 20 or len(k) < 20 or bad:
        raise RuntimeError(f"planning kernels outside their register budgets (name, VGPRs, spills, limit): {bad}; {len(s)} sampler and "
                           f"{len(k)} solve variants found (compiler: {compiler_version()})")
    return s + k


def check_rollout_registers(obj: str = None):
    """Raise RuntimeError when a rollout variant needs more than 256 vector registers or spills any; returns the counts."""
    counts = rollout_register_counts(obj)
    if counts is None:
        return None
    if len(counts) < 40:
        raise RuntimeError(f"only {len(counts)} control_rollout_kernel variants found in the code object")
    over = [(n[:80], v) for n, v, _ in counts if v > VGPR_LIMIT]
    spills = [(n[:80], s) for n, _, s in counts if s > 0]
    if over or spills:
        raise RuntimeError(f"rollout kernels outside the two-waves-per-SIMD budget: > {VGPR_LIMIT} VGPRs {over}; spills {spills} "
                           f"(compiler: {compiler_version()})")
    return counts


class Ins(str):
    """One instruction of a disassembled kernel: its text (the str itself), its address and, for a branch, the address it may go to."""
    addr = None
    target = None

    def __new__(cls, text, addr=None, target=None):
        self = super().__new__(cls, text)
        self.addr, self.target = addr, target
        return self


_BRANCH = ("s_cbranch", "s_branch")
_INDIRECT = ("s_setpc", "s_swappc", "s_call", "s_rfe", "s_cbranch_g_fork", "s_cbranch_i_fork", "s_cbranch_join")


def _disassemble_cfg(obj: str):
    """{kernel name: [Ins]} of the gfx950 code object inside `obj` -- instruction text with its address and branch target, which is
    what a check needs to follow the control flow -- or None when the object file or the LLVM tools are missing."""
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    if not (os.path.exists(obj) and os.path.exists(objdump)):
        return None
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(obj, os.path.join(tmp, "o.o"))
        subprocess.run([objdump, "--offloading", "o.o"], cwd=tmp, check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
        text = subprocess.run([objdump, "-d", "--no-show-raw-insn", os.path.join(tmp, co)], check=True, capture_output=True,
                              text=True).stdout
    finally:
        shutil.rmtree(tmp)
    kernels, cur, base = {}, None, 0
    for line in text.splitlines():
        head = re.match(r"([0-9a-f]+) <([^>]+)>:", line)
        if head:
            base = int(head.group(1), 16)
            cur = kernels.setdefault(head.group(2), [])
        elif cur is not None:
            code, _, note = line.partition("//")
            ins = code.strip()
            if not ins or ins.endswith(":"):
                continue
            at = re.match(r"\s*([0-9A-Fa-f]+):", note)
            to = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$", note) if ins.startswith(_BRANCH) else None
            if ins.startswith(_BRANCH) and to is None and re.search(r"<[^>+]+>\s*$", note):
                to_addr = base                                            # a branch to the kernel's first instruction
            else:
                to_addr = base + int(to.group(1), 16) if to else None
            cur.append(Ins(ins, int(at.group(1), 16) if at else None, to_addr))
    return kernels


def _disassemble(obj: str):
    """{kernel name: [instruction text]}: `_disassemble_cfg` for the checks that read straight-line code."""
    return _disassemble_cfg(obj)


def _successors(ins):
    """Successor indices of every instruction of a kernel (fall-through and branch targets); raises on a branch the disassembly
    gives no target for."""
    index = {x.addr: i for i, x in enumerate(ins) if getattr(x, "addr", None) is not None}
    succ = []
    for i, x in enumerate(ins):
        if x.startswith("s_endpgm"):
            succ.append(())
        elif x.startswith(_INDIRECT):
            raise RuntimeError(f"indirect control flow ('{x}'): the check cannot follow it")
        elif x.startswith(_BRANCH):
            t = index.get(getattr(x, "target", None))
            if t is None:
                raise RuntimeError(f"branch '{x}' without a known target")
            succ.append((t,) if x.startswith("s_branch") else (i + 1, t))
        else:
            succ.append((i + 1,))
    return [tuple(j for j in s if j < len(ins)) for s in succ]


def _vregs(text: str):
    regs = set()
    for lo, hi, one in re.findall(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        regs.update(range(int(lo), int(hi) + 1) if lo else (int(one),))
    return regs


_ROW_LOAD = re.compile(r"global_load_dwordx4 v\[(\d+):(\d+)\], (v\[\d+:\d+\]), off(?: offset:(\d+))?$")


def check_row_prefetch(obj: str = None):
    """The row-fed rollout kernels prefetch a trajectory row with loads the compiler does not know to be in flight (inline asm,
    control_rollout.hip row_issue / row_wait).  That is only right while the loads land in the very registers row_wait() hands
    on: for every row-fed variant, (1) every group of five row loads writes the same twenty registers, (2) no other instruction
    reads or writes one of them at a point that a row load can reach without passing an `s_waitcnt vmcnt(0)` -- decided on the
    kernel's CONTROL-FLOW GRAPH (branch targets from the disassembly; a forward may-analysis "row loads possibly in flight"), not
    on the linear layout: a wait that sits in a block the executed path branches over proves nothing (round-5 advice).  Code
    that no row load reaches (ahead of the first one) is free to use the registers.  Raises RuntimeError otherwise; returns the
    number of variants checked, None when the object file cannot be read."""
    kernels = _disassemble_cfg(obj or os.path.join(PKG, "build", "control_rollout.o"))
    if kernels is None:
        return None
    checked, bad = 0, []
    for name, ins in kernels.items():
        if "control_rollout_kernel" not in name:
            continue
        groups, i = [], 0
        while i + 4 < len(ins):
            ms = [_ROW_LOAD.match(x) for x in ins[i:i + 5]]
            if all(ms) and [int(m.group(4) or 0) for m in ms] == [0, 16, 32, 48, 64] and len({m.group(3) for m in ms}) == 1:
                groups.append((i, frozenset(r for m in ms for r in range(int(m.group(1)), int(m.group(2)) + 1))))
                i += 5
            else:
                i += 1
        poly = re.search(r"control_rollout_kernelILi\d+ELi\d+ELb[01]ELb[01]ELb[01]ELb([01])E", name).group(1) == "1"
        if poly:                                   # (its coefficient loads look alike; the compiler issues and waits for those itself)
            continue
        checked += 1
        if len(groups) != 2 or len({g for _, g in groups}) != 1 or len(groups[0][1]) != 20:
            bad.append((name[:90], f"row-load groups: {[(i, sorted(g)[:1], len(g)) for i, g in groups]}"))
            continue
        dest, inside = groups[0][1], {j for i, _ in groups for j in range(i, i + 5)}
        try:
            succ = _successors(ins)
        except RuntimeError as exc:
            bad.append((name[:90], str(exc)))
            continue
        # in_flight[j]: some path reaches instruction j with a row load issued and no vmcnt(0) wait since
        in_flight = [False] * (len(ins) + 1)
        work = []
        for j in inside:
            for k in succ[j]:
                if not in_flight[k]:
                    in_flight[k] = True
                    work.append(k)
        while work:
            j = work.pop()
            if j >= len(ins) or j in inside or re.match(r"s_waitcnt vmcnt\(0\)", ins[j]):
                continue                            # (a row load's own successors are seeded above; a wait ends the flight)
            for k in succ[j]:
                if not in_flight[k]:
                    in_flight[k] = True
                    work.append(k)
        for j, x in enumerate(ins):
            if j not in inside and in_flight[j] and (_vregs(x) & dest):
                bad.append((name[:90], f"'{x}' touches a row register while the row loads may be in flight"))
                break
    if bad or checked < 16:
        raise RuntimeError(f"row prefetch of the row-fed rollout kernels: {checked} variants checked; {bad} (compiler: {compiler_version()})")
    return checked


_LDS_LOAD = re.compile(r"ds_read_b128 v\[(\d+):(\d+)\], (v\d+)(?: offset:(\d+))?$")


def check_heading_prefetch(obj: str = None):
    """The streaming sampler fetches the heading polynomial's twenty coefficients from LDS with ten `ds_read_b128` the compiler does not
    know to be in flight (minsnap_yaw.h, HeadingFromLds::begin / ready) while the division runs.  Same question as for the rollout's row
    prefetch, same answer: in every sampler variant, between each group of ten such loads and the next `s_waitcnt ... lgkmcnt(0)` no
    instruction touches a destination register and no branch intervenes.  Raises RuntimeError otherwise; returns the number of groups
    checked, None when the object file cannot be read."""
    kernels = _disassemble(obj or os.path.join(PKG, "build", "minsnap_sample_stream.o"))
    if kernels is None:
        return None
    groups, bad = 0, []
    for name, ins in kernels.items():
        if "minsnap_sample_stream_kernel" not in name:
            continue
        i = 0
        while i + 9 < len(ins):
            ms = [_LDS_LOAD.match(x) for x in ins[i:i + 10]]
            if not (all(ms) and [int(m.group(4) or 0) for m in ms] == list(range(0, 160, 16)) and len({m.group(3) for m in ms}) == 1):
                i += 1
                continue
            dest = {r for m in ms for r in range(int(m.group(1)), int(m.group(2)) + 1)}
            groups += 1
            j = i + 10
            while j < len(ins) and not re.match(r"s_waitcnt .*lgkmcnt\(0\)", ins[j]):
                if _vregs(ins[j]) & dest or ins[j].startswith(("s_cbranch", "s_branch", "s_endpgm")):
                    bad.append((name[:80], f"'{ins[j]}' between the coefficient loads and their wait"))
                    break
                j += 1
            i += 10
    if bad or groups < 20:
        raise RuntimeError(f"heading prefetch of the streaming sampler: {groups} load groups checked; {bad} (compiler: {compiler_version()})")
    return groups


def check_no_diagnostics(lib_path: str = None):
    """The shipped library exports no diagnostic entry point (`uavac_diag_*`): those exist only in tools/diag builds."""
    lib_path = lib_path or os.path.join(PKG, "lib", "libuavac.so")
    readelf = os.path.join(LLVM_BIN, "llvm-readelf")
    if not (os.path.exists(lib_path) and os.path.exists(readelf)):
        return None
    syms = subprocess.run([readelf, "--dyn-syms", "-W", lib_path], check=True, capture_output=True, text=True).stdout
    diag = [line.split()[-1] for line in syms.splitlines() if "diag" in line.lower() and " UND " not in line]
    if diag:
        raise RuntimeError(f"{lib_path} exports diagnostic symbols {diag}: it was built with a UAVAC_DIAG_* define")
    return True


def library_sha256(lib_path: str = None) -> str:
    import hashlib
    with open(lib_path or os.path.join(PKG, "lib", "libuavac.so"), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


STAMP = os.path.join(PKG, "lib", "libuavac.buildcheck.json")


def run_all(verbose: bool = False, write_stamp: bool = True) -> dict:
    """EVERY build check, for every place that builds the library (`__graft_entry__.build()`, the autobuild of
    `uav_ac._native.lib()`, `make check`): the rollout kernels' register budget, the row prefetch and the sampler's heading
    prefetch in the disassembly, no diagnostic symbol, the planning kernels' budgets.  A check that cannot run (no object files,
    no LLVM tools) is a FAILURE here: the output-only asm operands of row_issue / HeadingFromLds are only as safe as these
    checks, so a build they did not see must not pass for one they did.  Raises RuntimeError; on success writes
    lib/libuavac.buildcheck.json -- the library's sha256, its uavac_build_info() and the compiler the checks saw -- which
    travels with the library (tests/test_gpu_round6.py compares it with the library the GPU process loaded)."""
    import ctypes
    import json
    result = {}
    steps = (("rollout_registers", check_rollout_registers, lambda r: f"{len(r)} rollout kernels, at most {max(v for _, v, _ in r)} VGPRs, no spills"),
             ("row_prefetch", check_row_prefetch, lambda r: f"row prefetch of {r} row-fed rollout kernels verified on the control-flow graph of the disassembly"),
             ("heading_prefetch", check_heading_prefetch, lambda r: f"{r} coefficient prefetches of the streaming sampler verified in the disassembly"),
             ("no_diagnostics", check_no_diagnostics, lambda r: "no diagnostic symbol exported"),
             ("planning_registers", check_planning_registers, lambda r: f"{len(r)} sampler / solve kernels inside their register budgets, no spills"))
    for key, fn, say in steps:
        r = fn()
        if r is None:
            raise RuntimeError(f"build check '{key}' could not run (object files under {PKG}/build or the LLVM tools under {LLVM_BIN} are "
                               "missing): the library would ship unchecked")
        result[key] = len(r) if isinstance(r, list) else r
        if verbose:
            print(f"build: {say(r)}")
    lib_path = os.path.join(PKG, "lib", "libuavac.so")
    fn = ctypes.CDLL(lib_path).uavac_build_info
    fn.restype = ctypes.c_char_p
    stamp = {"library_sha256": library_sha256(lib_path), "build_info": fn().decode(), "checked_with": compiler_version(), "checks": result}
    if write_stamp:
        tmp = STAMP + f".tmp.{os.getpid()}"
        with open(tmp, "w") as fh:
            json.dump(stamp, fh, indent=1)
        os.replace(tmp, STAMP)
    return stamp


def read_stamp():
    """The record `run_all` left beside the library, or None."""
    import json
    if not os.path.exists(STAMP):
        return None
    with open(STAMP) as fh:
        return json.load(fh)


def compiler_version() -> str:
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout
        return " / ".join(line.strip() for line in out.splitlines()[:2])
    except Exception:                                        # pragma: no cover
        return "unknown"


if __name__ == "__main__":                                   # `make` (the Makefile's default target) runs this after linking
    import sys
    try:
        _s = run_all(verbose=True)
    except Exception as exc:
        print(f"build checks FAILED: {exc}", file=sys.stderr)
        sys.exit(1)
    print(f"build: checks passed for library {_s['library_sha256'][:16]}")
