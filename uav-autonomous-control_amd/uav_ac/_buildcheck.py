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
    if s is None or k is None:
        return None
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
    if len(s) < 20 or len(k) < 8 or bad:
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


def compiler_version() -> str:
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=60).stdout
        return " / ".join(line.strip() for line in out.splitlines()[:2])
    except Exception:                                        # pragma: no cover
        return "unknown"
