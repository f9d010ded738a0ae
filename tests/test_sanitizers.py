"""Sanitizer runs on the CPU build (round-2 VERDICT item 8): the C oracle's own tests under ASan + UBSan, and the
host side of the C ABI -- argument checking, context creation without a device, the C demo's start-up -- against
lib/libuavac_asan.so (host code instrumented, device code untouched: GPU sanitizers and xnack are not available on this pool)."""
import os
import subprocess
import sys

import pytest

from conftest import PKG, REPO

CLANG = "/opt/rocm/lib/llvm/bin/clang"
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}
REPORTS = ("ERROR: AddressSanitizer", "runtime error:", "ERROR: LeakSanitizer")


def _gpu_present():
    import torch
    return torch.cuda.device_count() > 0


def test_c_oracle_passes_its_tests_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "asan"], check=True, capture_output=True)
    so = os.path.join(REPO, "oracle", "_build", "liboracle_asan.so")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), "gcc's libasan.so not found"
    env = dict(os.environ, LD_PRELOAD=libasan, UAVAC_ORACLE_SO=so, **SAN_ENV)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_oracle_c.py",
                        "tests/test_oracle_rrt.py"], cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert not any(tag in out for tag in REPORTS), out[-3000:]
    assert " passed" in out


@pytest.fixture(scope="module")
def asan_lib():
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang not found")
    subprocess.run(["make", "-C", PKG, "asan", "-j4"], check=True, capture_output=True, timeout=1500)
    lib = os.path.join(PKG, "lib", "libuavac_asan.so")
    assert os.path.exists(lib)
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, check=True).stdout.strip()
    return os.path.dirname(lib), os.path.dirname(rt)


def _build(src, exe, asan_lib):
    lib_dir, rt_dir = asan_lib
    subprocess.run([CLANG, "-fsanitize=address,undefined", "-shared-libsan", "-fno-omit-frame-pointer", "-g", src,
                    "-I" + os.path.join(REPO, "include"), "-L" + lib_dir, "-luavac_asan", "-Wl,-rpath," + lib_dir,
                    "-Wl,-rpath," + rt_dir, "-lm", "-Wall", "-Werror", "-o", exe], check=True, capture_output=True)
    return exe


def test_host_side_of_every_entry_point_under_asan_and_ubsan(tmp_path, asan_lib):
    exe = _build(os.path.join(REPO, "tests", "asan_abi_driver.c"), str(tmp_path / "asan_abi_driver"), asan_lib)
    r = subprocess.run([exe], env=dict(os.environ, **SAN_ENV), capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert not any(tag in out for tag in REPORTS), out[-3000:]
    assert "asan driver:" in r.stdout


def test_c_demo_host_side_under_asan_and_ubsan(tmp_path, asan_lib):
    """examples/c_abi_demo.c built with the sanitizers against the instrumented library.  Without a GPU it must stop at
    uavac_create with the library's message (exit 1) -- cleanly: no sanitizer report on that path."""
    if _gpu_present():
        pytest.skip("a GPU is present: the demo would fly (the -m gpu suite runs it uninstrumented)")
    exe = _build(os.path.join(REPO, "examples", "c_abi_demo.c"), str(tmp_path / "c_abi_demo_asan"), asan_lib)
    r = subprocess.run([exe], env=dict(os.environ, **SAN_ENV), capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 1 and "uavac_create" in out, out[-2000:]
    assert not any(tag in out for tag in REPORTS), out[-3000:]
