"""The library's own HOST code under AddressSanitizer / UBSan / ThreadSanitizer, on a machine with no GPU (SURVEY.md
section 5: "host tests under -fsanitize=address,undefined").

tests/host_sanitize/Makefile compiles dsp_amd/csrc/capi.hip and host_path.hip -- the tap cache, the per-stream workspace
sets, the staging ring with its copy threads -- plus the C++ veneer with g++ against a host-only stand-in of the HIP
runtime, and the CPU oracle on its own.  hammer.cpp drives them through the C ABI from 15 threads (300 tap sets through
the 256-slot cache, 16-48 MiB batches through the ring slots from 8 threads on 3 "devices", 12 streams through the 8
workspace sets) and checks that every byte comes back where it belongs.  A sanitizer report fails the run (exit code).
"""
import os
import shutil
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_sanitize")


@pytest.fixture(scope="module")
def built():
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("needs g++ and make")
    p = subprocess.run(["make", "-j3", "-C", HERE], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    return HERE


def _run(path, *args, env=None, timeout=600):
    e = dict(os.environ, **(env or {}))
    e.pop("LD_PRELOAD", None)  # a preloaded library in front of the sanitizer runtime makes ASan refuse to start
    p = subprocess.run([path, *args], capture_output=True, text=True, timeout=timeout, env=e)
    # a sanitizer runtime that cannot start on this kernel (address-space layout it does not know) is not a finding
    for msg in ("unexpected memory mapping", "Shadow memory range interleaves", "ReserveShadowMemoryRange failed"):
        if p.returncode != 0 and msg in p.stderr and "host hammer" not in p.stdout:
            pytest.skip(f"sanitizer runtime cannot start here: {msg}")
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-6000:])
    return p.stdout


def test_oracle_under_asan_ubsan(built):
    assert "oracle sweep ok" in _run(os.path.join(built, "oracle_asan"))


def test_host_code_under_asan_ubsan(built):
    out = _run(os.path.join(built, "hammer_asan"), "1", env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0"})
    assert "host hammer ok" in out


def test_host_code_under_tsan(built):
    out = _run(os.path.join(built, "hammer_tsan"), "1", env={"TSAN_OPTIONS": "halt_on_error=1:second_deadlock_stack=1"})
    assert "host hammer ok" in out
