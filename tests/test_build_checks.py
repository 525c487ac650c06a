"""Checks on the machine code that ships (CPU; needs the built library and ROCm's llvm-objdump)."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dsp_amd", "lib", "libdsp_gpu.so")


def _checker():
    spec = importlib.util.spec_from_file_location("check_agpr_prefetch", os.path.join(ROOT, "scripts", "check_agpr_prefetch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def disassembly():
    chk = _checker()
    if not os.path.exists(LIB) or not os.path.exists(chk.OBJDUMP):
        pytest.skip("needs dsp_amd/lib/libdsp_gpu.so and llvm-objdump")
    return chk, chk.disassemble(LIB)


def test_hand_scheduled_prefetch_is_behind_its_wait(disassembly):
    """fir_os_wave_kernel loads into accumulator registers by asm and waits with a hand-counted vmcnt: no instruction of
    the shipped code may name one of those registers while its load can still be pending, on any path; no spills."""
    chk, text = disassembly
    rows = chk.analyse(text)
    assert len(rows) == 8  # I0 = 1 .. 8
    for label, count, stats, errors in rows:
        assert not errors, (label, errors[:5])
        assert len(stats["prefetch_registers"]) == 128, label
        assert not stats["scratch"], label


def test_the_check_sees_a_wait_that_is_one_short(disassembly):
    """The same walk on the same code with every hand-counted wait loosened by one must complain: the last prefetch load
    is then no longer covered."""
    chk, text = disassembly
    loosened, n = re.subn(r"s_waitcnt vmcnt\((\d+)\)", lambda m: m.group(0) if m.group(1) == "0" else
                          f"s_waitcnt vmcnt({int(m.group(1)) + 1})", text)
    assert n > 0
    rows = chk.analyse(loosened)
    assert all(errors for _, _, _, errors in rows), [(label, len(errors)) for label, _, _, errors in rows]
