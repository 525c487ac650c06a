#!/usr/bin/env python3
"""Build-time check of fir_os_wave_kernel's hand-scheduled prefetch (dsp_amd/csrc/fir_kernels.hip).

That kernel loads the next block pair into accumulator registers with asm loads the compiler knows nothing about and
waits for them with a hand-counted `s_waitcnt vmcnt(2 * (16 - I0))`.  The language does not promise that the register
allocator leaves those registers alone between a load and its wait, nor that exactly that many memory operations follow
the last prefetch instalment, so this script checks the machine code that ships:

  * every accumulator register that is the destination of a global load is tracked from the load to the first
    `s_waitcnt vmcnt(k)` that covers it (k <= number of vector memory operations issued after the load), along every
    path of the kernel's control-flow graph (fixpoint, minimum age over paths);
  * any instruction that names such a register while its load may still be pending -- v_accvgpr_read / write / mov, a
    second load into it, a store from it -- is an error;
  * scratch instructions (register spills to memory) in these kernels are an error.

The kernel chooses between `vmcnt(0)` (first pair) and `vmcnt(#stores)` with a wave-uniform flag that the compiler keeps
in a scalar register pair (s_mov_b64 s[..], 0 / -1; s_andn2_b64 vcc, exec, s[..]; s_cbranch_vccz / vccnz): the walk
propagates exactly those constants, so that it follows the paths the flag allows and no others.

    python scripts/check_agpr_prefetch.py [dsp_amd/lib/libdsp_gpu.so]

Exit status 0 = every instance clean.  tests/test_build_checks.py runs it on the built library.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
KERNEL = "fir_os_wave_kernel"
VMEM = ("global_", "buffer_", "flat_", "scratch_", "tbuffer_")
CAP = 64  # vmcnt is a 6-bit counter on gfx9: an older operation has completed by any wait that matters

INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
TARGET = re.compile(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$")
AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
VMCNT = re.compile(r"vmcnt\((\d+)\)")
SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def sregs(text):
    regs = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def step_consts(consts, mnem, ops):
    """Constants the walk knows: ('s', lo, hi) -> 0 | -1 for scalar pairs, 'vcc' -> 'z' | 'nz'.  exec is taken to be non-zero."""
    parts = [p.strip() for p in ops.split(",")] if ops else []
    out = dict(consts)
    if mnem.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_endpgm", "s_barrier")):
        return out
    if any("vcc" in p for p in parts):
        out.pop("vcc", None)
    if parts:
        written = sregs(parts[0])
        for key in [k for k in out if k != "vcc"]:
            if written & set(range(key[1], key[2] + 1)):
                del out[key]
    if mnem == "s_mov_b64" and len(parts) == 2 and parts[1] in ("0", "-1"):
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", parts[0])
        if m:
            out[("s", int(m.group(1)), int(m.group(2)))] = int(parts[1])
    if mnem == "s_andn2_b64" and len(parts) == 3 and parts[0] == "vcc" and parts[1] == "exec":
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", parts[2])
        if m and ("s", int(m.group(1)), int(m.group(2))) in consts:
            out["vcc"] = "nz" if consts[("s", int(m.group(1)), int(m.group(2)))] == 0 else "z"
    return out


def agprs(text):
    regs = []
    for m in AREG.finditer(text):
        if m.group(1) is not None:
            regs.append(int(m.group(1)))
        else:
            regs.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def parse_functions(disassembly):
    """{name: (start_address, [(addr, mnemonic, operands, target_or_None)])} for the kernels of interest."""
    funcs, cur = {}, None
    for line in disassembly.splitlines():
        head = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if head:
            cur = None
            if KERNEL in head.group(2) and "wave2" not in head.group(2):
                cur = funcs.setdefault(head.group(2), (int(head.group(1), 16), []))
            continue
        if cur is None:
            continue
        m = INSN.match(line)
        if not m:
            continue
        mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        target = None
        if mnem.startswith("s_cbranch") or mnem == "s_branch":
            t = TARGET.search(line)
            if t is None:
                raise SystemExit(f"branch without a resolvable target: {line}")
            target = cur[0] + int(t.group(1), 16)
        cur[1].append((addr, mnem, ops, target))
    return funcs


def check_function(name, start, insns):
    """Returns (errors, stats)."""
    index = {a: i for i, (a, _, _, _) in enumerate(insns)}
    leaders = {0}
    for i, (_, mnem, _, target) in enumerate(insns):
        if target is not None:
            if target not in index:
                raise SystemExit(f"{name}: branch to {target:#x} leaves the function")
            leaders.add(index[target])
            if i + 1 < len(insns):
                leaders.add(i + 1)
        elif mnem == "s_endpgm" and i + 1 < len(insns):
            leaders.add(i + 1)
    order = sorted(leaders)
    block_of = {b: (b, (order[k + 1] if k + 1 < len(order) else len(insns))) for k, b in enumerate(order)}

    errors, stats = [], {"prefetch_loads": set(), "waits": set(), "scratch": set(), "prefetch_registers": set()}
    seen = set()  # (address, register): one report per place, whatever the number of constant sets that reach it

    def run_block(b, entry_state, report):
        lo, hi = block_of[b]
        state, consts = dict(entry_state[0]), dict(entry_state[1])
        for i in range(lo, hi):
            addr, mnem, ops, target = insns[i]
            if i == hi - 1:
                branch_consts = consts  # what the closing branch sees
            consts = step_consts(consts, mnem, ops)
            is_vmem = mnem.startswith(VMEM)
            regs = agprs(ops)
            if mnem.startswith("scratch_") and report and addr not in stats["scratch"]:
                stats["scratch"].add(addr)
                errors.append(f"{addr:#x}: {mnem} {ops}: register spill to scratch memory")
            for r in regs:
                if r in state:
                    if report and (addr, r) not in seen:
                        seen.add((addr, r))
                        errors.append(f"{addr:#x}: {mnem} {ops}: a{r} is named while its prefetch may be pending "
                                      f"({state[r]} memory operations behind it)")
            if mnem == "s_waitcnt":
                k = VMCNT.search(ops)
                if k:
                    if report:
                        stats["waits"].add(addr)
                    state = {r: age for r, age in state.items() if age < int(k.group(1))}
            if is_vmem:
                state = {r: age + 1 for r, age in state.items() if age + 1 < CAP}
                if mnem.startswith("global_load") and regs:
                    dest = agprs(ops.split(",")[0])
                    for r in dest:
                        state[r] = 0
                        if report:
                            stats["prefetch_registers"].add(r)
                    if report and dest:
                        stats["prefetch_loads"].add(addr)
        succ = []
        last = insns[hi - 1]
        if last[3] is not None:
            taken, fall = True, last[1] != "s_branch"
            known = branch_consts.get("vcc")
            if known and last[1] in ("s_cbranch_vccz", "s_cbranch_vccnz"):
                taken = (known == "z") == (last[1] == "s_cbranch_vccz")
                fall = not taken
            if taken:
                succ.append(index[last[3]])
            if fall and hi < len(insns):
                succ.append(hi)
        elif last[1] != "s_endpgm" and hi < len(insns):
            succ.append(hi)
        return (state, consts), succ

    # entry states per block, one per distinct set of known constants (the flag's two values must not be merged: the
    # choice between the two waits depends on it), ages merged by minimum within each
    entry = {b: {} for b in order}
    entry[0][frozenset()] = {}
    work = [(0, frozenset())]
    while work:
        b, key = work.pop()
        (out, out_consts), succ = run_block(b, (entry[b][key], dict(key)), report=False)
        okey = frozenset(out_consts.items())
        for s in succ:
            old = entry[s].get(okey)
            if old is None:
                new = dict(out)
            else:
                new = dict(old)
                for r, age in out.items():
                    new[r] = min(age, new.get(r, CAP))
            if new != old:
                if len(entry[s]) > 64:
                    raise SystemExit(f"{name}: too many constant sets at one block")
                entry[s][okey] = new
                work.append((s, okey))
    for b in order:  # one reporting pass with the fixpoint states
        for key, ages in entry[b].items():
            run_block(b, (ages, dict(key)), report=True)
    return errors, stats


def disassemble(lib):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, cwd=tmp, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            path = os.path.join(tmp, f)
            syms = subprocess.run([OBJDUMP, "-t", path], check=True, capture_output=True, text=True).stdout
            if KERNEL in syms:
                return subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], check=True, capture_output=True,
                                      text=True).stdout
    raise SystemExit(f"no gfx950 code object of {lib} holds {KERNEL}")


def analyse(disassembly):
    """[(label, instruction count, stats, errors)] for every instance of the kernel in a disassembly."""
    funcs = parse_functions(disassembly)
    if not funcs:
        raise SystemExit(f"{KERNEL}: no instance found")
    rows = []
    for name, (start, insns) in sorted(funcs.items()):
        errors, stats = check_function(name, start, insns)
        short = re.search(r"ILi(\d+)E", name)
        if not stats["prefetch_loads"]:
            errors.append("no prefetch load found: the check does not see what it is meant to check")
        rows.append((f"{KERNEL}<{short.group(1)}>" if short else name, len(insns), stats, errors))
    return rows


def main(argv):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = argv[1] if len(argv) > 1 else os.path.join(root, "dsp_amd", "lib", "libdsp_gpu.so")
    bad = 0
    for label, count, stats, errors in analyse(disassemble(lib)):
        print(f"{label}: {count} instructions, {len(stats['prefetch_loads'])} prefetch loads into "
              f"{len(stats['prefetch_registers'])} accumulator registers, {len(stats['waits'])} vmcnt waits, "
              f"{len(stats['scratch'])} scratch operations, {len(errors)} errors")
        for e in errors[:20]:
            print("   ", e)
        bad += len(errors)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
