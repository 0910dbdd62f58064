"""Check of the generated gfx950 code of kernel A (rattle_amd/csrc/bv_filter.hip), run by the library's own build (Makefile: the
object is only accepted with it) and by tests/test_build_checks.py.

Kernel A streams the seed vector through the scalar cache with hand double-buffered `s_load_dwordx16` / `s_waitcnt lgkmcnt(0)`
pairs written as SEPARATE inline-asm statements (BVF_SLOAD / BVF_SWAIT).  Between the two the compiler sees the bank as an
ordinary defined SGPR value; nothing in the language stops it from copying, spilling or reading it there, before the scalar load
has delivered.  Correctness therefore rests on the code one compiler version emits -- so the emitted code is what is checked: no
instruction between a bank's s_load and the next full lgkmcnt(0) wait touches a register of that bank.

usage: check_asm.py [bv_filter.s]      (without an argument: compiles rattle_amd/csrc/bv_filter.hip to assembly first)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def device_asm(src, out_dir):
    out = os.path.join(str(out_dir), os.path.basename(src) + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                           "-o", out, os.path.join(ROOT, "rattle_amd", "csrc", src)], stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def _sgprs(line):
    regs = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", line):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(x) for x in re.findall(r"\bs(\d+)\b", line))
    return regs


def check_bv_filter(asm_lines):
    lines = [l.strip() for l in asm_lines]
    loads = [i for i, l in enumerate(lines) if l.startswith("s_load_dwordx16")]
    assert len(loads) >= 16                       # eight per strand variant of the kernel (+ the cross-seed prefetch): the hand-written loads are there
    labels = {re.match(r"^(\.LBB\d+_\d+):", l).group(1): i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    for i in loads:
        m = re.match(r"s_load_dwordx16 s\[(\d+):(\d+)\]", lines[i])
        bank = set(range(int(m.group(1)), int(m.group(2)) + 1))
        # every path from the load to the next full lgkmcnt(0) wait (the prefetch of the next seed's first eight dwords crosses
        # the loop's back edge): follow fall-through and branch targets
        todo, seen, waits = [i + 1], set(), 0
        while todo:
            j = todo.pop()
            while j < len(lines) and j not in seen:
                seen.add(j)
                l = lines[j]
                if not l or l.startswith((";", ".", "//")) or l.split(";")[0].strip().endswith(":"):
                    j += 1
                    continue
                body = l.split(";")[0].strip()
                if body.startswith("s_waitcnt") and "lgkmcnt(0)" in body:
                    waits += 1
                    break
                assert not (bank & _sgprs(body)), f"line {j + 1}: `{l}` touches s[{min(bank)}:{max(bank)}] before the wait for its load (line {i + 1})"
                assert not body.startswith(("s_setpc", "s_endpgm", "s_swappc")), f"line {j + 1}: the kernel may end before the load of line {i + 1} has landed"
                t = re.match(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", body)
                if t:
                    todo.append(labels[t.group(2)])
                    if t.group(1) == "s_branch":
                        break
                j += 1
        assert waits >= 1 and len(seen) < 400, f"the wait for the load of line {i + 1} is {len(seen)} instructions away"


if __name__ == "__main__":
    if len(sys.argv) > 1:
        check_bv_filter(open(sys.argv[1]).read().splitlines())
    else:
        with tempfile.TemporaryDirectory() as d:
            check_bv_filter(device_asm("bv_filter.hip", d))
    print("bv_filter: scalar banks untouched between load and wait")
