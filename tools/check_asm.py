"""Checks of the generated gfx950 code: kernel A (rattle_amd/csrc/bv_filter.hip), run by the library's own build (Makefile: the
object is only accepted with it) and by tests/test_build_checks.py; and the row loops of kernel C (poa.hip; `--poa`), run by
tests/test_build_checks.py and reported by __graft_entry__.build().

Kernel A streams the seed vector through the scalar cache with hand double-buffered `s_load_dwordx16` / `s_waitcnt lgkmcnt(0)`
pairs written as SEPARATE inline-asm statements (BVF_SLOAD / BVF_SWAIT).  Between the two the compiler sees the bank as an
ordinary defined SGPR value; nothing in the language stops it from copying, spilling or reading it there, before the scalar load
has delivered.  Correctness therefore rests on the code one compiler version emits -- so the emitted code is what is checked: no
instruction between a bank's s_load and the next full lgkmcnt(0) wait touches a register of that bank.

usage: check_asm.py [bv_filter.s] | --poa [poa.s]      (without a file: compiles the source to assembly first)"""
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


# ---- kernel C (rattle_amd/csrc/poa.hip): the row loops -----------------------------------------------------------------------
# poa_kernel<...> is one template with 26 instances of ~100 scalar registers and 160-240 spilled ones each; whether a spill reload
# (v_readlane from a spill register, a scratch load) lands INSIDE the hot blocks of a row loop used to depend on unrelated edits
# ("the shipped build is the measured one").  What is checked on the code THIS compiler emits, per instance:
#   * its row loops are found (loops with packed 16-bit arithmetic and DPP scans in them), and in their HOT blocks -- the basic
#     blocks that hold packed arithmetic and no load from HBM, i.e. the recurrence itself, not the far-predecessor / retry paths --
#   * there is no scratch_ / flat_ / buffer_ access (a workspace pointer from scratch turns every access through it into a flat
#     one, which counts on lgkmcnt and vmcnt: round 6's 665-cycle band rows), and
#   * at most MAX_HOT_RELOADS reloads of spilled scalar registers per loop.
POA_PK_ROW_LOOPS = {1: "barrier (dp_rows_v3)", 7: "teams (dp_rows_mt)", 8: "band (dp_rows_band)"}
MAX_HOT_RELOADS = 4


def poa_row_loop_stats(asm_lines):
    """{kernel name: [(first line, last line, hot instructions, spill reloads in hot blocks, memory ops through scratch/flat in hot blocks)]}
    for the instances whose row loop is packed 16-bit (PK 1, 7, 8)."""
    L = asm_lines
    heads = [(i, re.match(r"^(_ZN6rattle10poa_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEEvNS_8poa_argsE):", l)) for i, l in enumerate(L)]
    heads = [(i, m) for i, m in heads if m]
    ends = [i for i, l in enumerate(L) if l.startswith(".Lfunc_end")]
    out = {}
    for a, m in heads:
        if int(m.group(5)) not in POA_PK_ROW_LOOPS:
            continue
        b = min(e for e in ends if e > a)
        K = [l.split(";")[0].rstrip() for l in L[a:b]]
        # registers that hold spilled SGPRs: written lane by lane (v_writelane from an SGPR) at eight or more lanes
        lanes = {}
        for l in K:
            w = re.match(r"\s*v_writelane_b32 (v\d+), s\d+, (\d+)", l)
            if w:
                lanes.setdefault(w.group(1), set()).add(w.group(2))
        spill_regs = {v for v, ls in lanes.items() if len(ls) >= 8}
        labels = {mm.group(1): i for i, l in enumerate(K) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        loops = []
        for i, l in enumerate(K):
            t = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if t and t.group(1) in labels and labels[t.group(1)] < i:
                loops.append((labels[t.group(1)], i))
        rows = []
        for x, y in sorted(loops, key=lambda q: q[1] - q[0]):
            body = K[x:y + 1]
            if sum("v_pk_" in z for z in body) < 30 or sum("dpp" in z for z in body) < 4:
                continue
            if any(x <= y2 and x2 <= y for x2, y2, *_ in rows):
                continue                              # (the unrolled pair of rows and the loops around it: the innermost ones count)
            hot = reloads = mem = 0
            blk = []
            for z in body + [".LBB_end:"]:
                if re.match(r"^\.LBB\w+:", z):
                    if sum("v_pk_" in q for q in blk) >= 4 and not any("global_load" in q for q in blk):      # (a block that loads from HBM is a far-predecessor path)
                        hot += len(blk)
                        reloads += sum(1 for q in blk for r in [re.match(r"\s*v_readlane_b32 s\d+, (v\d+),", q)] if r and r.group(1) in spill_regs)
                        mem += sum(q.strip().startswith(("scratch_", "flat_", "buffer_")) for q in blk)
                    blk = []
                elif re.match(r"\s+(v_|s_|ds_|global_|scratch_|flat_|buffer_)", z):
                    blk.append(z)
            rows.append((x, y, hot, reloads, mem))
        out[m.group(1)] = rows
    return out


def check_poa_row_loops(asm_lines, verbose=False):
    stats = poa_row_loop_stats(asm_lines)
    assert len(stats) >= 18, f"only {len(stats)} instances of poa_kernel with packed rows found"
    for name, rows in stats.items():
        assert rows, f"{name}: no row loop found"
        for x, y, hot, reloads, mem in rows:
            if verbose:
                print(f"{name}: row loop of {y - x} lines, {hot} instructions in its hot blocks, {reloads} spill reloads, {mem} scratch / flat accesses there")
            assert mem == 0, f"{name}: {mem} scratch / flat / buffer accesses in the hot blocks of a row loop"
            assert reloads <= MAX_HOT_RELOADS, f"{name}: {reloads} reloads of spilled scalar registers in the hot blocks of a row loop"
    return stats


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--poa":
        if len(sys.argv) > 2:
            check_poa_row_loops(open(sys.argv[2]).read().splitlines(), verbose=True)
        else:
            with tempfile.TemporaryDirectory() as d:
                check_poa_row_loops(device_asm("poa.hip", d), verbose=True)
        print("poa: no scratch / flat access and at most %d spill reloads in the hot blocks of every row loop" % MAX_HOT_RELOADS)
    else:
        if len(sys.argv) > 1:
            check_bv_filter(open(sys.argv[1]).read().splitlines())
        else:
            with tempfile.TemporaryDirectory() as d:
                check_bv_filter(device_asm("bv_filter.hip", d))
        print("bv_filter: scalar banks untouched between load and wait")
