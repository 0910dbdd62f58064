"""Build-time checks on the generated gfx950 code (no GPU: hipcc cross-compiles).

Kernel A streams the seed vector through the scalar cache with hand double-buffered `s_load_dwordx16` / `s_waitcnt lgkmcnt(0)`
pairs written as SEPARATE inline-asm statements (rattle_amd/csrc/bv_filter.hip: BVF_SLOAD / BVF_SWAIT).  Between the two the
compiler sees the bank as an ordinary defined SGPR value; nothing in the language stops it from copying, spilling or reading
it there, before the scalar load has delivered.  Correctness therefore rests on the code one compiler version emits -- so the
emitted code is what this test checks: no instruction between a bank's s_load and the next full lgkmcnt(0) wait touches a
register of that bank."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


import sys
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_asm


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_bv_filter_scalar_banks_are_not_touched_before_their_wait(tmp_path):
    """(the same check gates the library's build: rattle_amd/csrc/Makefile, bv_filter.o)"""
    check_asm.check_bv_filter(check_asm.device_asm("bv_filter.hip", tmp_path))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_poa_row_loops_keep_spills_and_scratch_out_of_their_hot_blocks(tmp_path):
    """Kernel C is one template of ~100 scalar registers and 160-240 spilled ones per instance (26 instances); the row loops' speed used to
    depend on where the register allocator put the reloads ("the shipped build is the measured one").  The emitted code
    is checked instead: in the hot blocks of every packed row loop (barrier, teams, band) no scratch / flat access and at most
    check_asm.MAX_HOT_RELOADS reloads of spilled scalar registers (today: 0-2)."""
    stats = check_asm.check_poa_row_loops(check_asm.device_asm("poa.hip", tmp_path))
    names = " ".join(stats)
    for inst in ("ILi4ELi8ELi4ELi1E", "ILi6ELi4ELi4ELi1E", "ILi4ELi4ELi4ELi7E", "ILi4ELi2ELi4ELi7E", "ILi4ELi8ELi1ELi8E", "ILi4ELi8ELi4ELi8E"):
        assert inst in names                                     # stage 1's two classes, the team kernels, both band kernels were looked at
    # and the checker itself notices what it is there for
    bad = ["_ZN6rattle10poa_kernelILi4ELi8ELi4ELi1EEEvNS_8poa_argsE:", ".LBB0_1:"] + ["\tv_writelane_b32 v40, s3, %d" % k for k in range(9)] \
        + ["\tv_pk_max_i16 v1, v2, v3"] * 40 + ["\tv_mov_b32_dpp v1, v2 row_shr:1"] * 5 + ["\tv_readlane_b32 s4, v40, 3"] * 6 + ["\tscratch_load_dword v5, off, off"] \
        + ["\ts_cbranch_scc1 .LBB0_1", ".Lfunc_end0:"]
    one = check_asm.poa_row_loop_stats(bad)
    (x, y, hot, reloads, mem), = one["_ZN6rattle10poa_kernelILi4ELi8ELi4ELi1EEEvNS_8poa_argsE"]
    assert reloads == 6 and mem == 1


def test_bench_device_sampler_summarises_what_sysfs_says(monkeypatch):
    """bench.py samples sysfs (clock lines like '2406Mhz', power in watts) on a thread during the warm-up steps; the summary is part of the
    driver's bench line, so the parsing must survive whatever the box says: strings with units, missing keys, nothing at all."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    script = iter([{"sclk": "2406Mhz", "mclk": "2000Mhz", "power_w": 900.0}, {"sclk": "1800Mhz", "power_w": 700.0, "temp_c": None}, None, {"sclk": "N/A"}])
    monkeypatch.setattr(bench, "device_state", lambda light=False: next(script, None))
    s = bench.DeviceSampler(period=0.01).start()
    time.sleep(0.15)
    out = s.summary()
    assert out["samples"] == 3                                   # the None sample is dropped
    assert out["sclk"] == {"mean": 2103.0, "min": 1800.0, "max": 2406.0}
    assert out["power_w"]["max"] == 900.0 and "temp_c" not in out
    monkeypatch.setattr(bench, "device_state", lambda light=False: None)
    assert bench.DeviceSampler(period=0.01).start().summary() == {"samples": 0}
