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
