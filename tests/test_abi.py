"""The C-ABI library loads, exports every symbol include/rattle_hip.h declares, and fails
loudly (no CPU fallback) when there is no device.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT
from rattle_amd import _lib


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "rattle_amd", "csrc")])
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rattle_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rattle_hip_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names
    assert lib.rattle_hip_abi_version() == 4


def test_header_is_plain_c():
    src = '#include "rattle_hip.h"\nint main(void){return rattle_hip_abi_version()==0;}\n'
    p = os.path.join("/tmp", "abi_c_check.c")
    open(p, "w").write(src)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), p])


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    h = C.c_void_p()
    rc = lib.rattle_hip_ctx_create(0, C.byref(h))
    assert rc == -1 and not h.value
    assert b"no HIP device" in lib.rattle_hip_last_error()
    with pytest.raises(_lib.RattleError):
        from rattle_amd.api import Context
        Context(0)


def test_phred_symbol_table_matches_libm():
    """Kernel D never evaluates log10: it bisects a table of thresholds built with the host libm
    (post_msa.hip).  Same lookup on the host against `(int)(-10*log10(p)+33)` itself: exact powers
    10^(-q/10) (where the expression sits on an integer), their neighbours by a few ulps, means of
    a few of them (what a column's mean error looks like) and random magnitudes."""
    import ctypes as C

    import numpy as np

    from rattle_amd import _lib
    lib = _lib.load()
    t, m = C.c_int(), C.c_int()
    rng = np.random.default_rng(0)
    base = np.array([10.0 ** (-(c - 33) / 10.0) for c in range(0, 128)])
    cases = list(base)
    for b in base:
        bits = int(np.array([b]).view(np.uint64)[0])
        cases += list(np.array([bits + d for d in range(-6, 7)], np.uint64).view(np.float64))
    for _ in range(20000):
        k = int(rng.integers(1, 6))
        cases.append(float(np.sum(rng.choice(base[33:100], k)) / k))
    cases += list(10.0 ** rng.uniform(-12, 18, 20000))
    bad = []
    for p in cases:
        lib.rattle_hip_debug_phred_symbol(float(p), C.byref(t), C.byref(m))
        if t.value != m.value:
            bad.append((p, t.value, m.value))
    assert not bad, bad[:5]
