"""The mailbox protocol of kernel C's skewed wavefront pipeline (rattle_amd/csrc/poa.hip: dp_rows_sk), restated on host threads
(tests/stubs/sk_protocol_sim.cpp: one thread per wavefront, sequentially consistent atomics for in-order LDS operations) and
run without a GPU: every value a wavefront takes from its left neighbour's mailbox must be the one written for exactly that
row, no slot may be overwritten while it is still needed, and nobody may wait forever (the first version of the kernel did: its
last wavefront never published the counter its left neighbour's back-pressure reads -- this test hangs on that version).
The kernel itself is compared with the oracle in tests/test_gpu_poa.py / test_gpu_correct.py."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = tmp_path_factory.mktemp("sk") / "sk_protocol_sim"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-o", str(exe), os.path.join(ROOT, "tests", "stubs", "sk_protocol_sim.cpp")])
    return str(exe)


# NW, wavefronts with columns, rows, mailbox slots, ring rows, ring format
@pytest.mark.parametrize("cfg", [(4, 4, 3000, 12, 8, 0), (4, 4, 3000, 12, 8, 1), (4, 4, 3000, 12, 4, 0), (4, 3, 3000, 12, 6, 1), (8, 8, 3000, 12, 8, 1),      # the kernel's own: SK_D = 12
                                 (4, 4, 3000, 16, 8, 0), (4, 4, 3000, 16, 8, 1), (8, 8, 3000, 16, 8, 1), (8, 5, 3000, 16, 8, 0), (4, 2, 2000, 16, 8, 1),
                                 (4, 1, 500, 16, 8, 1), (4, 4, 3000, 10, 8, 0), (4, 4, 3000, 3, 1, 1), (16, 16, 1500, 16, 4, 1), (4, 3, 40, 16, 8, 0)])
def test_mailbox_protocol_neither_races_nor_deadlocks(sim, cfg):
    for seed in (1, 2, 3):
        r = subprocess.run([sim] + [str(x) for x in cfg] + [str(seed)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "SK_PROTOCOL_OK" in r.stdout, (cfg, seed, r.stderr[-500:])
