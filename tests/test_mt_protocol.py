"""The synchronisation of kernel C's teams of wavefronts (rattle_amd/csrc/poa.hip: dp_rows_mt), restated on host threads
(tests/stubs/mt_protocol_sim.cpp: one thread per wavefront, sequentially consistent atomics for in-order LDS operations) and run
without a GPU: a row must find exactly its predecessor rows in the ring (not read before they are written, not overwritten while
still needed: the `slack` rule), exactly its own row's prefix in the mailbox of the column block to its left, and nobody may wait
forever.  A reader that looks further back than slots - slack is the negative control: the simulation must catch it.
The kernel itself is compared with the oracle in tests/test_gpu_poa.py / test_gpu_correct.py."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = tmp_path_factory.mktemp("mt") / "mt_protocol_sim"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-o", str(exe), os.path.join(ROOT, "tests", "stubs", "mt_protocol_sim.cpp")])
    return str(exe)


# NW, column blocks in use, teams, rows, ring slots, slack, mailbox entries
@pytest.mark.parametrize("cfg", [(4, 4, 4, 4000, 24, 8, 4), (4, 4, 4, 4000, 13, 4, 4), (4, 4, 2, 4000, 10, 2, 4), (4, 4, 2, 4000, 11, 4, 4), (4, 4, 1, 4000, 7, 0, 4),
                                 (4, 3, 4, 3000, 17, 8, 4), (4, 1, 4, 4000, 24, 8, 4), (4, 2, 2, 4000, 8, 2, 4), (4, 4, 4, 3000, 10, 4, 2), (4, 4, 3, 3000, 12, 6, 4),
                                 (4, 4, 4, 37, 24, 8, 4), (4, 4, 4, 3, 24, 8, 4), (4, 4, 2, 3000, 4, 2, 1)])
def test_team_protocol_neither_races_nor_deadlocks(sim, cfg):
    for seed in (1, 2, 3):
        r = subprocess.run([sim] + [str(x) for x in cfg] + [str(seed)], capture_output=True, text=True, timeout=180)
        assert r.returncode == 0 and "MT_PROTOCOL_OK" in r.stdout, (cfg, seed, r.stderr[-500:])


def test_the_simulation_catches_a_reader_beyond_the_slack_rule(sim):
    """Negative control: four teams, eight slots, the writer waits for the rows up to row - 4, but readers look EIGHT rows back
    (the kernel's rule is slots - slack = 4): a row may then overwrite an entry that a row of another team, still in flight, has
    yet to read.  The simulation must see it in at least one of a few runs."""
    caught = 0
    for seed in range(1, 9):
        r = subprocess.run([sim, "4", "4", "4", "6000", "8", "4", "4", str(seed), "8"], capture_output=True, text=True, timeout=180)
        caught += r.returncode == 1 and "ring entry" in r.stderr
    assert caught > 0
