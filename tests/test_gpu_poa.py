"""Parity of the HIP POA kernel (kernel C) with the oracle's spoa restatement: MSA rows must be
byte-identical (same width, same column for every base)."""
import numpy as np
import pytest

from rattle_amd import synth

pytestmark = pytest.mark.gpu


def _packs_from_synth(n, genes, seed, both=False, max_pack=40):
    seqs, _, tid, _ = synth.reads(n, genes, 1, both, seed=seed)
    packs = []
    for g in range(genes):
        mem = [seqs[i] for i in range(n) if tid[i] == g]
        mem.sort(key=lambda s: -len(s))
        if len(mem) >= 2:
            packs.append(mem[:max_pack])
    return packs


def test_tiny_hand_cases(gpu_ctx, oracle):
    packs = [
        [b"ACGTACGTAC"],                                               # single sequence
        [b"ACGTACGTACGTTTGACA", b"ACGTACGTACGTTTGACA"],                # identical
        [b"ACGTACGTACGTTTGACA", b"ACGTACCTACGTTGACA", b"TTTTTTTT"],    # mismatch, deletion, unalignable
        [b"AAAAAAAAAACCCCCCCCCC", b"GGGGGGGGGG", b"AAAAAAAAAAGGGGGGGGGGCCCCCCCCCC"],
        [b"ACGT" * 30, b"ACGT" * 28 + b"AC", b"CGT" + b"ACGT" * 29, b"ACGA" * 30],
    ]
    rows, width, counters = gpu_ctx.poa_msa(packs)
    for p, pack in enumerate(packs):
        want, _ = oracle.poa_msa(pack)
        assert rows[p] == want, p
        assert width[p] == len(want[0])


def test_synthetic_packs_match_oracle(gpu_ctx, oracle):
    packs = _packs_from_synth(400, 10, seed=5)
    assert len(packs) >= 8
    rows, width, counters = gpu_ctx.poa_msa(packs)
    cells = 0
    for p, pack in enumerate(packs):
        want, c = oracle.poa_msa(pack)
        cells += c
        assert width[p] == len(want[0]), p
        assert rows[p] == want, p
    assert int(counters[0]) == cells                      # exact DP cell count


def test_toyset_clusters_match_oracle(gpu_ctx, oracle, toyset, toyset_clusters):
    """Real reads (long, noisy, up to 4.5 kb): clusters of 6..40 reads from the toyset."""
    cids = [c for c, (m, mem) in enumerate(toyset_clusters) if 6 <= len(mem) <= 40][:12]
    packs = [[toyset[s[0]][1] for s in toyset_clusters[c][1]] for c in cids]
    rows, width, _ = gpu_ctx.poa_msa(packs)
    for p, pack in enumerate(packs):
        want, _ = oracle.poa_msa(pack)
        assert rows[p] == want, cids[p]


def test_multi_segment_rows(gpu_ctx, oracle):
    """Sequences longer than one 1024-column segment and not a multiple of 16."""
    rng = np.random.default_rng(2)
    base = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 2500)]
    pack = []
    for i in range(6):
        s = base.copy()
        idx = rng.integers(0, len(s), 120)
        s[idx] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 120)]
        cut = int(rng.integers(0, 200))
        pack.append(s[cut:len(s) - int(rng.integers(0, 37))].tobytes())
    rows, width, _ = gpu_ctx.poa_msa([pack])
    want, _ = oracle.poa_msa(pack)
    assert rows[0] == want


@pytest.mark.parametrize("length,depth", [(700, 40), (1300, 30), (1800, 25), (2300, 20), (2900, 16), (4500, 10), (7000, 6)])
def test_length_classes_match_oracle(gpu_ctx, oracle, length, depth):
    """One pack per column class of kernel C (1024 / 1536 / 2048 / 2560 packed int16 rows, 4096 / 6144 / 8192 32-bit rows):
    noisy copies (sub / ins / del) of one random transcript, deep enough for bubbles, ties and far predecessors."""
    rng = np.random.default_rng(length)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    tx = acgt[rng.integers(0, 4, length)]
    pack = []
    for _ in range(depth):
        r = rng.random(len(tx))
        s = tx.copy()
        sub = (r >= 0.03) & (r < 0.06)
        s[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        s = s[r >= 0.03]
        pos = np.sort(rng.integers(0, len(s) + 1, int(0.03 * len(s))))
        s = np.insert(s, pos, acgt[rng.integers(0, 4, len(pos))])
        pack.append(s[int(rng.integers(0, 30)):].tobytes())
    pack.sort(key=lambda x: -len(x))
    rows, width, counters = gpu_ctx.poa_msa([pack, pack[::-1]])
    for got, p in zip(rows, (pack, pack[::-1])):
        want, _ = oracle.poa_msa(p)
        assert got == want


@pytest.mark.parametrize("mode", ["dense", "mt4", "mt2", "mt1"])
@pytest.mark.parametrize("length,depth", [(700, 40), (1300, 30), (1800, 25), (2300, 20)])
def test_packed_classes_in_every_form_of_the_row_loop(gpu_ctx, oracle, monkeypatch, length, depth, mode):
    """The four packed column classes under every form of the row loop (RATTLE_POA_MODE): barrier + record-word ring, and the teams of
    wavefronts with the ready-made ring (4 / 2 / 1 teams: rows that do not depend on each other run at the same time, poa.hip
    dp_rows_mt; one team = the lean pipeline without a row barrier) -- all byte-identical to the oracle."""
    monkeypatch.setenv("RATTLE_POA_MODE", mode)
    rng = np.random.default_rng(length + 1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    tx = acgt[rng.integers(0, 4, length)]
    pack = []
    for _ in range(depth):
        r = rng.random(len(tx))
        s = tx.copy()
        sub = (r >= 0.03) & (r < 0.07)
        s[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        s = s[r >= 0.03]
        pos = np.sort(rng.integers(0, len(s) + 1, int(0.03 * len(s))))
        s = np.insert(s, pos, acgt[rng.integers(0, 4, len(pos))])
        pack.append(s[int(rng.integers(0, 30)):].tobytes())
    pack.sort(key=lambda x: -len(x))
    rows, width, counters = gpu_ctx.poa_msa([pack, pack[::-1], pack[: depth // 2]])
    for got, p in zip(rows, (pack, pack[::-1], pack[: depth // 2])):
        want, _ = oracle.poa_msa(p)
        assert got == want


@pytest.mark.parametrize("env", [{"RATTLE_POA_DEBUG": "1"}, {"RATTLE_POA_DEBUG": "2"}, {"RATTLE_POA_DEBUG": "3"}, {"RATTLE_POA_DEBUG": "4"},
                                 {"RATTLE_POA_DEBUG": "4", "RATTLE_POA_MODE": "dense"},
                                 {"RATTLE_POA_NODE_CAP": "700"}, {"RATTLE_POA_MODE": "dense"},
                                 {"RATTLE_POA_MODE": "mt4"}, {"RATTLE_POA_MODE": "mt2"}, {"RATTLE_POA_MODE": "mt1"},
                                 {"RATTLE_POA_MODE": "mt4", "RATTLE_POA_DEBUG": "3"}, {"RATTLE_POA_MODE": "mt2", "RATTLE_POA_NODE_CAP": "700"},
                                 {"RATTLE_POA_MODE": "mt4", "RATTLE_POA_MT_SLOTS": "11"}, {"RATTLE_POA_MODE": "mt2", "RATTLE_POA_MT_SLOTS": "6"},
                                 {"RATTLE_POA_MODE": "mt1", "RATTLE_POA_MT_SLOTS": "3"}])
def test_fallback_paths_match_oracle(gpu_ctx, oracle, monkeypatch, env):
    """The slow paths behind the fast ones stay exact: full topological sort for ties (bit 0), traceback without
    the LDS chain (bit 1), packs re-run with a larger arena after a node-capacity overflow; and every form of the row loop
    (RATTLE_POA_MODE: the barrier form, teams of wavefronts on 4 / 2 / 1 teams -- the last with
    rings so short that many predecessors come from the record in HBM)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    packs = _packs_from_synth(400, 10, seed=11)
    rows, width, counters = gpu_ctx.poa_msa(packs)
    for p, pack in enumerate(packs):
        want, _ = oracle.poa_msa(pack)
        assert rows[p] == want, p


@pytest.mark.parametrize("env", [{}, {"RATTLE_POA_MODE": "dense"}, {"RATTLE_POA_MODE": "mt4"},
                                 {"RATTLE_POA_MODE": "mt2"}, {"RATTLE_POA_MODE": "mt1"}, {"RATTLE_POA_MODE": "mt4", "RATTLE_POA_MT_SLOTS": "11"}])
def test_predecessors_hundreds_of_rows_back_and_many_in_edges(gpu_ctx, oracle, monkeypatch, env):
    """The row loop reads a COMPACT plan record: the distances to a row's first eight predecessor rows in a byte each, saturated
    at 255, the in-degree capped at 255 (poa.hip, round 4).  Reads that skip 300-600 bases of the others (an exon left out) give
    nodes whose predecessor lies far more than 255 rows back; reads that resume at many different places give one node more than
    eight in-edges (the edge-list walk); both next to ordinary rows, in the barrier form and the team forms of the loop."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    tx = acgt[rng.integers(0, 4, 1400)]

    def noisy(a, err=0.06):
        r = rng.random(len(a))
        b = a.copy()
        sub = r < err * 0.4
        b[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        return b[(r >= err * 0.7) | (r < err * 0.4)]          # a few deletions too

    pack = [noisy(tx).tobytes() for _ in range(10)]
    # exon skipping: 300 .. 620 bases left out at different places
    for a, n in ((200, 300), (450, 620), (800, 410), (150, 505)):
        pack.append(noisy(np.concatenate([tx[:a], tx[a + n:]])).tobytes())
    # many different resume points into the same downstream node: prefixes of different lengths glued to the common tail from 1000 on
    for cut in range(300, 960, 55):
        pack.append(noisy(np.concatenate([tx[:cut], tx[1000:]]), 0.03).tobytes())
    pack.sort(key=lambda s: -len(s))
    rows, width, counters = gpu_ctx.poa_msa([pack, pack[::-1]])
    for got, p in zip(rows, (pack, pack[::-1])):
        want, _ = oracle.poa_msa(p)
        assert got == want
