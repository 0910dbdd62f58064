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
