"""The synthetic generator is seeded and deterministic; the packed variant follows the same model."""
import difflib

import numpy as np

from rattle_amd import synth


def test_reads_deterministic_and_well_formed():
    a = synth.reads(50, 4, 2, True, seed=5)
    b = synth.reads(50, 4, 2, True, seed=5)
    assert a[0] == b[0] and a[1] == b[1]
    assert all(set(s) <= set(b"ACGT") for s in a[0])
    assert all(len(s) == len(q) for s, q in zip(a[0], a[1]))
    assert all(36 <= min(q) and max(q) <= 73 for q in a[1])


def test_reads_packed_model():
    cat, q, off, tid, flip = synth.reads_packed(400, 5, 1, True, seed=7, exon=(50, 210))
    cat2, q2, off2, _, _ = synth.reads_packed(400, 5, 1, True, seed=7, exon=(50, 210))
    assert np.array_equal(cat, cat2) and np.array_equal(off, off2) and np.array_equal(q, q2)
    L = np.diff(off.astype(np.int64))
    assert len(cat) == L.sum() == len(q) and 700 < L.mean() < 1300
    assert set(np.unique(cat)) <= set(b"ACGT") and 0.3 < flip.mean() < 0.7
    tx, _ = synth.transcriptome(5, 1, exon=(50, 210))
    i = int(np.nonzero(flip == 0)[0][0])
    r = difflib.SequenceMatcher(None, cat[int(off[i]):int(off[i + 1])].tobytes(), tx[tid[i]].tobytes(), autojunk=False).ratio()
    assert r > 0.85
