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


def test_par_baseline_helper_runs(tmp_path):
    """oracle/par_baseline.py (bench.py's all-cores CPU leg): one oracle task per transcript on a process pool."""
    import json
    import os
    import subprocess
    import sys

    import numpy as np

    from rattle_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cat, qcat, off, tid, _ = synth.reads_packed(120, 4, 1, True, seed=9, exon=(20, 60))
    path = str(tmp_path / "s.npz")
    np.savez(path, cat=cat, qcat=qcat, off=off, grp=tid)
    r = subprocess.run([sys.executable, os.path.join(root, "oracle", "par_baseline.py"), path, "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["reads"] == 120 and j["workers"] == 2 and j["tasks"] == len(set(tid.tolist()))
