"""Pin the POA + correction oracle against toyset/rna/output/consensi.fq and uncorrected.fq.

The fixture was produced by an older RATTLE build whose column-vote tie order had A before C
("U-GTAC"; the current source gives "U-GTCA" with libstdc++, see oracle/orc_correct.hpp and
test_vote_order_is_libstdcxx_iteration_order below).  With that order selected, every single-pack
cluster's consensus is reproduced exactly; the 8 multi-pack clusters (> 200 reads, POA #3,
correct.cpp:489-556) depend on the order in which the fixture's worker threads finished their packs
(correct.cpp:469): PACK_ORDER records, per cluster, the completion order under which the fixture's
sequence is reproduced (found by search over the permutations; 2 of the 8 need none).  The whole-fixture
check -- all 175 consensi and the 739 uncorrected records in order -- runs by default with the oracle's AVX2 int16 POA rows
(test_whole_fixture_with_simd_rows, ~90 s) and, opt-in, with the scalar rows (RATTLE_SLOW=1, ~5 CPU-minutes); the same check runs
through the HIP path by default (tests/test_gpu_correct.py).
"""
import gzip
import os
import re

import pytest

from conftest import GOLDEN
from rattle_amd import hps

# pack orders (found by search) under which the fixture's multi-pack consensi are reproduced
PACK_ORDER = {29: [2, 1, 0], 51: [1, 0], 59: [2, 0, 1], 75: [2, 3, 1, 0], 111: [2, 1, 0], 320: [1, 0], 44: [0, 1], 291: [0, 1]}


def fixture_consensi():
    lines = gzip.open(os.path.join(GOLDEN, "toyset_rna.consensi.fq.gz"), "rt").read().split("\n")
    out = {}
    for i in range(0, len(lines) - 1, 4):
        cid = int(re.match(r"@cluster_(\d+) reads=(\d+)", lines[i]).group(1))
        out[cid] = lines[i + 1]
    return out


def run_subset(oracle, toyset, toyset_clusters, cids, pack_order=None):
    """`correct` restricted to the chosen clusters (cluster ids are preserved by keeping
    empty placeholders out: we renumber and map back)."""
    sub = [toyset_clusters[c] for c in cids]
    b = hps.encode(sub, fields=3)
    headers = [r[0] for r in toyset]
    seqs = [r[1] for r in toyset]
    quals = [r[2] for r in toyset]
    po = {cids.index(c): v for c, v in (pack_order or {}).items() if c in cids}
    corrected, uncorrected, consensi, counters = oracle.correct(headers, seqs, quals, b, pack_order=po)
    lines = consensi.decode().split("\n")
    got = {}
    for i in range(0, len(lines) - 1, 4):
        local = int(re.match(r"@gene_cluster_(\d+) reads=(\d+)", lines[i]).group(1))
        got[cids[local]] = lines[i + 1]
    return got, uncorrected.decode(), counters


def test_consensi_fixture_subset(oracle, toyset, toyset_clusters):
    """30 clusters spread over the size range 6..~60 reads (about half a minute of CPU)."""
    want = fixture_consensi()
    sizes = sorted((len(toyset_clusters[c][1]), c) for c in want if len(toyset_clusters[c][1]) <= 60)
    cids = sorted(c for _, c in sizes[::4][:30])
    oracle.set_cv_order(b"U-GTAC")
    try:
        got, _, counters = run_subset(oracle, toyset, toyset_clusters, cids)
    finally:
        oracle.set_cv_order(b"U-GTCA")
    assert set(got) == set(cids)
    for c in cids:
        assert got[c] == want[c], f"cluster {c}"
    assert counters[0] > 0


def test_current_source_vote_order_differs_only_in_ties(oracle, toyset, toyset_clusters):
    """With the CURRENT source's order the same clusters differ from the old fixture only by
    same-length A<->C substitutions (tie columns)."""
    want = fixture_consensi()
    cids = [1, 7, 8]
    got, _, _ = run_subset(oracle, toyset, toyset_clusters, cids)
    for c in cids:
        assert len(got[c]) == len(want[c])
        diff = {(a, b) for a, b in zip(got[c], want[c]) if a != b}
        assert diff <= {("C", "A")}


@pytest.mark.skipif(not os.environ.get("RATTLE_SLOW"), reason="4 CPU-minutes; RATTLE_SLOW=1 to run")
def test_all_single_pack_consensi_and_uncorrected(oracle, toyset, toyset_clusters):
    want = fixture_consensi()
    cids = sorted(c for c in want if len(toyset_clusters[c][1]) <= 200)
    oracle.set_cv_order(b"U-GTAC")
    try:
        got, _, _ = run_subset(oracle, toyset, toyset_clusters, cids)
    finally:
        oracle.set_cv_order(b"U-GTCA")
    assert all(got[c] == want[c] for c in cids) and len(cids) == 167


@pytest.mark.skipif(not os.environ.get("RATTLE_SLOW"), reason="~6 CPU-minutes; RATTLE_SLOW=1 to run")
def test_whole_fixture_consensi_and_uncorrected(oracle, toyset, toyset_clusters):
    """Every cluster of the fixture: 175 consensi (the 8 multi-pack ones with the recorded completion orders)
    and uncorrected.fq's 739 records, in order."""
    want = fixture_consensi()
    cids = list(range(len(toyset_clusters)))
    oracle.set_cv_order(b"U-GTAC")
    try:
        got, unc, _ = run_subset(oracle, toyset, toyset_clusters, cids, PACK_ORDER)
    finally:
        oracle.set_cv_order(b"U-GTCA")
    assert len(want) == 175 and set(got) == set(want)
    bad = [c for c in want if got[c] != want[c]]
    assert not bad, bad
    ids = [l.split(",")[0] for l in unc.split("\n")[0::4] if l]
    assert ids == open(os.path.join(GOLDEN, "toyset_rna.uncorrected.ids")).read().split()


def test_whole_fixture_with_simd_rows(oracle, toyset, toyset_clusters):
    """The whole fixture by DEFAULT (round 6; until then only the HIP path ran all of it unasked): every cluster of the reference's
    clusters.out through the oracle with its AVX2 int16 POA rows -- the same H / F / E values and traceback as the scalar rows
    (test_avx2_row_fill_gives_the_scalar_alignments below checks that on whole packs), a quarter of the time -- must give the 175 consensi of
    consensi.fq and the 739 records of uncorrected.fq, in order.  (The scalar-row run of the same check stays opt-in: RATTLE_SLOW=1.)"""
    if not oracle.set_poa_simd(True):
        pytest.skip("no AVX2 on this host: the scalar rows take 5 CPU-minutes (RATTLE_SLOW=1)")
    want = fixture_consensi()
    cids = list(range(len(toyset_clusters)))
    oracle.set_cv_order(b"U-GTAC")
    try:
        got, unc, _ = run_subset(oracle, toyset, toyset_clusters, cids, PACK_ORDER)
    finally:
        oracle.set_cv_order(b"U-GTCA")
        oracle.set_poa_simd(False)
    assert len(want) == 175 and set(got) == set(want)
    bad = [c for c in want if got[c] != want[c]]
    assert not bad, bad
    ids = [l.split(",")[0] for l in unc.split("\n")[0::4] if l]
    assert ids == open(os.path.join(GOLDEN, "toyset_rna.uncorrected.ids")).read().split()


def test_vote_order_is_libstdcxx_iteration_order(tmp_path):
    """correct.cpp:105-110 inserts A, C, T, U, G, '-' into an unordered_map<char, ...> and :174 iterates it:
    the tie order of the column vote is that container's iteration order.  Measured here with the host's
    libstdc++ (the product default "U-GTCA" and the oracle default must equal it)."""
    import subprocess
    src = tmp_path / "probe.cpp"
    src.write_text('#include <cstdio>\n#include <unordered_map>\nint main(){std::unordered_map<char,int> m;'
                   'for(char c:{\'A\',\'C\',\'T\',\'U\',\'G\',\'-\'})m[c]=0;for(auto&kv:m)putchar(kv.first);return 0;}\n')
    exe = tmp_path / "probe"
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-o", str(exe), str(src)])
    order = subprocess.check_output([str(exe)]).decode()
    assert order == "U-GTCA"
    from rattle_amd import _lib
    assert b'"U-GTCA"' in open(os.path.join(os.path.dirname(_lib.__file__), "csrc", "correct_driver.hip"), "rb").read()
    assert b'"U-GTCA"' in open(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "orc_correct.hpp"), "rb").read()


def test_fix_msa_ends_hand_built_rows(oracle):
    """correct.cpp:32-92 on three hand-built rows, expectations worked out by hand from the reference's control flow:
    (a) a block of fewer than 10 bases followed by >= 20 gaps at the LEFT end is blanked and its bases leave seq / quality;
    (b) the same at the RIGHT end happens in the second phase, on the reversed row: the LAST bases leave;
    (c) phase 1 stops at a small block followed by only 15 gaps (reverses), phase 2 then finds the same block followed by 25
        gaps and blanks it, runs off the row end and never reverses back: the row is all gaps and the read is empty (the
        "stays reversed" quirk has nothing left to show);
    (d) a 9-base block followed by 19 gaps stays (the limit is >= 20), and 4 gaps inside a block split it, 3 do not."""
    G = lambda k: b"-" * k
    rows = [b"ACGTACGTA" + G(25) + b"CCCCCGGGGGTTTTT",
            b"CCCCCGGGGGTTTTT" + G(22) + b"ACGTACGT" + G(4),
            G(25) + b"ACGTTGCAA" + G(15),
            b"ACGTACGTA" + G(19) + b"CCCCCGGGGGTTTTT" + G(6),
            b"ACGTA---CGTAC" + G(20) + b"GGGGGCCCCC" + G(6)]
    seqs = [r.replace(b"-", b"") for r in rows]
    quals = [bytes(range(40, 40 + len(s))) for s in seqs]
    out_rows, out_s, out_q = oracle.fix_msa_ends(rows, seqs, quals)
    assert out_rows[0] == G(34) + b"CCCCCGGGGGTTTTT" and out_s[0] == b"CCCCCGGGGGTTTTT" and out_q[0] == quals[0][9:]
    assert out_rows[1] == b"CCCCCGGGGGTTTTT" + G(34) and out_s[1] == b"CCCCCGGGGGTTTTT" and out_q[1] == quals[1][:15]
    assert out_rows[2] == G(49) and out_s[2] == b"" and out_q[2] == b""
    assert out_rows[3] == rows[3] and out_s[3] == seqs[3] and out_q[3] == quals[3]
    # "ACGTA---CGTAC" is ONE block of 10 bases (3 gaps do not end it): nothing is cut although 20 gaps follow
    assert out_rows[4] == rows[4] and out_s[4] == seqs[4]


def test_avx2_row_fill_gives_the_scalar_alignments(oracle):
    """The AVX2 int16 fill of H / F / E (bench.py's CPU baseline) against the scalar restatement: identical MSAs on noisy packs
    (deep graphs, ties, long gaps) and identical `correct` outputs."""
    import numpy as np
    from rattle_amd import hps, synth
    if not oracle.set_poa_simd(False):
        pytest.skip("no AVX2 on this CPU")
    seqs, quals, tid, _ = synth.reads(260, 4, 1, False, seed=3)
    packs = [[seqs[i] for i in range(len(seqs)) if tid[i] == g][:40] for g in range(4)]
    packs.append([b"ACGT" * 30, b"ACGT" * 28 + b"AC", b"CGT" + b"ACGT" * 29, b"ACGA" * 30, b"TTTTTTTT", b"ACGT" * 12 + b"G" * 40 + b"ACGT" * 18])
    headers = [b"@r%d" % i for i in range(len(seqs))]
    clusters = [((int(np.nonzero(tid == g)[0][0]), 0, -1), [(int(i), 0, -1) for i in np.nonzero(tid == g)[0]]) for g in range(4)]
    want = [oracle.poa_msa(p)[0] for p in packs]
    want_c = oracle.correct(headers, seqs, quals, hps.encode(clusters), split=30)
    try:
        oracle.set_poa_simd(True)
        got = [oracle.poa_msa(p)[0] for p in packs]
        got_c = oracle.correct(headers, seqs, quals, hps.encode(clusters), split=30)
    finally:
        oracle.set_poa_simd(False)
    assert got == want
    assert got_c[:3] == want_c[:3]
