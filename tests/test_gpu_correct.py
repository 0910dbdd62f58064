"""Parity of the HIP `correct` path (pack builder + kernel C x3 + kernel D post-MSA logic) with the
oracle and with the reference's shipped consensi fixture."""
import gzip
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN
from rattle_amd import hps, synth
from rattle_amd.api import cluster_command, correct_command

pytestmark = pytest.mark.gpu


def test_correct_synthetic_cdna_matches_oracle(gpu_ctx, oracle):
    """cluster (both strands) then correct on 700 synthetic reads; split=40 forces multi-pack
    clusters (POA #3) and reverse-strand members exercise the in-place reverse complement."""
    seqs, quals, _, _ = synth.reads(700, 5, 1, True, seed=8)
    headers = [b"@r%d" % i for i in range(len(seqs))]
    clusters, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))))
    assert any(s[1] for _, mem in clusters for s in mem)
    assert any(len(mem) > 40 for _, mem in clusters)
    got = correct_command(gpu_ctx, headers, seqs, quals, clusters, split=40)
    want = oracle.correct(headers, seqs, quals, hps.encode(clusters), split=40)
    assert got[0] == want[0], "corrected.fq differs"
    assert got[1] == want[1], "uncorrected.fq differs"
    assert got[2] == want[2], "consensi.fq differs"
    assert int(got[3][0]) == int(want[3][0])           # DP cells, exact


def test_correct_big_cluster_stage_split_matches_oracle(gpu_ctx, oracle, monkeypatch):
    """Clusters with many packs take POA #2 first and share a pass with the others' POA #2 for their
    POA #3 (correct_driver.hip); forced here on a small input (thresholds are read per call)."""
    seqs, quals, _, _ = synth.reads(900, 12, 1, True, seed=12)
    headers = [b"@r%d" % i for i in range(len(seqs))]
    clusters, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))))
    want = oracle.correct(headers, seqs, quals, hps.encode(clusters), split=25)
    sizes = sorted(len(mem) for _, mem in clusters)
    assert sizes[-1] > 75 and sizes[0] < 50                     # clusters with >= 3 packs and with < 3 packs exist
    monkeypatch.setenv("RATTLE_BIG_CLUSTER_PACKS", "3")
    monkeypatch.setenv("RATTLE_BIG_MIN_PACKS", "0")
    for _ in range(2):
        got = correct_command(gpu_ctx, headers, seqs, quals, clusters, split=25)
        assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
        assert int(got[3][0]) == int(want[3][0])


def test_correct_toyset_subset_matches_reference_fixture(gpu_ctx, toyset, toyset_clusters):
    """40 toyset clusters (6..200 reads) against toyset/rna/output/consensi.fq, with the old
    fixture build's vote order (see tests/test_oracle_correct.py)."""
    lines = gzip.open(os.path.join(GOLDEN, "toyset_rna.consensi.fq.gz"), "rt").read().split("\n")
    want = {}
    for i in range(0, len(lines) - 1, 4):
        want[int(re.match(r"@cluster_(\d+) ", lines[i]).group(1))] = lines[i + 1]
    sizes = sorted((len(toyset_clusters[c][1]), c) for c in want if len(toyset_clusters[c][1]) <= 200)
    cids = sorted(c for _, c in sizes[::4])[:40] + [sizes[-1][1]]
    sub = [toyset_clusters[c] for c in cids]
    headers = [r[0] for r in toyset]; seqs = [r[1] for r in toyset]; quals = [r[2] for r in toyset]
    res = gpu_ctx.correct_reads(seqs, quals, sub, vote_order=b"U-GTAC")
    got = {cids[r[1]]: r[3].decode() for r in res["consensi"]}
    assert set(got) == set(cids)
    bad = [c for c in cids if got[c] != want[c]]
    assert not bad, bad


def test_staged_reads_give_identical_results(gpu_ctx):
    """rattle_hip_stage_reads: cluster + correct on HBM-resident reads == the same calls on host buffers."""
    cat, qcat, off, _, _ = synth.reads_packed(3000, 12, 1, True, seed=5, exon=(50, 210))
    cl_a = gpu_ctx.cluster_unsorted_packed(cat, off)
    res_a = gpu_ctx.correct_packed(cat, qcat, off, cl_a, split=40, digest=True)
    gpu_ctx.stage_reads(cat, qcat, off)
    try:
        cl_b = gpu_ctx.cluster_unsorted_packed(cat, off)
        res_b = gpu_ctx.correct_packed(cat, qcat, off, cl_b, split=40, digest=True)
        # different arrays with the same content are not the staged buffers: falls back to uploading
        res_c = gpu_ctx.correct_packed(cat.copy(), qcat.copy(), off, cl_b, split=40, digest=True)
    finally:
        gpu_ctx.unstage_reads()
    assert cl_a.as_list() == cl_b.as_list()
    assert res_a[:3] == res_b[:3] == res_c[:3] and res_a[4] == res_b[4] == res_c[4]
    assert np.array_equal(res_a[3][:3], res_b[3][:3])


def test_correct_with_a_cluster_of_long_reads(gpu_ctx, oracle):
    """A cluster of ~7 kb reads (segmented int32 POA rows) next to ordinary ones, through cluster + correct."""
    rng = np.random.default_rng(23)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seqs, quals, _, _ = synth.reads(150, 3, 1, True, seed=2)
    tx = acgt[rng.integers(0, 4, 7000)]
    for _ in range(8):
        r = rng.random(len(tx))
        s = tx.copy()
        sub = (r >= 0.02) & (r < 0.05)
        s[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        s = s[r >= 0.02]
        seqs.append(s.tobytes())
        quals.append(bytes(rng.integers(40, 70, len(s)).astype(np.uint8)))
    headers = [b"@r%d" % i for i in range(len(seqs))]
    clusters, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))))
    assert any(len(mem) >= 8 and len(seqs[mem[0][0]]) > 6144 for _, mem in clusters)
    got = correct_command(gpu_ctx, headers, seqs, quals, clusters)
    want = oracle.correct(headers, seqs, quals, hps.encode(clusters))
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]


def test_whole_toyset_fixture_consensi_and_uncorrected(gpu_ctx, toyset, toyset_clusters):
    """The reference's whole `correct` fixture through the HIP path: all 175 consensi of
    toyset/rna/output/consensi.fq -- including the 8 clusters of more than 200 reads that go through POA #3
    (correct.cpp:489-556), fed in the worker completion orders recorded in tests/test_oracle_correct.py --
    and the 739 records of uncorrected.fq, in order (old fixture build's vote order)."""
    from test_oracle_correct import PACK_ORDER, fixture_consensi
    want = fixture_consensi()
    seqs = [r[1] for r in toyset]; quals = [r[2] for r in toyset]
    res = gpu_ctx.correct_reads(seqs, quals, toyset_clusters, vote_order=b"U-GTAC", pack_order=PACK_ORDER)
    got = {r[1]: r[3].decode() for r in res["consensi"]}
    assert len(want) == 175 and set(got) == set(want)
    bad = [c for c in want if got[c] != want[c]]
    assert not bad, bad
    multi = [c for c in want if len(toyset_clusters[c][1]) > 200]
    assert sorted(multi) == sorted(PACK_ORDER) and len(multi) == 8
    ids = [toyset[r[0]][0].decode() for r in res["uncorrected"]]
    assert ids == open(os.path.join(GOLDEN, "toyset_rna.uncorrected.ids")).read().split()
    assert len(res["corrected"]) + len(res["uncorrected"]) == len(toyset) and res["skipped"] == []


def test_pack_consensus_order_matches_oracle(gpu_ctx, oracle):
    """rattle_correct_params::pack_order (the completion order of correct.cpp:469): same permutation through
    the oracle and through the HIP path, on clusters with 3+ packs; a rejected non-permutation."""
    from rattle_amd._lib import RattleError
    seqs, quals, _, _ = synth.reads(700, 5, 1, True, seed=8)
    headers = [b"@r%d" % i for i in range(len(seqs))]
    clusters, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))))
    big = [c for c, (_, mem) in enumerate(clusters) if len(mem) > 80]
    assert big
    po = {}
    for c in big:
        n = len(clusters[c][1])
        nf = (n - 1) // 40 + 1
        po[c] = list(reversed(range(nf)))
    base = correct_command(gpu_ctx, headers, seqs, quals, clusters, split=40)
    got = correct_command(gpu_ctx, headers, seqs, quals, clusters, split=40, pack_order=po)
    want = oracle.correct(headers, seqs, quals, hps.encode(clusters), split=40, pack_order=po)
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
    assert got[0] == base[0] and got[1] == base[1]                 # only the cluster consensi can depend on the order
    with pytest.raises(RattleError):
        correct_command(gpu_ctx, headers, seqs, quals, clusters, split=40, pack_order={big[0]: [0, 0, 1]})


def test_uncorrected_records_keep_their_annotation_line(gpu_ctx, oracle):
    """correct.cpp:362-366,289-293 push the ORIGINAL read_t (third line included) into uncorrected.fq."""
    seqs, quals, _, _ = synth.reads(200, 6, 1, True, seed=3)
    headers = [b"@r%d" % i for i in range(len(seqs))]
    ann = [b"+note%d" % i for i in range(len(seqs))]
    clusters, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))))
    out = correct_command(gpu_ctx, headers, seqs, quals, clusters, ann=ann, min_reads=30)       # clusters of <= 30 reads stay uncorrected
    lines = out[1].split(b"\n")
    assert len(lines) > 4
    for i in range(0, len(lines) - 1, 4):
        rid = int(lines[i].split(b",")[0][2:])
        assert lines[i + 2] == ann[rid]
    assert all(l == b"+" for l in out[0].split(b"\n")[2::4])


def test_packs_beyond_the_device_are_skipped_and_reported(gpu_ctx, oracle, monkeypatch):
    """A pack whose DP record does not fit the arena is left out, not fatal: its reads come back untouched in
    `uncorrected`, it contributes no consensus, it is listed in rattle_correction::skipped, and everything
    else equals the oracle run on the remaining packs.  Forced with a tiny arena budget; then the static
    max_pack_cells rule."""
    seqs, quals, _, _ = synth.reads(300, 4, 1, True, seed=4)
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    tx = acgt[rng.integers(0, 4, 5200)]
    long_ids = []
    for _ in range(9):                       # one cluster of 5 kb reads: ~16 x the DP record of the 1 kb clusters
        r = rng.random(len(tx))
        s = tx.copy()
        sub = (r >= 0.03) & (r < 0.06)
        s[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        s = s[r >= 0.03]
        long_ids.append(len(seqs))
        seqs.append(s.tobytes())
        quals.append(bytes(rng.integers(40, 70, len(s)).astype(np.uint8)))
    headers = [b"@r%d" % i for i in range(len(seqs))]
    clusters, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))))
    lc = [c for c, (_, mem) in enumerate(clusters) if mem[0][0] in long_ids]
    assert len(lc) == 1 and len(clusters[lc[0]][1]) == 9
    full = correct_command(gpu_ctx, headers, seqs, quals, clusters, with_skipped=True)
    assert full[4] == []
    # without the long cluster (made a singleton list so cluster ids stay the same): what a skip must leave
    rest = [(m, mem if c != lc[0] else mem[:0]) for c, (m, mem) in enumerate(clusters)]
    want = oracle.correct(headers, seqs, quals, hps.encode([(m, mem) if mem else (m, [m]) for m, mem in rest]))
    # (a) dynamic: the arena is too small for the long pack
    monkeypatch.setenv("RATTLE_POA_BUDGET_MB", "64")
    got = correct_command(gpu_ctx, headers, seqs, quals, clusters, with_skipped=True)
    monkeypatch.delenv("RATTLE_POA_BUDGET_MB")
    sk = got[4]
    assert len(sk) == 1 and sk[0]["cluster"] == lc[0] and sk[0]["stage"] == 1 and sorted(sk[0]["reads"]) == sorted(long_ids)
    assert int(got[3][3]) == 1 and int(got[3][4]) == 9
    assert got[0] == want[0]                                       # corrected reads of every other pack
    assert got[2] == want[2]                                       # and every other cluster's consensus; none for the skipped one
    unc = got[1].split(b"\n")
    by_id = {int(unc[i].split(b",")[0][2:]): (unc[i + 1], unc[i + 3]) for i in range(0, len(unc) - 1, 4)}
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for (rid, rev, _) in clusters[lc[0]][1]:
        s, q = by_id[rid]
        assert s == (seqs[rid].translate(comp)[::-1] if rev else seqs[rid]) and q == (quals[rid][::-1] if rev else quals[rid])
    n_rec = lambda t: t.count(b"\n") // 4
    assert n_rec(got[0]) + n_rec(got[1]) == len(seqs)
    # (b) static rule: (6 L + 64) L cells for the longest read of the pack
    got2 = correct_command(gpu_ctx, headers, seqs, quals, clusters, with_skipped=True, max_pack_cells=60_000_000)
    assert len(got2[4]) == 1 and got2[4][0]["stage"] == 0 and sorted(got2[4][0]["reads"]) == sorted(long_ids)
    assert got2[0] == want[0] and n_rec(got2[0]) + n_rec(got2[1]) == len(seqs)


def test_fix_msa_ends_trims_through_the_hip_path(gpu_ctx, oracle):
    """Packs built so that fix_msa_ends (correct.cpp:32-92, kernel D) has something to cut: a read with 9 junk bases in
    front of the shared core while a longer read's unaligned 29-base prefix opens 29 columns between them (left end, phase
    1), the mirror image at the right end (phase 2, on the reversed row), and a 9-base read that aligns nowhere (its row is
    blanked whole, the read comes back empty in uncorrected.fq).  Byte-identical to the oracle, and the cuts really happen."""
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda k: acgt[rng.integers(0, 4, k)].tobytes()

    def noisy(s, rate=0.02):
        a = np.frombuffer(s, np.uint8).copy()
        m = rng.random(len(a)) < rate
        a[m] = acgt[rng.integers(0, 4, int(m.sum()))]
        return a.tobytes()

    seqs, clusters = [], []

    def cluster_of(members):
        base = len(seqs)
        seqs.extend(members)
        ids = sorted(range(base, base + len(members)), key=lambda i: -len(seqs[i]))
        clusters.append(((ids[0], 0, -1), [(i, 0, -1) for i in ids]))

    core = rnd(300)
    # left end: R1 = junk9 + core + tail25 (longest, first into the graph), R2 = junk29' + core (its prefix stays unaligned)
    cluster_of([b"GATTACAGA" + core + rnd(25), rnd(29) + noisy(core)] + [noisy(core) for _ in range(6)])
    core2 = rnd(320)
    # right end: first lead40 + core2 + junk29, then core2 + junk9 (9 unaligned bases behind 29 foreign columns)
    cluster_of([rnd(40) + core2 + b"A" * 29, noisy(core2) + b"CGCGTCGCG"] + [noisy(core2) for _ in range(6)])
    core3 = rnd(280)
    cluster_of([core3] + [noisy(core3) for _ in range(6)] + [b"ACGGTCAAT"])        # a 9-base member that aligns nowhere
    quals = [bytes(rng.integers(40, 70, len(s)).astype(np.uint8)) for s in seqs]
    headers = [b"@t%d" % i for i in range(len(seqs))]
    got = correct_command(gpu_ctx, headers, seqs, quals, clusters)
    want = oracle.correct(headers, seqs, quals, hps.encode(clusters))
    assert got[0] == want[0], "corrected.fq differs"
    assert got[1] == want[1], "uncorrected.fq differs"
    assert got[2] == want[2], "consensi.fq differs"
    cor = {l.split(b",")[0]: s for l, s in zip(got[0].split(b"\n")[0::4], got[0].split(b"\n")[1::4])}
    assert not cor[b"@t0"].startswith(b"GATTACAGA") and len(cor[b"@t0"]) <= 300 + 25 + 3        # the junk prefix was cut
    assert not cor[b"@t9"].endswith(b"CGCGTCGCG") and len(cor[b"@t9"]) <= 320 + 3                # the junk suffix was cut
    unc = got[1].split(b"\n")
    k = [i for i, l in enumerate(unc) if l.startswith(b"@t%d," % (len(seqs) - 1))]
    assert k and unc[k[0] + 1] == b""                                                            # blanked whole: empty read


def test_correct_outputs_do_not_depend_on_scheduling(gpu_ctx, monkeypatch):
    """Round 3's two-flow experiment (never merged) once showed ONE differing quality symbol between two schedules of the same job.
    Byte-exactness must not depend on how the device happens to run the packs: the same `correct` job is repeated with one POA
    stream and with twelve, with the big-cluster stage split forced and not, with every form of kernel C's row loop, with and without the head start of
    the few-workgroup groups, and
    while a second context on a second host thread keeps the device busy with another job (its kernels interleave with this
    job's, its allocations move this job's buffers) -- every output byte, the skip list and the work counters must stay the
    same."""
    import threading
    from rattle_amd.api import Context
    seqs, quals, _, _ = synth.reads(2500, 9, 1, True, seed=33)
    headers = [b"@r%d" % i for i in range(len(seqs))]
    clusters, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))))
    assert max(len(mem) for _, mem in clusters) > 120

    def run():
        out = correct_command(gpu_ctx, headers, seqs, quals, clusters, split=30, with_skipped=True)
        return out[0], out[1], out[2], [int(x) for x in out[3][:5]], out[4]

    base = run()
    variants = [{"RATTLE_POA_STREAMS": "1"}, {"RATTLE_POA_STREAMS": "12"}, {"RATTLE_BIG_CLUSTER_PACKS": "3", "RATTLE_BIG_MIN_PACKS": "0"},
                {"RATTLE_POA_MODE": "mt4", "RATTLE_POA_STREAMS": "2"}, {"RATTLE_POA_MODE": "mt2"}, {"RATTLE_POA_MODE": "mt1"},
                {"RATTLE_POA_MODE": "dense"}, {"RATTLE_POA_HEAD_START_US": "0"}, {"RATTLE_POA_HEAD_START_US": "200", "RATTLE_POA_STREAMS": "3"}]
    for env in variants:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert run() == base, env
        for k in env:
            monkeypatch.delenv(k)
    # a noisy neighbour on the same device
    stop = threading.Event()
    s2, q2, _, _ = synth.reads(1200, 4, 1, True, seed=34)

    def neighbour():
        c2 = Context(0)
        h2 = [b"@n%d" % i for i in range(len(s2))]
        cl2, _ = cluster_command(c2, s2, list(range(len(s2))))
        while not stop.is_set():
            correct_command(c2, h2, s2, q2, cl2, split=50)
        c2.close()

    t = threading.Thread(target=neighbour)
    t.start()
    try:
        for _ in range(3):
            assert run() == base, "result changed with a second job on the device"
    finally:
        stop.set()
        t.join()
