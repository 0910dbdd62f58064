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
