"""The drop-in CLI (`rattle cluster` / `rattle correct`) against the reference fixture and against
the oracle CLI, comparing output FILES."""
import gzip
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT
from rattle_amd import hps, synth

pytestmark = pytest.mark.gpu
RATTLE = os.path.join(ROOT, "rattle_amd", "csrc", "rattle")
ORACLE = os.path.join(ROOT, "oracle", "oracle_cli")


@pytest.fixture(scope="module")
def built(oracle):
    if not os.path.exists(RATTLE):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.dirname(RATTLE)])
    assert os.path.exists(ORACLE)


def test_cluster_cli_reproduces_toyset_fixture(built, tmp_path):
    fq = tmp_path / "sample.fastq"
    fq.write_bytes(gzip.open(os.path.join(GOLDEN, "toyset_rna.fastq.gz")).read())
    out = subprocess.run([RATTLE, "cluster", "-i", str(fq), "-o", str(tmp_path), "--rna", "--lower-length", "0", "-t", "4"],
                         capture_output=True, text=True, check=True)
    assert "Reads: 8306" in out.stdout
    got = hps.decode((tmp_path / "clusters.out").read_bytes(), fields=3)
    want = hps.decode(open(os.path.join(GOLDEN, "toyset_rna.clusters.out"), "rb").read(), fields=2)
    assert got == want


def test_cluster_and_correct_cli_match_oracle_cli(built, tmp_path):
    """gz input, length filter (150..100000 drops short reads and shifts nothing: ids are record
    indices), --iso two-level clustering, then correct; every output file byte-identical."""
    seqs, quals, _, _ = synth.reads(500, 4, 2, True, seed=17)
    seqs[5] = seqs[5][:100]; quals[5] = quals[5][:100]              # filtered out by --lower-length 150
    seqs[9] = seqs[9][:40] + b"N" + seqs[9][41:]                    # skipped: contains N
    text = synth.fastq_text(seqs, quals)
    (tmp_path / "in.fastq.gz").write_bytes(gzip.compress(text))
    (tmp_path / "in2.fastq").write_bytes(text)
    a = tmp_path / "a"; b = tmp_path / "b"
    a.mkdir(); b.mkdir()
    subprocess.run([RATTLE, "cluster", "-i", str(tmp_path / "in.fastq.gz"), "-o", str(a), "--iso"], check=True, capture_output=True)
    subprocess.run([ORACLE, "cluster", "-i", str(tmp_path / "in2.fastq"), "-o", str(b), "--iso"], check=True, capture_output=True)
    assert (a / "clusters.out").read_bytes() == (b / "clusters.out").read_bytes()
    subprocess.run([RATTLE, "correct", "-i", str(tmp_path / "in2.fastq"), "-c", str(a / "clusters.out"), "-o", str(a), "-s", "30"],
                   check=True, capture_output=True)
    subprocess.run([ORACLE, "correct", "-i", str(tmp_path / "in2.fastq"), "-c", str(b / "clusters.out"), "-o", str(b), "-s", "30"],
                   check=True, capture_output=True)
    for f in ("corrected.fq", "uncorrected.fq", "consensi.fq"):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f
    assert (a / "consensi.fq").read_bytes().startswith(b"@transcript_cluster_0 gene_cluster_0 reads=")


def test_polish_cli_reproduces_transcriptome_fixture_and_oracle(built, tmp_path):
    """`rattle polish --rna` on the shipped consensi.fq == toyset/rna/output/transcriptome.fq (175
    singletons, length-sorted, same total_reads) and == the oracle CLI byte for byte."""
    import re
    (tmp_path / "consensi.fq").write_bytes(gzip.open(os.path.join(GOLDEN, "toyset_rna.consensi.fq.gz")).read())
    a = tmp_path / "a"; b = tmp_path / "b"
    a.mkdir(); b.mkdir()
    subprocess.run([RATTLE, "polish", "-i", str(tmp_path / "consensi.fq"), "-o", str(a), "--rna"], check=True, capture_output=True)
    subprocess.run([ORACLE, "polish", "-i", str(tmp_path / "consensi.fq"), "-o", str(b), "--rna"], check=True, capture_output=True)
    got = (a / "transcriptome.fq").read_bytes()
    assert got == (b / "transcriptome.fq").read_bytes()
    want = gzip.open(os.path.join(GOLDEN, "toyset_rna.transcriptome.fq.gz"), "rt").read().split("\n")
    mine = got.decode().split("\n")
    assert len(mine) == len(want) == 175 * 4 + 1
    for i in range(0, 175 * 4, 4):
        assert mine[i + 1] == want[i + 1]
        assert mine[i].split()[0] == want[i].split()[0]
        assert re.search(r"total_reads=(\d+)", mine[i]).group(1) == re.search(r"total_reads=(\d+)", want[i]).group(1)


def test_polish_cli_merges_similar_consensi(built, tmp_path):
    """cDNA-mode polish on consensi that DO cluster (near-duplicate sequences, some reverse-complemented)."""
    seqs, quals, _, _ = synth.reads(40, 3, 1, True, seed=23, sub=0.01, ins=0.005, dele=0.005)
    recs = b"".join(b"@gene_cluster_%d reads=%d labels=\n%s\n+\n%s\n" % (i, 6 + i, s, b"K" * len(s)) for i, s in enumerate(seqs))
    (tmp_path / "c.fq").write_bytes(recs)
    a = tmp_path / "a"; b = tmp_path / "b"
    a.mkdir(); b.mkdir()
    subprocess.run([RATTLE, "polish", "-i", str(tmp_path / "c.fq"), "-o", str(a)], check=True, capture_output=True)
    subprocess.run([ORACLE, "polish", "-i", str(tmp_path / "c.fq"), "-o", str(b)], check=True, capture_output=True)
    got = (a / "transcriptome.fq").read_bytes()
    assert got == (b / "transcriptome.fq").read_bytes()
    assert 1 <= got.count(b"@cluster_") < 40


def test_cli_one_job_over_several_ranks(built, tmp_path):
    """`--devices a,b,...`: one host thread per rank, the library shards candidates / gene clusters / packs.  Three ranks on
    the box's one GPU through the in-process host exchange (RCCL wants one GPU per rank): every output file identical to
    the single-device run."""
    seqs, quals, _, _ = synth.reads(1500, 9, 2, True, seed=19)
    fq = tmp_path / "in.fastq"
    fq.write_bytes(synth.fastq_text(seqs, quals))
    a = tmp_path / "one"; b = tmp_path / "three"
    a.mkdir(); b.mkdir()
    for out, extra in ((a, []), (b, ["--devices", "0,0,0", "--host-exchange"])):
        subprocess.run([RATTLE, "cluster", "-i", str(fq), "-o", str(out), "--iso"] + extra, check=True, capture_output=True)
        subprocess.run([RATTLE, "correct", "-i", str(fq), "-c", str(out / "clusters.out"), "-o", str(out), "-s", "40"] + extra, check=True, capture_output=True)
    for f in ("clusters.out", "corrected.fq", "uncorrected.fq", "consensi.fq"):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f
    assert (a / "corrected.fq").stat().st_size > 100000


@pytest.mark.parametrize("batch", ["1", "7", "64"])
def test_clusters_do_not_depend_on_the_seed_batch(built, tmp_path, batch):
    """The greedy driver evaluates the next B un-clustered seeds together and resolves them in index order; B (RATTLE_SEED_BATCH,
    default 512, adapted per round) must not show in the result.  B = 1 is the reference's own seed-at-a-time loop."""
    seqs, quals, _, _ = synth.reads(1200, 8, 3, True, seed=23)
    fq = tmp_path / "in.fastq"
    fq.write_bytes(synth.fastq_text(seqs, quals))
    a = tmp_path / "default"; b = tmp_path / "small"
    a.mkdir(); b.mkdir()
    for out, env in ((a, {}), (b, {"RATTLE_SEED_BATCH": batch})):
        subprocess.run([RATTLE, "cluster", "-i", str(fq), "-o", str(out), "--iso"], check=True, capture_output=True, env=dict(os.environ, **env))
    assert (a / "clusters.out").read_bytes() == (b / "clusters.out").read_bytes()
    assert (a / "clusters.out").stat().st_size > 1000


@pytest.mark.parametrize("k", ["10", "12"])
def test_both_count_passes_give_the_same_clusters(built, tmp_path, k):
    """Kernel B's count pass exists twice (per pair with binary searches; per seed with an LDS bit set, which for k > 10
    folds the hash and returns an upper bound of |common|): the exact rejection built on either must lead to the same
    clusters at both levels of `--iso`."""
    seqs, quals, _, _ = synth.reads(1500, 6, 3, True, seed=29)
    fq = tmp_path / "in.fastq"
    fq.write_bytes(synth.fastq_text(seqs, quals))
    outs = {}
    for mode in ("seed", "search"):
        d = tmp_path / mode
        d.mkdir()
        subprocess.run([RATTLE, "cluster", "-i", str(fq), "-o", str(d), "--iso", "-k", k, "--iso-kmer-size", str(int(k) + 1)], check=True, capture_output=True,
                       env=dict(os.environ, RATTLE_PAIR_COUNT=mode))
        outs[mode] = (d / "clusters.out").read_bytes()
    assert outs["seed"] == outs["search"] and len(outs["seed"]) > 1000


def test_labels_and_several_input_files_match_oracle_cli(built, tmp_path):
    """`-i a.fq,b.fq -l s1,s2` (main.cpp:16-112: one running record index over the files, "," + label appended to every
    header) through cluster and correct: the consensus headers carry `labels=s1:<n>,s2:<n>,` counted from the members'
    headers (correct.cpp:453-469,495-515); every file byte-identical to the oracle CLI."""
    seqs, quals, _, _ = synth.reads(360, 3, 1, True, seed=31)
    (tmp_path / "a.fq").write_bytes(synth.fastq_text(seqs[:200], quals[:200]))
    (tmp_path / "b.fastq").write_bytes(synth.fastq_text(seqs[200:], quals[200:]))
    inp = f"{tmp_path / 'a.fq'},{tmp_path / 'b.fastq'}"
    a = tmp_path / "a"; b = tmp_path / "b"
    a.mkdir(); b.mkdir()
    for tool, out in ((RATTLE, a), (ORACLE, b)):
        subprocess.run([tool, "cluster", "-i", inp, "-l", "s1,s2", "-o", str(out)], check=True, capture_output=True)
    assert (a / "clusters.out").read_bytes() == (b / "clusters.out").read_bytes()
    for tool, out in ((RATTLE, a), (ORACLE, b)):
        subprocess.run([tool, "correct", "-i", inp, "-l", "s1,s2", "-c", str(out / "clusters.out"), "-o", str(out), "-s", "50"], check=True, capture_output=True)
    for f in ("corrected.fq", "uncorrected.fq", "consensi.fq"):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f
    heads = (a / "consensi.fq").read_bytes().split(b"\n")[0::4]
    assert heads[0].startswith(b"@gene_cluster_0 reads=") and b" labels=s1:" in heads[0] and b",s2:" in heads[0]
    import re
    n1 = sum(int(re.search(rb"s1:(\d+),", h).group(1)) for h in heads if h)
    n2 = sum(int(re.search(rb"s2:(\d+),", h).group(1)) for h in heads if h)
    assert 0 < n1 <= 200 and 0 < n2 <= 160
    txt = (a / "corrected.fq").read_bytes()
    assert b",s1,gene_cluster_" in txt and b",s2,gene_cluster_" in txt


def test_fasta_input_matches_oracle_cli(built, tmp_path):
    """FASTA input (fasta.cpp:33-205): multi-line records, lower-case bases upper-cased (:131), a record with N skipped, a
    short one filtered; `correct` gives every base quality '~'.  Also a mix of one FASTA and one FASTQ file."""
    seqs, quals, _, _ = synth.reads(260, 3, 1, True, seed=37)
    lines = []
    for i, s in enumerate(seqs[:160]):
        if i == 4:
            s = s[:30] + b"N" + s[31:]
        if i == 7:
            s = s[:90]
        if i % 3 == 0:
            s = s.lower()
        lines.append(b">f%d some text\n" % i)
        lines += [s[p:p + 70] + b"\n" for p in range(0, len(s), 70)]
    (tmp_path / "x.fasta").write_bytes(b"".join(lines))
    (tmp_path / "y.fq").write_bytes(synth.fastq_text(seqs[160:], quals[160:]))
    for inp, tag in ((str(tmp_path / "x.fasta"), "one"), (f"{tmp_path / 'x.fasta'},{tmp_path / 'y.fq'}", "two")):
        a = tmp_path / ("a" + tag); b = tmp_path / ("b" + tag)
        a.mkdir(); b.mkdir()
        for tool, out in ((RATTLE, a), (ORACLE, b)):
            subprocess.run([tool, "cluster", "-i", inp, "-o", str(out)], check=True, capture_output=True)
        assert (a / "clusters.out").read_bytes() == (b / "clusters.out").read_bytes(), tag
        for tool, out in ((RATTLE, a), (ORACLE, b)):
            subprocess.run([tool, "correct", "-i", inp, "-c", str(out / "clusters.out"), "-o", str(out), "-s", "60"], check=True, capture_output=True)
        for f in ("corrected.fq", "uncorrected.fq", "consensi.fq"):
            assert (a / f).read_bytes() == (b / f).read_bytes(), (tag, f)
        cor = (a / "corrected.fq").read_bytes().split(b"\n")
        assert cor[0].startswith(b">f") and len(cor) > 400               # FASTA headers keep their '>' (fasta.cpp:47)


def test_mixed_length_flow_matches_oracle_cli(built, tmp_path):
    """BASELINE configs[4] in miniature, exact: --rna reads from 200 nt to 14 kb (every column class of kernel C up to the
    segmented rows beyond 8192 columns) through cluster -> correct -> polish, every file byte-identical to the oracle CLI."""
    import numpy as np
    rng = np.random.default_rng(41)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    lens = [220, 700, 1300, 1900, 2400, 3300, 5200, 7300, 9800, 14000]
    depth = [14, 12, 10, 9, 8, 7, 6, 6, 6, 6]
    seqs, quals = [], []
    for L, d in zip(lens, depth):
        tx = acgt[rng.integers(0, 4, L)]
        for _ in range(d):
            r = rng.random(L)
            s = tx.copy()
            sub = (r >= 0.02) & (r < 0.05)
            s[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
            s = s[r >= 0.02]
            pos = np.sort(rng.integers(0, len(s) + 1, int(0.02 * len(s))))
            s = np.insert(s, pos, acgt[rng.integers(0, 4, len(pos))])
            s = s[int(rng.integers(0, max(1, L // 20))):]
            seqs.append(s.tobytes())
            quals.append(bytes(rng.integers(36, 74, len(s)).astype(np.uint8)))
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]; quals = [quals[i] for i in order]
    fq = tmp_path / "mixed.fastq"
    fq.write_bytes(synth.fastq_text(seqs, quals))
    a = tmp_path / "a"; b = tmp_path / "b"
    a.mkdir(); b.mkdir()
    for tool, out in ((RATTLE, a), (ORACLE, b)):
        subprocess.run([tool, "cluster", "-i", str(fq), "-o", str(out), "--rna"], check=True, capture_output=True)
    assert (a / "clusters.out").read_bytes() == (b / "clusters.out").read_bytes()
    assert len(hps.decode((a / "clusters.out").read_bytes(), fields=3)) >= 10
    for tool, out in ((RATTLE, a), (ORACLE, b)):
        subprocess.run([tool, "correct", "-i", str(fq), "-c", str(out / "clusters.out"), "-o", str(out)], check=True, capture_output=True)
    for f in ("corrected.fq", "uncorrected.fq", "consensi.fq"):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f
    assert max(len(l) for l in (a / "consensi.fq").read_bytes().split(b"\n")[1::4]) > 13000
    for tool, out in ((RATTLE, a), (ORACLE, b)):
        subprocess.run([tool, "polish", "-i", str(out / "consensi.fq"), "-o", str(out), "--rna"], check=True, capture_output=True)
    assert (a / "transcriptome.fq").read_bytes() == (b / "transcriptome.fq").read_bytes()
