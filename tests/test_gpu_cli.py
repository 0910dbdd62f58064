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
