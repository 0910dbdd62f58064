"""The host-only CLI modes (`cluster_summary`, `extract_clusters`; no GPU involved) against the
fixtures the reference ships in toyset/rna/output/."""
import gzip
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

RATTLE = os.path.join(ROOT, "rattle_amd", "csrc", "rattle")


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    if not os.path.exists(RATTLE):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.dirname(RATTLE)])
    d = tmp_path_factory.mktemp("toy")
    (d / "sample.fastq").write_bytes(gzip.open(os.path.join(GOLDEN, "toyset_rna.fastq.gz")).read())
    return d


def test_cluster_summary_matches_fixture(sample):
    out = subprocess.run([RATTLE, "cluster_summary", "-i", str(sample / "sample.fastq"), "-c",
                          os.path.join(GOLDEN, "toyset_rna.clusters.out")], capture_output=True, text=True, check=True).stdout
    got = out.split("\n")[:-1]
    want = gzip.open(os.path.join(GOLDEN, "toyset_rna.cluster_summary.tsv.gz"), "rt").read().split("\n")[:-1]
    assert len(got) == len(want) == 8306
    # the fixture is the older two-column layout `header,<cid>`; the current source writes `header,gene_cluster_<cid>`
    assert [g.replace(",gene_cluster_", ",") for g in got] == want


def test_extract_clusters_matches_fixture(sample, tmp_path):
    subprocess.run([RATTLE, "extract_clusters", "-i", str(sample / "sample.fastq"), "-c", os.path.join(GOLDEN, "toyset_rna.clusters.out"),
                    "-o", str(tmp_path), "--fastq", "-m", "0"], check=True, capture_output=True)
    assert len(list(tmp_path.glob("cluster_*.fq"))) == 546
    for cid in (0, 7, 545):
        assert (tmp_path / f"cluster_{cid}.fq").read_bytes() == open(os.path.join(GOLDEN, f"toyset_rna.cluster_{cid}.fq"), "rb").read()
    fa = tmp_path / "fa"
    fa.mkdir()
    subprocess.run([RATTLE, "extract_clusters", "-i", str(sample / "sample.fastq"), "-c", os.path.join(GOLDEN, "toyset_rna.clusters.out"),
                    "-o", str(fa), "-m", "5"], check=True, capture_output=True)
    assert len(list(fa.glob("cluster_*.fa"))) == 175
    first = (fa / "cluster_1.fa").read_text().split("\n")
    assert first[0].startswith("@ERR") and set(first[1]) <= set("ACGT") and len(first) == 14 * 2 + 1
