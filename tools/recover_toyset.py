#!/usr/bin/env python3
"""Rebuild the RNA toyset input and copy its expected outputs into tests/golden/.

Runs ONLY in the authoring container (needs /root/reference).  The reference
ships `toyset/rna/output/` but not `toyset/rna/input/sample.fastq`
(/root/reference/.MISSING_LARGE_BLOBS:3).  Every read of that input is present
in `output/clusters/cluster_<cid>.fq` (written by `extract_clusters --fastq`,
/root/reference/main.cpp:575-589) in `c.seqs` order, and `output/clusters.out`
gives each record's seq_id, so the input is recovered exactly.

Outputs (data only: inputs and expected outputs, no reference source text):
  tests/golden/toyset_rna.fastq.gz        recovered input, records in seq_id order
  tests/golden/toyset_rna.clusters.out    expected `cluster --rna` result (old 2-field hps)
  tests/golden/toyset_rna.consensi.fq.gz  expected `correct` consensi
  tests/golden/toyset_rna.uncorrected.ids expected headers of uncorrected.fq
  tests/golden/toyset_rna.transcriptome.fq.gz expected `polish` result
  tests/golden/toyset_rna.cluster_summary.tsv.gz  expected `cluster_summary` output (old 2-column layout)
  tests/golden/toyset_rna.cluster_{0,7,545}.fq    expected `extract_clusters --fastq` files (three samples)
  tests/golden/toyset_iso.clusters.out    3-field format KAT (cluster_benchmark)
  tests/golden/toyset_iso.summary.tsv.gz
"""
import gzip
import os
import shutil
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from rattle_amd import hps  # noqa: E402

REF = "/root/reference/toyset"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")


def gz_write(path, data: bytes):
    with open(path, "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, compresslevel=9) as f:
            f.write(data)


def main():
    os.makedirs(OUT, exist_ok=True)
    buf = open(f"{REF}/rna/output/clusters.out", "rb").read()
    clusters = hps.decode(buf, fields=2)
    n = sum(len(s) for _, s in clusters)
    recs = [None] * n
    for cid, (_, seqs) in enumerate(clusters):
        lines = open(f"{REF}/rna/output/clusters/cluster_{cid}.fq").read().split("\n")
        if lines and lines[-1] == "":
            lines.pop()
        assert len(lines) == 4 * len(seqs), (cid, len(lines), len(seqs))
        for i, (sid, rev, _) in enumerate(seqs):
            assert rev == 0
            assert recs[sid] is None
            recs[sid] = lines[4 * i:4 * i + 4]
    assert all(r is not None for r in recs)
    lens = [len(r[1]) for r in recs]
    assert lens == sorted(lens, reverse=True), "ids are expected to be length-descending"
    text = "".join("\n".join(r) + "\n" for r in recs)
    gz_write(f"{OUT}/toyset_rna.fastq.gz", text.encode())
    shutil.copyfile(f"{REF}/rna/output/clusters.out", f"{OUT}/toyset_rna.clusters.out")
    gz_write(f"{OUT}/toyset_rna.consensi.fq.gz", open(f"{REF}/rna/output/consensi.fq", "rb").read())
    gz_write(f"{OUT}/toyset_rna.transcriptome.fq.gz", open(f"{REF}/rna/output/transcriptome.fq", "rb").read())
    unc = open(f"{REF}/rna/output/uncorrected.fq").read().split("\n")
    ids = [unc[i] for i in range(0, len(unc) - 1, 4)]
    open(f"{OUT}/toyset_rna.uncorrected.ids", "w").write("\n".join(ids) + "\n")
    gz_write(f"{OUT}/toyset_rna.cluster_summary.tsv.gz", open(f"{REF}/rna/output/cluster_summary.tsv", "rb").read())
    for cid in (0, 7, 545):
        shutil.copyfile(f"{REF}/rna/output/clusters/cluster_{cid}.fq", f"{OUT}/toyset_rna.cluster_{cid}.fq")
    shutil.copyfile(f"{REF}/cluster_benchmark/output/clusters.out", f"{OUT}/toyset_iso.clusters.out")
    gz_write(f"{OUT}/toyset_iso.summary.tsv.gz", open(f"{REF}/cluster_benchmark/output/summary.tsv", "rb").read())
    print(f"recovered {n} reads, {len(clusters)} clusters, lengths {min(lens)}..{max(lens)}")


if __name__ == "__main__":
    main()
