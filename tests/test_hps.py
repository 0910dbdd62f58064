"""clusters.out (hps) codec KATs on the two fixtures the reference ships."""
import gzip
import os

from conftest import GOLDEN
from rattle_amd import hps


def test_old_two_field_fixture_roundtrip():
    b = open(os.path.join(GOLDEN, "toyset_rna.clusters.out"), "rb").read()
    cs, fields = hps.decode_auto(b)
    assert fields == 2 and len(cs) == 546
    ids = sorted(s[0] for _, seqs in cs for s in seqs)
    assert ids == list(range(8306))
    assert hps.encode(cs, fields=2) == b
    assert b[:7] == bytes.fromhex("a2040000050000")


def test_current_three_field_fixture_roundtrip_and_gene_ids():
    b = open(os.path.join(GOLDEN, "toyset_iso.clusters.out"), "rb").read()
    cs, fields = hps.decode_auto(b)
    assert fields == 3 and len(cs) == 942 and sum(len(s) for _, s in cs) == 8036
    assert hps.encode(cs, fields=3) == b
    # every gene_id equals the gene_cluster_<g> column of the shipped summary.tsv
    rows = gzip.open(os.path.join(GOLDEN, "toyset_iso.summary.tsv.gz"), "rt").read().split("\n")
    genes = {}
    for r in rows:
        if not r:
            continue
        f = r.split(",")
        tc = [x for x in f if x.startswith("transcript_cluster_")]
        gc = [x for x in f if x.startswith("gene_cluster_")]
        if tc and gc:
            genes[int(tc[0].split("_")[-1])] = int(gc[0].split("_")[-1])
    if genes:
        for cid, (main, seqs) in enumerate(cs):
            if cid in genes:
                assert main[2] == genes[cid]
                assert all(s[2] == genes[cid] for s in seqs)


def test_hand_made_values():
    cs = [((-1, 0, -1), []), ((2 ** 31 - 1, 1, 7), [(0, 0, 7), (300, 1, 7), (-5, 0, -1)])]
    b = hps.encode(cs)
    assert hps.decode(b) == cs
    # svarint(-1) == 0x01 and uvarint(300) == ac 02
    assert hps.encode([((-1, 0, -1), [])]) == bytes([1, 0x01, 0, 0x01, 0])
    assert bytes([0xd8, 0x04]) in hps.encode([((300, 0, -1), [])])
