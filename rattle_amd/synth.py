"""Seeded synthetic cDNA / direct-RNA read generator.

Transcriptome: G genes x I isoforms; a gene is 8 exons of U[80,300] nt random ACGT; an isoform
keeps the first and last exon and each internal exon with p=0.7 (distinct isoforms only).
Abundance: Zipf(s=1) over transcripts.  Reads: full length with 5' truncation U[0,10%];
per-base errors 4% substitution / 3% insertion / 3% deletion; qualities Phred ~ N(10,3)
clipped to [3,40]; cDNA mode reverse-complements a read with p=0.5.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


def transcriptome(genes: int, isoforms: int, seed: int = 20260928, exon=(80, 300)):
    rng = np.random.default_rng(seed)
    tx, tx_gene = [], []
    for g in range(genes):
        exons = [_ACGT[rng.integers(0, 4, rng.integers(exon[0], exon[1] + 1))] for _ in range(8)]
        seen = set()
        tries = 0
        while len(seen) < isoforms and tries < 100:
            tries += 1
            keep = [True] + [bool(rng.random() < 0.7) for _ in range(6)] + [True]
            if isoforms == 1:
                keep = [True] * 8
            key = tuple(keep)
            if key in seen:
                continue
            seen.add(key)
            tx.append(np.concatenate([e for e, k in zip(exons, keep) if k]))
            tx_gene.append(g)
    return tx, np.array(tx_gene)


def reads(n: int, genes: int, isoforms: int = 1, both_strands: bool = True, seed: int = 20260929,
          tx_seed: int = 20260928, sub=0.04, ins=0.03, dele=0.03, exon=(80, 300)):
    """Returns (seqs: list[bytes], quals: list[bytes], tx_id: ndarray, strand: ndarray)."""
    tx, _ = transcriptome(genes, isoforms, tx_seed, exon)
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, len(tx) + 1)
    perm = rng.permutation(len(tx))
    p = np.zeros(len(tx))
    p[perm] = w / w.sum()
    tid = rng.choice(len(tx), size=n, p=p)
    flip = rng.random(n) < 0.5 if both_strands else np.zeros(n, bool)
    seqs, quals = [], []
    for i in range(n):
        t = tx[tid[i]]
        cut = int(rng.integers(0, len(t) // 10 + 1))
        t = t[cut:]
        L = len(t)
        r = rng.random(L)
        is_del = r < dele
        is_sub = (r >= dele) & (r < dele + sub)
        out = t.copy()
        out[is_sub] = _ACGT[(np.searchsorted(_ACGT, out[is_sub]) + rng.integers(1, 4, int(is_sub.sum()))) % 4]
        n_ins = rng.random(L) < ins
        pieces = np.empty(L + int(n_ins.sum()), np.uint8)
        keep = ~is_del
        # build with insertions after position j
        idx = np.cumsum(np.where(keep, 1, 0) + np.where(n_ins, 1, 0))
        total = int(idx[-1]) if L else 0
        pieces = np.empty(total, np.uint8)
        start = idx - (np.where(keep, 1, 0) + np.where(n_ins, 1, 0))
        pieces[start[keep]] = out[keep]
        ins_pos = start[n_ins] + np.where(keep[n_ins], 1, 0)
        pieces[ins_pos] = _ACGT[rng.integers(0, 4, int(n_ins.sum()))]
        if flip[i]:
            pieces = _COMP[pieces[::-1]]
        q = np.clip(np.rint(rng.normal(10, 3, len(pieces))), 3, 40).astype(np.uint8) + 33
        seqs.append(pieces.tobytes())
        quals.append(q.tobytes())
    return seqs, quals, tid, flip.astype(np.uint8)


def fastq_text(seqs, quals, prefix="r") -> bytes:
    out = []
    for i, (s, q) in enumerate(zip(seqs, quals)):
        out.append(b"@%s%d\n%s\n+\n%s\n" % (prefix.encode(), i, s, q))
    return b"".join(out)
