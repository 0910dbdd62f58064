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


def mixed_transcriptome(n_tx: int, seed: int = 20260928, lo: int = 150, hi: int = 100000, body_hi: int = 8000, tail_frac: float = 0.01):
    """Transcripts for BASELINE configs[4]: lengths log-uniform on [lo, hi] "truncated so that
    the mean is ~2 kb": a body log-uniform on [lo, body_hi] (mean (body_hi - lo) / ln(body_hi / lo) ~ 2 kb for 150..8000)
    plus a thin tail (tail_frac of the transcripts) log-uniform on [body_hi, hi].  Random ACGT, no exon structure."""
    rng = np.random.default_rng(seed)
    tail = rng.random(n_tx) < tail_frac
    u = rng.random(n_tx)
    lens = np.where(tail, np.exp(np.log(body_hi) + u * (np.log(hi) - np.log(body_hi))), np.exp(np.log(lo) + u * (np.log(body_hi) - np.log(lo))))
    lens = np.clip(np.rint(lens), lo, hi).astype(np.int64)
    return [_ACGT[rng.integers(0, 4, int(l))] for l in lens]


def reads_packed(n: int, genes: int, isoforms: int = 1, both_strands: bool = True, seed: int = 20260929,
                 tx_seed: int = 20260928, sub=0.04, ins=0.03, dele=0.03, exon=(80, 300), chunk: int = 500, tx=None):
    """Vectorised variant of reads() for large n: same model, returns packed arrays
    (seq uint8, qual uint8, offsets uint64[n+1], tx_id, strand).  Different random stream than
    reads() (chunked numpy draws), same distributions.  tx: explicit transcript list (e.g. mixed_transcriptome)."""
    if tx is None:
        tx, _ = transcriptome(genes, isoforms, tx_seed, exon)
    txlen = np.array([len(t) for t in tx], np.int64)
    txoff = np.zeros(len(tx) + 1, np.int64)
    txoff[1:] = np.cumsum(txlen)
    txcat = np.concatenate(tx)
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, len(tx) + 1)
    perm = rng.permutation(len(tx))
    p = np.zeros(len(tx))
    p[perm] = w / w.sum()
    tid = rng.choice(len(tx), size=n, p=p)
    flip = (rng.random(n) < 0.5) if both_strands else np.zeros(n, bool)
    seq_parts, qual_parts, lens_all = [], [], []
    for c0 in range(0, n, chunk):
        t = tid[c0:c0 + chunk]
        m = len(t)
        cut = (rng.random(m) * (txlen[t] // 10 + 1)).astype(np.int64)
        L = txlen[t] - cut
        T = int(L.sum())
        rstart = np.zeros(m + 1, np.int64)
        rstart[1:] = np.cumsum(L)
        rid = np.repeat(np.arange(m), L)
        within = np.arange(T) - rstart[rid]
        base = txcat[txoff[t][rid] + cut[rid] + within]
        r = rng.random(T)
        is_del = r < dele
        is_sub = (r >= dele) & (r < dele + sub)
        idx = np.searchsorted(_ACGT, base)
        shift = rng.integers(1, 4, T)
        base = np.where(is_sub, _ACGT[(idx + shift) % 4], base)
        has_ins = rng.random(T) < ins
        ins_base = _ACGT[rng.integers(0, 4, T)]
        keep = ~is_del
        cnt = keep.astype(np.int64) + has_ins.astype(np.int64)
        end = np.cumsum(cnt)
        start = end - cnt
        total = int(end[-1]) if T else 0
        out = np.empty(total, np.uint8)
        out[start[keep]] = base[keep]
        out[(start + keep)[has_ins]] = ins_base[has_ins]
        ostart = np.zeros(m + 1, np.int64)
        ostart[1:] = end[rstart[1:] - 1]
        olen = ostart[1:] - ostart[:-1]
        # reverse-complement flipped reads in place via an index map
        f = flip[c0:c0 + chunk]
        orid = np.repeat(np.arange(m), olen)
        owithin = np.arange(total) - ostart[orid]
        src = np.where(f[orid], ostart[orid] + olen[orid] - 1 - owithin, ostart[orid] + owithin)
        o2 = out[src]
        o2 = np.where(f[orid], _COMP[o2], o2)
        q = np.clip(np.rint(rng.normal(10, 3, total)), 3, 40).astype(np.uint8) + 33
        seq_parts.append(o2)
        qual_parts.append(q)
        lens_all.append(olen)
    lens = np.concatenate(lens_all)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens).astype(np.uint64)
    return np.concatenate(seq_parts), np.concatenate(qual_parts), off, tid, flip.astype(np.uint8)
