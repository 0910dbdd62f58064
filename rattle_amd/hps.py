"""clusters.out codec (hps stream) -- Python mirror used by tests and tools.

The reference writes `clusters.out` with `hps::to_stream(cluster_set_t, ...)`
(/root/reference/main.cpp:275,322) using the `serialize` methods of
/root/reference/cluster.hpp:15-18,30-33.  hps (jl2922/hps) is an un-vendored,
unpinned submodule; the grammar below was recovered empirically from the two
shipped fixtures:

    file    := uvarint(n_clusters) cluster*
    cluster := cseq(main_seq) uvarint(n_seqs) cseq*
    cseq    := svarint(seq_id) u8(rev) svarint(gene_id)      # current, 3-field
    cseq    := svarint(seq_id) u8(rev)                       # old, 2-field

The product codec is the C++ one in rattle_amd/csrc/rattle_main.cpp; this file is
the independent restatement the tests compare it with.
"""
from __future__ import annotations
from typing import List, Tuple

Cseq = Tuple[int, int, int]            # (seq_id, rev, gene_id)
Cluster = Tuple[Cseq, List[Cseq]]      # (main_seq, seqs)


def _uvarint(buf: bytes, p: int) -> Tuple[int, int]:
    x = 0
    s = 0
    while True:
        b = buf[p]
        p += 1
        x |= (b & 0x7F) << s
        if not b & 0x80:
            return x, p
        s += 7


def _svarint(buf: bytes, p: int) -> Tuple[int, int]:
    z, p = _uvarint(buf, p)
    return (z >> 1) ^ -(z & 1), p


def _put_uvarint(out: bytearray, x: int) -> None:
    while x >= 0x80:
        out.append((x & 0x7F) | 0x80)
        x >>= 7
    out.append(x)


def _put_svarint(out: bytearray, x: int) -> None:
    _put_uvarint(out, ((x << 1) ^ (x >> 31)) & 0xFFFFFFFF)


def decode(buf: bytes, fields: int = 3) -> List[Cluster]:
    """Decode a clusters.out stream; fields=2 for the old record layout."""
    p = 0

    def cseq(p):
        sid, p = _svarint(buf, p)
        rev = buf[p]
        p += 1
        if rev not in (0, 1):
            raise ValueError("bad rev byte")
        gid = -1
        if fields == 3:
            gid, p = _svarint(buf, p)
        return (sid, rev, gid), p

    n, p = _uvarint(buf, p)
    res = []
    for _ in range(n):
        main, p = cseq(p)
        m, p = _uvarint(buf, p)
        seqs = []
        for _ in range(m):
            c, p = cseq(p)
            seqs.append(c)
        res.append((main, seqs))
    if p != len(buf):
        raise ValueError("trailing bytes")
    return res


def decode_auto(buf: bytes) -> Tuple[List[Cluster], int]:
    """Try the current 3-field layout, fall back to the old 2-field one."""
    for f in (3, 2):
        try:
            return decode(buf, f), f
        except (ValueError, IndexError):
            continue
    raise ValueError("not a clusters.out stream")


def encode(clusters: List[Cluster], fields: int = 3) -> bytes:
    out = bytearray()
    _put_uvarint(out, len(clusters))

    def cseq(c):
        _put_svarint(out, c[0])
        out.append(1 if c[1] else 0)
        if fields == 3:
            _put_svarint(out, c[2])

    for main, seqs in clusters:
        cseq(main)
        _put_uvarint(out, len(seqs))
        for c in seqs:
            cseq(c)
    return bytes(out)
