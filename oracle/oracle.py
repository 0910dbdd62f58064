"""ctypes loader for the ORACLE libraries (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
liboracle.so = the CPU restatement (oracle/orc_*.hpp).  _ref/libref.so = the reference's own
kmer.cpp/similarity.cpp/utils.cpp/fasta.cpp compiled in place (authoring container only; the
prebuilt .so travels to the GPU box).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_P = C.POINTER


def build(ref: bool = True):
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _ptr(a, t):
    return a.ctypes.data_as(_P(t))


class _Lib:
    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        f = getattr(self.lib, prefix + "_var")
        f.restype = C.c_double

    def extract_kmers(self, seq: bytes, k: int, both: bool):
        nk = max(len(seq) - k, 0)
        fh = np.zeros(nk, np.uint32); fp = np.zeros(nk, np.int32)
        rh = np.zeros(nk, np.uint32); rp = np.zeros(nk, np.int32)
        bf = np.zeros(64, np.uint64); br = np.zeros(64, np.uint64)
        getattr(self.lib, self.prefix + "_extract_kmers")(
            C.c_char_p(seq), C.c_uint32(len(seq)), C.c_int(k), C.c_int(int(both)), _ptr(fh, C.c_uint32), _ptr(fp, C.c_int32),
            _ptr(rh, C.c_uint32), _ptr(rp, C.c_int32), _ptr(bf, C.c_uint64), _ptr(br, C.c_uint64))
        return fh, fp, rh, rp, bf, br

    def pair_score(self, a: bytes, b: bytes, k: int, strand: int, dist_cap: int = 1 << 16):
        bases = C.c_int32(); hc = C.c_int32(); nd = C.c_int32(); nm = C.c_int32(); var = C.c_double()
        dist = np.zeros(dist_cap, np.int32)
        getattr(self.lib, self.prefix + "_pair_score")(
            C.c_char_p(a), C.c_uint32(len(a)), C.c_char_p(b), C.c_uint32(len(b)), C.c_int(k), C.c_int(strand),
            C.byref(bases), C.byref(hc), C.byref(nd), C.byref(var), C.byref(nm), _ptr(dist, C.c_int32), C.c_int32(dist_cap))
        return bases.value, hc.value, nd.value, var.value, nm.value, dist[:min(nd.value, dist_cap)].copy()

    def var(self, v):
        a = np.ascontiguousarray(v, np.int32)
        return getattr(self.lib, self.prefix + "_var")(_ptr(a, C.c_int32), C.c_uint32(len(a)))


class Oracle(_Lib):
    def __init__(self):
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        super().__init__(path, "orc")
        self.lib.orc_poa_msa.restype = C.c_int64

    def cluster_reads(self, seqs, k=10, t_s=0.2, t_v=1000000.0, bvB=0.4, bvb=0.2, bvf=0.05, repr_pct=0.15, is_rna=False):
        """cluster_reads on reads already in processing order; returns (hps-style list, counters)."""
        n = len(seqs)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        cat = b"".join(seqs)
        mid = np.zeros(n, np.int32); mrev = np.zeros(n, np.uint8); co = np.zeros(n + 1, np.uint32)
        mem = np.zeros(n, np.int32); memr = np.zeros(n, np.uint8); cnt = np.zeros(3, np.uint64)
        nc = self.lib.orc_cluster_reads(C.c_char_p(cat), _ptr(off, C.c_uint64), C.c_uint32(n), C.c_int(k), C.c_double(t_s),
                                        C.c_double(t_v), C.c_double(bvB), C.c_double(bvb), C.c_double(bvf),
                                        C.c_double(repr_pct), C.c_int(int(is_rna)), _ptr(mid, C.c_int32), _ptr(mrev, C.c_uint8),
                                        _ptr(co, C.c_uint32), _ptr(mem, C.c_int32), _ptr(memr, C.c_uint8), _ptr(cnt, C.c_uint64))
        out = []
        for c in range(nc):
            out.append(((int(mid[c]), int(mrev[c]), -1),
                        [(int(mem[i]), int(memr[i]), -1) for i in range(int(co[c]), int(co[c + 1]))]))
        return out, cnt

    def poa_msa(self, seqs):
        n = len(seqs)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        cat = b"".join(seqs)
        cap = 64 * (int(off[-1]) + 1024) + n * 1024
        while True:
            buf = C.create_string_buffer(cap)
            cells = C.c_uint64()
            w = self.lib.orc_poa_msa(C.c_char_p(cat), _ptr(off, C.c_uint64), C.c_uint32(n), buf, C.c_uint64(cap), C.byref(cells))
            if w >= 0:
                break
            cap *= 4
        raw = buf.raw
        return [raw[i * w:(i + 1) * w] for i in range(n)], cells.value

    def set_cv_order(self, order: bytes):
        assert len(order) == 6
        self.lib.orc_set_cv_order(C.c_char_p(order))

    def correct(self, headers, seqs, quals, clusters_hps: bytes, min_occ=0.3, gap_occ=0.3, split=200, min_reads=5, pack_order=None):
        """Whole `correct`; returns (corrected, uncorrected, consensi) FASTQ texts + counters.
        pack_order = {cluster: [pack indices]}: the order a cluster's pack consensi enter POA #3 in."""
        n = len(seqs)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        cat = b"".join(seqs); qcat = b"".join(quals)
        H = (C.c_char_p * n)(*headers)
        a = C.c_void_p(); b = C.c_void_p(); c = C.c_void_p(); cnt = np.zeros(3, np.uint64)
        buf = (C.c_uint8 * len(clusters_hps)).from_buffer_copy(clusters_hps)
        po = pack_order or {}
        pcl = np.array(sorted(po), np.uint32)
        poff = np.zeros(len(pcl) + 1, np.uint32)
        poff[1:] = np.cumsum([len(po[int(x)]) for x in pcl])
        pperm = np.array([y for x in pcl for y in po[int(x)]] + [0], np.uint32)
        rc = self.lib.orc_correct(C.c_char_p(cat), C.c_char_p(qcat), _ptr(off, C.c_uint64), C.c_uint32(n), H, buf,
                                  C.c_uint64(len(clusters_hps)), C.c_double(min_occ), C.c_double(gap_occ), C.c_int(split),
                                  C.c_int(min_reads), C.byref(a), C.byref(b), C.byref(c), _ptr(cnt, C.c_uint64),
                                  C.c_uint32(len(pcl)), _ptr(pcl, C.c_uint32), _ptr(poff, C.c_uint32), _ptr(pperm, C.c_uint32))
        if rc != 0:
            raise RuntimeError("orc_correct failed")
        out = tuple(C.string_at(x) for x in (a, b, c))
        for x in (a, b, c):
            self.lib.orc_free(x)
        return out + (cnt,)


    def set_poa_simd(self, on: bool) -> bool:
        """POA matrices filled by the AVX2 int16 row kernel (what spoa's SIMD engine does) instead of the scalar loops;
        returns whether this CPU has AVX2 (without it the scalar fill keeps running)."""
        return bool(self.lib.orc_set_poa_simd(C.c_int(int(on))))

    def fix_msa_ends(self, rows, seqs, quals):
        """correct.cpp:32-92 on hand-built rows; returns (rows, seqs, quals) as the reference leaves them."""
        n, width = len(rows), len(rows[0])
        assert all(len(r) == width for r in rows)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        rb = C.create_string_buffer(b"".join(rows), n * width)
        sb = C.create_string_buffer(b"".join(seqs) + b"\0"); qb = C.create_string_buffer(b"".join(quals) + b"\0")
        ln = np.zeros(n, np.uint32)
        self.lib.orc_fix_msa_ends(rb, C.c_uint32(n), C.c_uint32(width), sb, qb, _ptr(off, C.c_uint64), _ptr(ln, C.c_uint32))
        out_rows = [rb.raw[i * width:(i + 1) * width] for i in range(n)]
        out_s = [sb.raw[int(off[i]):int(off[i]) + int(ln[i])] for i in range(n)]
        out_q = [qb.raw[int(off[i]):int(off[i]) + int(ln[i])] for i in range(n)]
        return out_rows, out_s, out_q


class Ref(_Lib):
    """The real reference TUs (oracle/_ref/libref.so), when built."""

    def __init__(self):
        super().__init__(ref_path(), "ref")


def ref_path():
    return os.path.join(_HERE, "_ref", "libref.so")


def have_ref():
    return os.path.exists(ref_path())
