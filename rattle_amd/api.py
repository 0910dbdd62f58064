"""Host-side mirror of the reference interface over the C ABI.

`Context.cluster_reads` has the argument meaning of cluster_reads
(/root/reference/cluster.hpp:44); `cluster_command` is the `rattle cluster` flow of
/root/reference/main.cpp:245-323 (length sort, gene level, optional --iso second level,
translation to original record indices).  Python is used here because this image has the
reference's C++ toolchain but the test harness is pytest; the C++ host with the same
structure is rattle_amd/csrc/rattle_main.cpp.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import ClusterParams, ClusterSet, Correction, CorrectParams, MsaSet, check

K_KMER, K_FILTER, K_SCORE, K_POA, K_POST = 0, 1, 2, 3, 4


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def pack_reads(seqs: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    cat = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if len(seqs) else np.zeros(0, np.uint8)
    return cat, off


@dataclass
class Clusters:
    main_id: np.ndarray
    main_rev: np.ndarray
    offsets: np.ndarray
    member_id: np.ndarray
    member_rev: np.ndarray
    counters: np.ndarray

    def as_list(self):
        """[( (main_id, main_rev, -1), [(id, rev, -1), ...] ), ...] like rattle_amd.hps."""
        out = []
        for c in range(len(self.main_id)):
            a, b = int(self.offsets[c]), int(self.offsets[c + 1])
            out.append(((int(self.main_id[c]), int(self.main_rev[c]), -1),
                        [(int(self.member_id[i]), int(self.member_rev[i]), -1) for i in range(a, b)]))
        return out


def correct_params(min_occ=0.3, gap_occ=0.3, err_ratio=30.0, split=200, min_reads=5, n_threads=0, vote_order: bytes = b"",
                   pack_order=None, max_pack_cells=0):
    """rattle_correct_params; pack_order = {cluster: [pack indices in the order their consensi enter POA #3]}.
    Returns (struct, keep-alive list)."""
    P = CorrectParams(min_occ, gap_occ, err_ratio, split, min_reads, n_threads, vote_order)
    keep = []
    if pack_order:
        cl = np.array(sorted(pack_order), np.uint32)
        offs = np.zeros(len(cl) + 1, np.uint32)
        offs[1:] = np.cumsum([len(pack_order[int(c)]) for c in cl])
        perm = np.array([x for c in cl for x in pack_order[int(c)]], np.uint32)
        P.n_pack_orders = len(cl)
        P.pack_order_cluster = _ptr(cl, C.c_uint32); P.pack_order_offsets = _ptr(offs, C.c_uint32); P.pack_order_perm = _ptr(perm, C.c_uint32)
        keep = [cl, offs, perm]
    P.max_pack_cells = int(max_pack_cells)
    return P, keep


def unpack_correction(R) -> dict:
    """rattle_correction -> dict of record lists (read_id, cluster_id, n_reads, seq, qual), counters, skip list."""
    def unpack(S):
        n = S.n
        o = np.ctypeslib.as_array(S.off, (n + 1,)).copy()
        tot = int(o[n])
        sq = C.string_at(S.seq, tot); ql = C.string_at(S.qual, tot)
        rid = np.ctypeslib.as_array(S.read_id, (max(n, 1),))[:n]
        cid = np.ctypeslib.as_array(S.cluster_id, (max(n, 1),))[:n]
        nr = np.ctypeslib.as_array(S.n_reads, (max(n, 1),))[:n]
        return [(int(rid[i]), int(cid[i]), int(nr[i]), sq[int(o[i]):int(o[i + 1])], ql[int(o[i]):int(o[i + 1])])
                for i in range(n)]

    K = R.skipped
    skipped = []
    if K.n:
        ro = np.ctypeslib.as_array(K.read_off, (K.n + 1,))
        rid = np.ctypeslib.as_array(K.read_id, (max(int(ro[K.n]), 1),))
        for i in range(K.n):
            skipped.append({"cluster": int(K.cluster_id[i]), "pack": int(K.pack[i]), "stage": int(K.stage[i]),
                            "reads": [int(x) for x in rid[int(ro[i]):int(ro[i + 1])]]})
    return {"corrected": unpack(R.corrected), "uncorrected": unpack(R.uncorrected), "consensi": unpack(R.consensi),
            "counters": np.array(list(R.counters), dtype=np.uint64), "skipped": skipped}


def correction_digest(R) -> int:
    """CRC over every output array of a rattle_correction (ids, offsets, bases, qualities)."""
    import zlib
    crc = 0
    for S in (R.corrected, R.uncorrected, R.consensi):
        n = S.n
        o = np.ctypeslib.as_array(S.off, (n + 1,))
        tot = int(o[n])
        crc = zlib.crc32(o.tobytes(), crc)
        for arr in (S.read_id, S.cluster_id, S.n_reads):
            crc = zlib.crc32(np.ctypeslib.as_array(arr, (max(n, 1),))[:n].tobytes(), crc)
        crc = zlib.crc32(C.string_at(S.seq, tot), crc)
        crc = zlib.crc32(C.string_at(S.qual, tot), crc)
    return crc


class CorrectionHandle:
    """A library-owned rattle_correction kept alive by the caller (bench.py digests it outside the timed region)."""

    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def counts(self):
        if self.ptr is None:
            return (0, 0, 0, np.zeros(8, np.uint64))
        R = self.ptr.contents
        return (R.corrected.n, R.uncorrected.n, R.consensi.n, np.array(list(R.counters), dtype=np.uint64))

    def digest(self):
        return None if self.ptr is None else correction_digest(self.ptr.contents)

    def host_bytes(self):
        """bases + qualities held by the three read sets (what free() gives back to the OS)"""
        if self.ptr is None:
            return 0
        R = self.ptr.contents
        tot = 0
        for S in (R.corrected, R.uncorrected, R.consensi):
            if S.n:
                tot += 2 * int(np.ctypeslib.as_array(S.off, (S.n + 1,))[S.n])
        return tot

    def free(self):
        if self.ptr is not None:
            self.lib.rattle_hip_correction_free(self.ptr)
            self.ptr = None


class Context:
    """One HIP device + the device-resident read index (rattle_ctx)."""

    def __init__(self, device: Optional[int] = 0):
        """device=None: a host-only context (exchange entry points only, rattle_hip_ctx_create_host)."""
        self.lib = _lib.load()
        h = C.c_void_p()
        if device is None:
            check(self.lib.rattle_hip_ctx_create_host(C.byref(h)))
        else:
            check(self.lib.rattle_hip_ctx_create(device, C.byref(h)))
        self.h = h
        self.n = 0
        self.both = False
        self.k = 0

    def close(self):
        if self.h:
            self.lib.rattle_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # a3
    def load_reads(self, seqs: Sequence[bytes], k: int, both_strands: bool):
        cat, off = pack_reads(seqs)
        self.load_packed(cat, off, k, both_strands)

    def stage_reads(self, cat: np.ndarray, qcat: Optional[np.ndarray], off: np.ndarray):
        """Keep the reads (and qualities) resident in HBM; later calls given the same arrays skip the upload."""
        q = _ptr(qcat, C.c_uint8) if qcat is not None else None
        check(self.lib.rattle_hip_stage_reads(self.h, _ptr(cat, C.c_uint8), q, _ptr(off, C.c_uint64), len(off) - 1))

    def unstage_reads(self):
        check(self.lib.rattle_hip_unstage_reads(self.h))

    def load_packed(self, cat: np.ndarray, off: np.ndarray, k: int, both_strands: bool):
        n = len(off) - 1
        check(self.lib.rattle_hip_load_reads(self.h, _ptr(cat, C.c_uint8), _ptr(off, C.c_uint64), n, k, int(both_strands)))
        self.n, self.both, self.k = n, bool(both_strands), k
        self._lens = (off[1:] - off[:-1]).astype(np.int64)

    def read_index(self, r: int, strand: int):
        nk = max(int(self._lens[r]) - self.k, 0)
        h = np.zeros(nk, np.uint32)
        p = np.zeros(nk, np.int32)
        bv = np.zeros(64, np.uint64)
        pc = C.c_uint32()
        check(self.lib.rattle_hip_get_read_index(self.h, r, strand, _ptr(h, C.c_uint32), _ptr(p, C.c_int32),
                                                 _ptr(bv, C.c_uint64), C.byref(pc)))
        return h, p, bv, pc.value

    # a4
    def bv_filter(self, seed_ids, cand_ids, first_cand, lut, fwd_bypass: bool) -> np.ndarray:
        s = np.ascontiguousarray(seed_ids, np.uint32)
        c = np.ascontiguousarray(cand_ids, np.uint32)
        f = np.ascontiguousarray(first_cand, np.uint32)
        l = np.ascontiguousarray(lut, np.uint16)
        assert len(l) == 4097 and len(f) == len(s)
        out = np.zeros((len(s), len(c)), np.uint8)
        check(self.lib.rattle_hip_bv_filter(self.h, _ptr(s, C.c_uint32), len(s), _ptr(c, C.c_uint32), len(c),
                                            _ptr(f, C.c_uint32), _ptr(l, C.c_uint16), int(fwd_bypass), _ptr(out, C.c_uint8)))
        return out

    # a5-a7
    def pair_score(self, i_ids, j_ids, strand):
        i = np.ascontiguousarray(i_ids, np.uint32)
        j = np.ascontiguousarray(j_ids, np.uint32)
        s = np.ascontiguousarray(strand, np.uint8)
        n = len(i)
        bases = np.zeros(n, np.int32); hc = np.zeros(n, np.int32); nd = np.zeros(n, np.int32)
        nm = np.zeros(n, np.int32); var = np.zeros(n, np.float64)
        check(self.lib.rattle_hip_pair_score(self.h, _ptr(i, C.c_uint32), _ptr(j, C.c_uint32), _ptr(s, C.c_uint8), n,
                                             _ptr(bases, C.c_int32), _ptr(hc, C.c_int32), _ptr(nd, C.c_int32),
                                             _ptr(var, C.c_double), _ptr(nm, C.c_int32)))
        return bases, hc, nd, var, nm

    # a8-a11
    def cluster_reads(self, t_s=0.2, t_v=1000000.0, bv_threshold=0.4, min_bv_threshold=0.2, bv_falloff=0.05,
                      min_reads_cluster=0, use_hc=False, repr_percentile=0.15, is_rna=False,
                      subset: Optional[np.ndarray] = None) -> Clusters:
        P = ClusterParams(t_s, t_v, bv_threshold, min_bv_threshold, bv_falloff, min_reads_cluster, int(use_hc),
                          repr_percentile, int(is_rna))
        out = C.POINTER(ClusterSet)()
        if subset is None:
            check(self.lib.rattle_hip_cluster_reads(self.h, C.byref(P), C.byref(out)))
        else:
            sub = np.ascontiguousarray(subset, np.uint32)
            check(self.lib.rattle_hip_cluster_subset(self.h, C.byref(P), _ptr(sub, C.c_uint32), len(sub), C.byref(out)))
        cs = out.contents
        nc = cs.n_clusters
        offsets = np.ctypeslib.as_array(cs.offsets, (nc + 1,)).copy()
        nm = int(offsets[nc])
        res = Clusters(np.ctypeslib.as_array(cs.main_id, (max(nc, 1),))[:nc].copy(),
                       np.ctypeslib.as_array(cs.main_rev, (max(nc, 1),))[:nc].copy(), offsets,
                       np.ctypeslib.as_array(cs.member_id, (max(nm, 1),))[:nm].copy(),
                       np.ctypeslib.as_array(cs.member_rev, (max(nm, 1),))[:nm].copy(),
                       np.array(list(cs.counters), dtype=np.uint64))
        self.lib.rattle_hip_cluster_set_free(out)
        return res

    def cluster_subsets(self, subsets: Sequence[np.ndarray], t_s=0.2, t_v=1000000.0, bv_threshold=0.4, min_bv_threshold=0.2,
                        bv_falloff=0.05, repr_percentile=0.15, is_rna=False, n_workers=0) -> List[Clusters]:
        """cluster_reads restricted to each subset (ids in processing order), all subsets in one call."""
        P = ClusterParams(t_s, t_v, bv_threshold, min_bv_threshold, bv_falloff, 0, 0, repr_percentile, int(is_rna))
        n = len(subsets)
        offs = np.zeros(n + 1, np.uint64)
        offs[1:] = np.cumsum([len(x) for x in subsets])
        ids = np.concatenate([np.asarray(x, np.uint32) for x in subsets]) if n and offs[n] else np.zeros(1, np.uint32)
        outs = (C.POINTER(ClusterSet) * max(n, 1))()
        check(self.lib.rattle_hip_cluster_subsets(self.h, C.byref(P), _ptr(ids, C.c_uint32), _ptr(offs, C.c_uint64), n, outs, n_workers))
        return [self._take_clusters(outs[i]) for i in range(n)]

    def _take_clusters(self, out) -> Clusters:
        cs = out.contents
        nc = cs.n_clusters
        offsets = np.ctypeslib.as_array(cs.offsets, (nc + 1,)).copy()
        nm = int(offsets[nc])
        res = Clusters(np.ctypeslib.as_array(cs.main_id, (max(nc, 1),))[:nc].copy(),
                       np.ctypeslib.as_array(cs.main_rev, (max(nc, 1),))[:nc].copy(), offsets,
                       np.ctypeslib.as_array(cs.member_id, (max(nm, 1),))[:nm].copy(),
                       np.ctypeslib.as_array(cs.member_rev, (max(nm, 1),))[:nm].copy(),
                       np.array(list(cs.counters), dtype=np.uint64))
        self.lib.rattle_hip_cluster_set_free(out)
        return res

    def cluster_unsorted_packed(self, cat: np.ndarray, off: np.ndarray, k=10, t_s=0.2, t_v=1000000.0, bv_threshold=0.4,
                                min_bv_threshold=0.2, bv_falloff=0.05, repr_percentile=0.15, is_rna=False) -> Clusters:
        """main.cpp:254-277 on packed arrays in file order; ids in the result index the caller's order."""
        P = ClusterParams(t_s, t_v, bv_threshold, min_bv_threshold, bv_falloff, 0, 0, repr_percentile, int(is_rna))
        out = C.POINTER(ClusterSet)()
        check(self.lib.rattle_hip_cluster_unsorted(self.h, _ptr(cat, C.c_uint8), _ptr(off, C.c_uint64), len(off) - 1, k,
                                                   C.byref(P), C.byref(out)))
        return self._take_clusters(out)

    def cluster_iso_unsorted_packed(self, cat: np.ndarray, off: np.ndarray, k=10, iso_k=11, t_s=0.2, t_v=1000000.0, iso_t_s=0.3,
                                    iso_t_v=25.0, bv_threshold=0.4, min_bv_threshold=0.2, bv_falloff=0.05, repr_percentile=0.15,
                                    is_rna=False):
        """`rattle cluster --iso` (main.cpp:254-323) on packed arrays in file order.  Returns (Clusters, gene_id per
        cluster, number of gene clusters)."""
        P = ClusterParams(t_s, t_v, bv_threshold, min_bv_threshold, bv_falloff, 0, 0, repr_percentile, int(is_rna))
        Q = ClusterParams(iso_t_s, iso_t_v, bv_threshold, min_bv_threshold, bv_falloff, 0, 0, repr_percentile, int(is_rna))
        out = C.POINTER(ClusterSet)()
        ng = C.c_uint32()
        check(self.lib.rattle_hip_cluster_iso_unsorted(self.h, _ptr(cat, C.c_uint8), _ptr(off, C.c_uint64), len(off) - 1, k, iso_k,
                                                       C.byref(P), C.byref(Q), C.byref(out), C.byref(ng)))
        nc = out.contents.n_clusters
        gid = np.ctypeslib.as_array(out.contents.gene_id, (max(nc, 1),))[:nc].copy()
        return self._take_clusters(out), gid, ng.value

    def correct_packed(self, cat: np.ndarray, qcat: np.ndarray, off: np.ndarray, cl: Clusters, min_occ=0.3, gap_occ=0.3,
                       split=200, min_reads=5, n_threads=0, vote_order: bytes = b"", digest: bool = False, max_pack_cells=0,
                       gather_root: Optional[int] = None, keep: bool = False):
        """correct_reads on packed arrays; returns (n_corrected, n_uncorrected, n_consensi, counters)
        without materialising Python objects (the library still builds every output record).
        digest=True appends a CRC over every output array (ids, offsets, bases, qualities).
        gather_root: with several ranks, reassemble the sharded result on that rank
        (rattle_hip_correction_gather); the other ranks get zero counts and digest None.
        keep=True returns a CorrectionHandle instead (counts now, digest / free later)."""
        P, keep_alive = correct_params(min_occ, gap_occ, 30.0, split, min_reads, n_threads, vote_order, None, max_pack_cells)
        out = C.POINTER(Correction)()
        mid = cl.member_id if len(cl.member_id) else np.zeros(1, np.int32)
        mrev = cl.member_rev if len(cl.member_rev) else np.zeros(1, np.uint8)
        check(self.lib.rattle_hip_correct_reads(self.h, _ptr(cat, C.c_uint8), _ptr(qcat, C.c_uint8), _ptr(off, C.c_uint64),
                                                len(off) - 1, len(cl.main_id), _ptr(cl.offsets, C.c_uint32),
                                                _ptr(mid, C.c_int32), _ptr(mrev, C.c_uint8), C.byref(P), C.byref(out)))
        final = out
        if gather_root is not None:
            merged = C.POINTER(Correction)()
            check(self.lib.rattle_hip_correction_gather(self.h, out, gather_root, C.byref(merged)))
            self.lib.rattle_hip_correction_free(out)
            final = merged
        h = CorrectionHandle(self.lib, final if final else None)
        if keep:
            return h
        res = h.counts()
        if digest:
            res = res + (h.digest(),)
        h.free()
        return res

    # a15
    def poa_msa(self, packs: Sequence[Sequence[bytes]]):
        """MSA rows (list of bytes) for each pack of sequences."""
        flat = [s for p in packs for s in p]
        cat, off = pack_reads(flat)
        first = np.zeros(len(packs) + 1, np.uint32)
        first[1:] = np.cumsum([len(p) for p in packs])
        out = C.POINTER(MsaSet)()
        check(self.lib.rattle_hip_poa_msa(self.h, _ptr(cat, C.c_uint8), _ptr(off, C.c_uint64), len(flat),
                                          _ptr(first, C.c_uint32), len(packs), C.byref(out)))
        ms = out.contents
        ro = np.ctypeslib.as_array(ms.row_offset, (len(flat) + 1,)).copy()
        width = np.ctypeslib.as_array(ms.width, (max(len(packs), 1),))[:len(packs)].copy()
        total = int(ro[len(flat)])
        raw = C.string_at(ms.rows, total) if total else b""
        counters = np.array(list(ms.counters), dtype=np.uint64)
        self.lib.rattle_hip_msa_set_free(out)
        res = []
        q = 0
        for pi, p in enumerate(packs):
            rows = []
            for _ in p:
                rows.append(raw[int(ro[q]):int(ro[q + 1])])
                q += 1
            res.append(rows)
        return res, width, counters

    # a14-a20
    def correct_reads(self, seqs: Sequence[bytes], quals: Sequence[bytes], clusters, min_occ=0.3, gap_occ=0.3,
                      err_ratio=30.0, split=200, min_reads=5, n_threads=0, vote_order: bytes = b"", pack_order=None,
                      max_pack_cells=0, gather_root: Optional[int] = None):
        """correct_reads (correct.hpp:44).  `clusters` in rattle_amd.hps list form (ids index `seqs`).
        Returns dict of three record lists: (read_id, cluster_id, n_reads, seq, qual), the counters and
        the list of skipped packs.  With several ranks and gather_root set, the root gets the merged
        result and the others None."""
        cat, off = pack_reads(seqs)
        qcat, qoff = pack_reads(quals)
        assert np.array_equal(off, qoff), "sequence and quality lengths differ"
        coff = np.zeros(len(clusters) + 1, np.uint32)
        coff[1:] = np.cumsum([len(m) for _, m in clusters])
        mid = np.array([s[0] for _, m in clusters for s in m], np.int32)
        mrev = np.array([s[1] for _, m in clusters for s in m], np.uint8)
        if len(mid) == 0:
            mid = np.zeros(1, np.int32); mrev = np.zeros(1, np.uint8)
        P, keep = correct_params(min_occ, gap_occ, err_ratio, split, min_reads, n_threads, vote_order, pack_order, max_pack_cells)
        out = C.POINTER(Correction)()
        check(self.lib.rattle_hip_correct_reads(self.h, _ptr(cat, C.c_uint8), _ptr(qcat, C.c_uint8), _ptr(off, C.c_uint64),
                                                len(seqs), len(clusters), _ptr(coff, C.c_uint32), _ptr(mid, C.c_int32),
                                                _ptr(mrev, C.c_uint8), C.byref(P), C.byref(out)))
        final = out
        if gather_root is not None:
            merged = C.POINTER(Correction)()
            check(self.lib.rattle_hip_correction_gather(self.h, out, gather_root, C.byref(merged)))
            self.lib.rattle_hip_correction_free(out)
            final = merged
        if not final:
            return None
        res = unpack_correction(final.contents)
        self.lib.rattle_hip_correction_free(final)
        return res

    # ---- one job over several GPUs (include/rattle_hip.h, "One job over the GPUs of a node")
    def set_exchange_gloo(self, group=None):
        """Host-buffer all-gather through torch.distributed (any backend that takes CPU tensors, e.g. gloo)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def fn(user, send, send_bytes, recv, recv_bytes):
            try:
                sizes = [int(recv_bytes[r]) for r in range(world)]
                mine = torch.frombuffer(C.string_at(send, send_bytes) if send_bytes else b"\0", dtype=torch.uint8)[:send_bytes].clone()
                pad = max(sizes + [1])
                buf = torch.zeros(pad, dtype=torch.uint8)
                buf[:send_bytes] = mine
                parts = [torch.zeros(pad, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, buf, group=group)
                at = 0
                for r in range(world):
                    if sizes[r]:
                        C.memmove(recv + at, parts[r].numpy().ctypes.data, sizes[r])
                    at += sizes[r]
                return 0
            except Exception as e:  # never unwind through the C frame
                import sys
                print(f"exchange callback failed: {e}", file=sys.stderr)
                return 1

        self._xchg_fn = _lib.ALLGATHERV_FN(fn)       # keep the thunk alive
        check(self.lib.rattle_hip_set_exchange(self.h, rank, world, self._xchg_fn, None))

    def comm_init_rccl(self, group=None):
        """RCCL communicator for this context; the unique id travels through torch.distributed."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = np.zeros(128, np.uint8)
        if rank == 0:
            check(self.lib.rattle_hip_comm_unique_id(_ptr(uid, C.c_uint8)))
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.from_numpy(uid).to(dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        uid = t.cpu().numpy().copy()
        check(self.lib.rattle_hip_comm_init(self.h, rank, world, _ptr(uid, C.c_uint8)))

    def comm_destroy(self):
        check(self.lib.rattle_hip_comm_destroy(self.h))

    def comm_probe(self):
        """Collective self-test of the attached transport (all ranks)."""
        check(self.lib.rattle_hip_comm_probe(self.h))

    def comm_stats(self):
        a = C.c_uint64(); b = C.c_uint64()
        check(self.lib.rattle_hip_comm_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_stats(self, kernel: int):
        ms = C.c_double(); n = C.c_uint64(); b = C.c_uint64()
        check(self.lib.rattle_hip_kernel_stats(self.h, kernel, C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value

    def reset_stats(self):
        check(self.lib.rattle_hip_kernel_stats_reset(self.h))

    def stage_ms(self, reset: bool = True):
        """Host wall time of `correct`'s stages since the last reset: {stage: ms} (rattle_hip_stage_ms)."""
        ms = (C.c_double * 8)()
        check(self.lib.rattle_hip_stage_ms(self.h, ms, 1 if reset else 0))
        return {"1": ms[1], "2a": ms[2], "2b+3a": ms[3], "3b": ms[4]}


def min_common_lut(thr: float) -> np.ndarray:
    """min_common_lut[m] = smallest c with float(c)/float(m) >= thr (cluster.cpp:19,43); 0xFFFF = never."""
    lut = np.full(4097, 0xFFFF, np.uint16)
    for m in range(1, 4097):
        c = np.arange(0, 4097, dtype=np.float64) / float(m)
        idx = np.nonzero(c >= thr)[0]
        if len(idx):
            lut[m] = idx[0]
    return lut


def cluster_command(ctx: Context, seqs: Sequence[bytes], ann: Sequence[int], *, k=10, t_s=0.2, t_v=1000000.0,
                    iso=False, iso_k=11, iso_t_s=0.3, iso_t_v=25.0, bv_threshold=0.4, bv_min_threshold=0.2,
                    bv_falloff=0.05, repr_percentile=0.15, is_rna=False):
    """`rattle cluster` after input parsing (main.cpp:254-323).  `seqs`/`ann` are the filtered reads
    and their original record indices.  Returns clusters in rattle_amd.hps list form."""
    order = sorted(range(len(seqs)), key=lambda i: -len(seqs[i]))       # stable, length desc (fasta.cpp:462)
    sseqs = [seqs[i] for i in order]
    sann = [ann[i] for i in order]
    ctx.load_reads(sseqs, k, not is_rna)
    gene = ctx.cluster_reads(t_s, t_v, bv_threshold, bv_min_threshold, bv_falloff, 0, False, repr_percentile, is_rna)
    gl = gene.as_list()
    if not iso:
        return [((sann[m[0]], m[1], -1), [(sann[s[0]], s[1], -1) for s in mem]) for m, mem in gl], gene.counters
    ctx.load_reads(sseqs, iso_k, not is_rna)
    out = []
    counters = gene.counters.copy()
    subsets = []
    for m, mem in gl:
        ids = [s[0] for s in mem]
        ids.sort(key=lambda x: -x)                                       # main.cpp:285-291
        ids.sort(key=lambda x: -len(sseqs[x]))
        subsets.append(np.array(ids, np.uint32))
    subs = ctx.cluster_subsets(subsets, iso_t_s, iso_t_v, bv_threshold, bv_min_threshold, bv_falloff, repr_percentile, is_rna)
    for gi, (ids, sub) in enumerate(zip(subsets, subs)):
        counters += sub.counters
        for im, imem in sub.as_list():
            out.append(((sann[int(ids[im[0]])], im[1], gi), [(sann[int(ids[s[0]])], s[1], gi) for s in imem]))
    return out, counters


def correct_command(ctx: Context, headers: Sequence[bytes], seqs: Sequence[bytes], quals: Sequence[bytes], clusters, *,
                    min_occ=0.3, gap_occ=0.3, split=200, min_reads=5, n_threads=0, vote_order=b"", pack_order=None,
                    ann: Optional[Sequence[bytes]] = None, max_pack_cells=0, gather_root: Optional[int] = None,
                    with_skipped=False):
    """`rattle correct` after input parsing (main.cpp:396-408): returns the three FASTQ texts
    (corrected, uncorrected, consensi) with the headers correct.cpp:348-353,540-549 builds
    (no -l labels: `labels=` is empty).  `ann` = the third line of each input record: uncorrected reads
    keep theirs (the reference pushes the original read_t, correct.cpp:362-366,289-293); corrected reads
    and consensi get "+" (:286, :469)."""
    res = ctx.correct_reads(seqs, quals, clusters, min_occ, gap_occ, 30.0, split, min_reads, n_threads, vote_order, pack_order,
                            max_pack_cells, gather_root)
    if res is None:
        return None
    gene_mode = len(clusters) == 0 or clusters[0][0][2] == -1

    def tag(cid):
        gid = clusters[cid][0][2]
        if gid == -1:
            return b",gene_cluster_%d" % cid
        return b",gene_cluster_%d,transcript_cluster_%d" % (gid, cid)

    def fq(recs, keep_ann):
        return b"".join(b"%s%s\n%s\n%s\n%s\n" % (headers[r[0]], tag(r[1]), r[3], ann[r[0]] if keep_ann and ann is not None else b"+", r[4])
                        for r in recs)

    cons = []
    for rid, cid, nr, s, q in res["consensi"]:
        if gene_mode:
            h = b"@gene_cluster_%d reads=%d labels=" % (cid, nr)
        else:
            h = b"@transcript_cluster_%d gene_cluster_%d reads=%d labels=" % (cid, clusters[cid][0][2], nr)
        cons.append(b"%s\n%s\n+\n%s\n" % (h, s, q))
    out = (fq(res["corrected"], False), fq(res["uncorrected"], True), b"".join(cons), res["counters"])
    return out + (res["skipped"],) if with_skipped else out
