"""The sharding of ONE `correct` job over ranks, on the CPU (world size 2, gloo): the REAL partition code
(rattle_hip_plan_packs / rattle_hip_lpt_assign, the same functions correct_reads uses), a stub in place of
kernels C + D (a deterministic function of a pack's members), and the REAL reassembly
(rattle_hip_correction_gather over rattle_hip_set_exchange on a host-only context).  The merged result of the
two ranks must equal the result of one rank processing every pack.  The same flow with the real kernels runs
under -m gpu (tests/test_gpu_dist.py)."""
import ctypes as C
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from rattle_amd import _lib
from rattle_amd._lib import CorrectParams, PackPlan


def test_lpt_assign_is_deterministic_and_balanced():
    lib = _lib.load()
    rng = np.random.default_rng(0)
    cost = rng.integers(1, 1000, 500).astype(np.uint64)
    cost[:5] = 50000                                    # a few giants, like the Zipf head of a read set
    for R in (1, 2, 4, 8):
        own = np.zeros(len(cost), np.uint32)
        assert lib.rattle_hip_lpt_assign(cost.ctypes.data_as(C.POINTER(C.c_uint64)), len(cost), R, own.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
        own2 = np.zeros(len(cost), np.uint32)
        lib.rattle_hip_lpt_assign(cost.ctypes.data_as(C.POINTER(C.c_uint64)), len(cost), R, own2.ctypes.data_as(C.POINTER(C.c_uint32)))
        assert np.array_equal(own, own2) and own.max() == R - 1
        load = np.bincount(own, weights=cost.astype(np.float64), minlength=R)
        # LPT: no bin exceeds the mean by more than the largest item
        assert load.max() <= load.mean() + cost.max()
        if R <= 4:
            assert load.max() <= 1.05 * max(load.mean(), float(cost.max()))


def _plan(off, coff, mid, mrev, nranks, split=200, min_reads=5, max_pack_cells=0):
    lib = _lib.load()
    P = CorrectParams(0.3, 0.3, 30.0, split, min_reads, 0, b"")
    P.max_pack_cells = max_pack_cells
    out = C.POINTER(PackPlan)()
    rc = lib.rattle_hip_plan_packs(off.ctypes.data_as(C.POINTER(C.c_uint64)), len(off) - 1, len(coff) - 1, coff.ctypes.data_as(C.POINTER(C.c_uint32)),
                                   mid.ctypes.data_as(C.POINTER(C.c_int32)), mrev.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(P), nranks, C.byref(out))
    assert rc == 0, lib.rattle_hip_last_error()
    p = out.contents
    n = p.n_packs
    first = np.ctypeslib.as_array(p.pack_first, (n + 1,)).copy()
    res = dict(first=first, member=np.ctypeslib.as_array(p.member_id, (max(int(first[n]), 1),))[:int(first[n])].copy(),
               cluster=np.ctypeslib.as_array(p.pack_cluster, (max(n, 1),))[:n].copy(), local=np.ctypeslib.as_array(p.pack_local, (max(n, 1),))[:n].copy(),
               cost=np.ctypeslib.as_array(p.pack_cost, (max(n, 1),))[:n].copy(), owner=np.ctypeslib.as_array(p.pack_owner, (max(n, 1),))[:n].copy(),
               unq=np.ctypeslib.as_array(p.unqueued_id, (max(p.n_unqueued, 1),))[:p.n_unqueued].copy())
    lib.rattle_hip_pack_plan_free(out)
    return res


def test_plan_packs_follows_the_reference_pack_builder():
    """correct.cpp:328-370: n_files = (n-1)/split+1 strided sub-packs, queued iff size > min_reads (strict)."""
    rng = np.random.default_rng(1)
    sizes = [1, 5, 6, 7, 199, 200, 201, 399, 400, 401, 1000, 0, 13]
    lens = rng.integers(150, 3000, sum(sizes))
    off = np.zeros(len(lens) + 1, np.uint64); off[1:] = np.cumsum(lens)
    coff = np.zeros(len(sizes) + 1, np.uint32); coff[1:] = np.cumsum(sizes)
    mid = rng.permutation(len(lens)).astype(np.int32)
    mrev = rng.integers(0, 2, len(lens)).astype(np.uint8)
    pl = _plan(off, coff, mid, mrev, 3)
    want_packs, want_unq = [], []
    for c, n in enumerate(sizes):
        mem = list(mid[coff[c]:coff[c + 1]])
        if n == 0:
            continue
        nf = (n - 1) // 200 + 1
        for f in range(nf):
            sub = mem[f::nf]
            (want_packs if len(sub) > 5 else want_unq).append((c, sub))
    got = [(int(pl["cluster"][p]), list(pl["member"][pl["first"][p]:pl["first"][p + 1]])) for p in range(len(pl["cluster"]))]
    assert got == want_packs
    assert list(pl["unq"]) == [x for _, sub in want_unq for x in sub]
    L = np.diff(off.astype(np.int64))
    for p, (c, sub) in enumerate(want_packs):
        assert int(pl["cost"][p]) == int(L[sub].max()) * int(L[sub].sum())
    assert set(pl["owner"]) == {0, 1, 2}
    # the static budget rule moves whole packs out of the queue
    pl2 = _plan(off, coff, mid, mrev, 1, max_pack_cells=(6 * 2000 + 64) * 2000)
    long_packs = [p for p, (c, sub) in enumerate(want_packs) if L[sub].max() > 2000]
    assert len(pl2["cluster"]) == len(want_packs) - len(long_packs) and len(long_packs) > 0


WORKER = textwrap.dedent('''
    import ctypes as C, os, sys, zlib
    import numpy as np
    sys.path.insert(0, os.environ["RATTLE_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RATTLE_ROOT"], "tests"))
    import torch.distributed as dist
    from rattle_amd import _lib
    from rattle_amd._lib import Correction, ReadSet, SkipList
    from rattle_amd.api import Context, correction_digest
    from test_dist_cpu import _plan, stub_correction, make_job
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    off, coff, mid, mrev, seqs = make_job()
    plan = _plan(off, coff, mid, mrev, world)
    ctx = Context(None)                       # host-only context: exchange entry points
    ctx.set_exchange_gloo()
    local, keep = stub_correction(plan, seqs, rank)
    merged = C.POINTER(Correction)()
    rc = ctx.lib.rattle_hip_correction_gather(ctx.h, C.byref(local), 0, C.byref(merged))
    assert rc == 0, ctx.lib.rattle_hip_last_error()
    if rank == 0:
        one = _plan(off, coff, mid, mrev, 1)
        whole, keep2 = stub_correction(one, seqs, 0)
        assert correction_digest(merged.contents) == correction_digest(whole), "merged result differs from the unsharded one"
        R = merged.contents
        pk = np.ctypeslib.as_array(R.corrected_pack, (R.corrected.n,))
        assert np.all(np.diff(pk.astype(np.int64)) >= 0), "corrected reads not in pack order"
        assert int(R.counters[0]) == sum(int(c) for c in plan["cost"]) and R.skipped.n == 2
        calls, nbytes = ctx.comm_stats()
        assert calls >= 1 and nbytes > 0
        ctx.lib.rattle_hip_correction_free(merged)
        print("DIST_OK", world, R.corrected.n)
    else:
        assert not merged
    ctx.close()
    dist.destroy_process_group()
''')


def make_job():
    rng = np.random.default_rng(3)
    sizes = [450, 3, 230, 12, 7, 2, 640, 31]
    lens = rng.integers(200, 1500, sum(sizes))
    off = np.zeros(len(lens) + 1, np.uint64); off[1:] = np.cumsum(lens)
    coff = np.zeros(len(sizes) + 1, np.uint32); coff[1:] = np.cumsum(sizes)
    mid = rng.permutation(len(lens)).astype(np.int32)
    mrev = rng.integers(0, 2, len(lens)).astype(np.uint8)
    seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(l))) for l in lens]
    return off, coff, mid, mrev, seqs


def stub_correction(plan, seqs, rank):
    """Stands in for kernels C + D: a pack's 'corrected' reads are its members upper->lower-cased (every 7th member is
    'uncorrected'), consensi are complete on every rank (as after the library's exchange).  Builds a rattle_correction
    the way correct_reads leaves it on rank `rank` (its own packs only; unqueued members on rank 0)."""
    from rattle_amd._lib import Correction

    def read_set(recs):
        n = len(recs)
        keep = [np.array([r[0] for r in recs] + [0], np.int32), np.array([r[1] for r in recs] + [0], np.int32), np.array([r[2] for r in recs] + [0], np.int32)]
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(r[3]) for r in recs]) if n else []
        seq = C.create_string_buffer(b"".join(r[3] for r in recs) + b"\0")
        qual = C.create_string_buffer(b"".join(r[4] for r in recs) + b"\0")
        keep += [off, seq, qual]
        S = _lib.ReadSet(n, keep[0].ctypes.data_as(C.POINTER(C.c_int32)), keep[1].ctypes.data_as(C.POINTER(C.c_int32)), keep[2].ctypes.data_as(C.POINTER(C.c_int32)),
                         off.ctypes.data_as(C.POINTER(C.c_uint64)), C.cast(seq, C.POINTER(C.c_char)), C.cast(qual, C.POINTER(C.c_char)))
        return S, keep

    cor, unc, cor_pk, unc_pk = [], [], [], []
    if rank == 0:
        for rid in plan["unq"]:
            unc.append((int(rid), -1, 0, seqs[rid], b"!" * len(seqs[rid]))); unc_pk.append(0xFFFFFFFF)
    cells = 0
    for p in range(len(plan["cluster"])):
        if int(plan["owner"][p]) != rank:
            continue
        cells += int(plan["cost"][p])
        for j, rid in enumerate(plan["member"][plan["first"][p]:plan["first"][p + 1]]):
            rec = (int(rid), int(plan["cluster"][p]), 0, seqs[rid].lower(), b"#" * len(seqs[rid]))
            if j % 7 == 6:
                unc.append(rec); unc_pk.append(p)
            else:
                cor.append(rec); cor_pk.append(p)
    cons = [(-1, int(c), 1, b"ACGT" * (int(c) + 1), b"K" * (4 * (int(c) + 1))) for c in sorted(set(int(x) for x in plan["cluster"]))]
    R = Correction()
    keep = []
    for name, recs in (("corrected", cor), ("uncorrected", unc), ("consensi", cons)):
        S, k = read_set(recs)
        setattr(R, name, S); keep.append(k)
    R.counters[0] = cells
    R.counters[2] = len(plan["cluster"])
    cp = np.array(cor_pk + [0], np.uint32); up = np.array(unc_pk + [0], np.uint32)
    R.corrected_pack = cp.ctypes.data_as(C.POINTER(C.c_uint32)); R.uncorrected_pack = up.ctypes.data_as(C.POINTER(C.c_uint32))
    # one skipped pack reported by each of the first two ranks
    sk = [np.array([rank], np.int32), np.array([rank], np.uint32), np.array([1], np.uint32), np.array([0, 2], np.uint64), np.array([5 + rank, 9 + rank], np.int32)]
    if rank < 2:
        R.skipped = _lib.SkipList(1, sk[0].ctypes.data_as(C.POINTER(C.c_int32)), sk[1].ctypes.data_as(C.POINTER(C.c_uint32)), sk[2].ctypes.data_as(C.POINTER(C.c_uint32)),
                                  sk[3].ctypes.data_as(C.POINTER(C.c_uint64)), sk[4].ctypes.data_as(C.POINTER(C.c_int32)))
        R.counters[3] = 1; R.counters[4] = 2
    else:
        z = np.zeros(1, np.uint64)
        R.skipped = _lib.SkipList(0, sk[0].ctypes.data_as(C.POINTER(C.c_int32)), sk[1].ctypes.data_as(C.POINTER(C.c_uint32)), sk[2].ctypes.data_as(C.POINTER(C.c_uint32)),
                                  z.ctypes.data_as(C.POINTER(C.c_uint64)), sk[4].ctypes.data_as(C.POINTER(C.c_int32)))
        keep.append(z)
    keep += [cp, up, sk]
    return R, keep


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_correct_reassembles_to_the_unsharded_result(tmp_path, world):
    """world 8 = the node the BASELINE metric is quoted on: the LPT partition with 8 bins, pack_owner, and the gather order"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RATTLE_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29600 + world), str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert f"DIST_OK {world}" in r.stdout
