"""Parity of the HIP cluster path (kernels K, A, B + greedy driver) with the oracle, through
the C ABI.  Bit-exact: integer / index work, and the double variance compared by bits."""
import numpy as np
import pytest

from rattle_amd import synth
from rattle_amd.api import Context, cluster_command, min_common_lut

pytestmark = pytest.mark.gpu


def _same_float(a, b):
    return a == b or (np.isnan(a) and np.isnan(b))


@pytest.fixture(scope="module")
def small_reads():
    seqs, quals, tid, flip = synth.reads(400, 8, 2, True, seed=3)
    order = sorted(range(len(seqs)), key=lambda i: -len(seqs[i]))
    return [seqs[i] for i in order]


@pytest.mark.parametrize("sort", ["radix", "bitonic"])
@pytest.mark.parametrize("k", [6, 10, 11, 16])
def test_kmer_index_matches_oracle(gpu_ctx, oracle, small_reads, k, sort, monkeypatch):
    """Kernel K with both list sorts: the block radix sort (default; lists of 256 .. 8192 k-mers) and the LDS bitonic network
    (shorter lists, and RATTLE_KMER_SORT=bitonic)."""
    monkeypatch.setenv("RATTLE_KMER_SORT", sort)
    reads = small_reads[:64] + [b"ACGTAC", b"ACGTACG", b"A" * 40, b"ACGU" * 30, b"G" * 700 + b"ACGT" * 100]     # edge: L<=k, L=k+1, homopolymers, U
    gpu_ctx.load_reads(reads, k, True)
    for r, s in enumerate(reads):
        fh, fp, rh, rp, bf, br = oracle.extract_kmers(s, k, True)
        h, p, bv, pc = gpu_ctx.read_index(r, 0)
        assert np.array_equal(h, fh) and np.array_equal(p, fp) and np.array_equal(bv, bf)
        assert pc == sum(bin(int(w)).count("1") for w in bf)
        h, p, bv, pc = gpu_ctx.read_index(r, 1)
        assert np.array_equal(h, rh) and np.array_equal(p, rp) and np.array_equal(bv, br)


def test_invalid_base_is_an_error(gpu_ctx):
    from rattle_amd._lib import RattleError
    with pytest.raises(RattleError):
        gpu_ctx.load_reads([b"ACGTNACGTACGTACGT" * 4], 10, False)


def test_long_read_global_sort_path(gpu_ctx, oracle):
    rng = np.random.default_rng(9)
    s = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 20000)])
    gpu_ctx.load_reads([s, s[:9000]], 11, True)
    for r, q in enumerate([s, s[:9000]]):
        fh, fp, rh, rp, bf, br = oracle.extract_kmers(q, 11, True)
        h, p, bv, _ = gpu_ctx.read_index(r, 1)
        assert np.array_equal(h, rh) and np.array_equal(p, rp) and np.array_equal(bv, br)


@pytest.mark.parametrize("thr", [0.4, 0.35000000000000003, 0.20000000000000007, 0.0])
def test_bv_filter_matches_oracle(gpu_ctx, oracle, small_reads, thr):
    reads = small_reads[:300]
    gpu_ctx.load_reads(reads, 10, True)
    idx = [oracle.extract_kmers(s, 10, True) for s in reads]
    bf = np.array([x[4] for x in idx]); br = np.array([x[5] for x in idx])
    pop = np.vectorize(lambda w: bin(int(w)).count("1"))
    pcf = pop(bf).sum(1)
    seeds = np.arange(0, 70, dtype=np.uint32)             # spans three seed tiles
    cands = np.arange(0, 300, dtype=np.uint32)
    first = seeds + 1
    got = gpu_ctx.bv_filter(seeds, cands, first, min_common_lut(thr), thr == 0.0)
    for s in seeds:
        for c in cands:
            want = 0
            if c >= first[s]:
                mmax = float(max(pcf[s], pcf[c]))
                cf = float(pop(bf[s] & bf[c]).sum()); cr = float(pop(bf[s] & br[c]).sum())
                if thr == 0.0 or cf / mmax >= thr:
                    want |= 1
                if cr / mmax >= thr:
                    want |= 2
            assert got[s, c] == want, (s, c)


@pytest.mark.parametrize("k", [10, 11, 6])
def test_pair_score_matches_oracle(gpu_ctx, oracle, small_reads, k):
    reads = small_reads[:120] + [b"ACACACACACACACACACACACACACACACAC" * 40, b"CACACACACACACACACACACACACACACACA" * 40 + b"GGT"]
    gpu_ctx.load_reads(reads, k, True)
    rng = np.random.default_rng(k)
    ii = rng.integers(0, len(reads), 600); jj = rng.integers(0, len(reads), 600); ss = rng.integers(0, 2, 600)
    ii[-1], jj[-1], ss[-1] = len(reads) - 2, len(reads) - 1, 0          # ~1.6M matches: global-scratch path
    bases, hc, nd, var, nm = gpu_ctx.pair_score(ii, jj, ss)
    big = 0
    for t in range(600):
        b, h, n, v, m, _ = oracle.pair_score(reads[ii[t]], reads[jj[t]], k, int(ss[t]), dist_cap=1)
        assert (bases[t], nd[t], nm[t]) == (b, n, m), t
        if m > 0:
            assert hc[t] == h
        assert _same_float(var[t], v), (t, var[t], v)
        big += m > 512
    assert big >= 1


@pytest.mark.parametrize("k", [10, 11])
def test_pair_score_on_rearranged_reads(gpu_ctx, oracle, k):
    """Kernel B's full pass advances by runs: chain-extending matches and kept chain elements 64 at a time, anything else one by
    one (pair_score.hip).  Reads made of the same segments in other orders, with a segment doubled, dropped or repeated in
    tandem give match lists full of elements that do NOT extend the chain, chains with elements that are NOT kept (exon-skip
    like jumps, which cluster.cpp:28-34 rejects on variance), runs that end inside a 64-element step, and both strands."""
    rng = np.random.default_rng(100 + k)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seg = [acgt[rng.integers(0, 4, n)] for n in (310, 95, 260, 64, 420, 33, 180)]
    orders = [(0, 1, 2, 3, 4, 5, 6), (0, 2, 1, 3, 4, 6, 5), (0, 1, 1, 2, 3, 4, 5, 6), (0, 2, 4, 6), (4, 5, 6, 0, 1, 2, 3), (0, 1, 2, 2, 2, 3, 4),
              (6, 5, 4, 3, 2, 1, 0), (0, 3, 4), (1, 3, 5, 1, 3, 5, 1, 3, 5), (0, 1, 2, 3, 4, 5, 6, 0, 1, 2)]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for o in orders:
        tx = np.concatenate([seg[i] for i in o])
        for err in (0.0, 0.03, 0.08, 0.12):
            r = rng.random(len(tx))
            sq = tx.copy()
            sub = r < err * 0.5
            sq[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
            sq = sq[r >= err * 0.25] if err else sq
            b = sq.tobytes()
            reads.append(b if len(reads) % 3 else b.translate(comp)[::-1])
    gpu_ctx.load_reads(reads, k, True)
    n = len(reads)
    ii = np.repeat(np.arange(n), n); jj = np.tile(np.arange(n), n)
    ss = np.asarray(rng.integers(0, 2, n * n), dtype=np.int64)
    bases, hc, nd, var, nm = gpu_ctx.pair_score(ii, jj, ss)
    searched = kept_gap = 0
    for t in range(n * n):
        b, h, d, v, m, _ = oracle.pair_score(reads[ii[t]], reads[jj[t]], k, int(ss[t]), dist_cap=1)
        assert (bases[t], nd[t], nm[t]) == (b, d, m), (t, ii[t], jj[t], ss[t])
        if m > 0:
            assert hc[t] == h, t
        assert _same_float(var[t], v), (t, var[t], v)
        searched += m > 0 and d + 1 < m           # matches that did not all end up as kept chain elements
        kept_gap += v > 25.0
    assert searched > 200 and kept_gap > 50


@pytest.mark.parametrize("k", [10, 11])
def test_pair_score_long_seed_against_a_fragment_of_it(gpu_ctx, oracle, k):
    """The swapped walk of kernel B (pair_score.hip: nA > 4 nB + 256 -- a long seed against a short candidate walks the SHORT list
    and sorts the matches back into (pos1, pos2) order).  A 300-450 nt fragment of a > 2 kb read shares 257..400 k-mers with it:
    the sort then works on more than 256 entries in arrays of 400 (round 3's advisor finding: the padded network ran past them).
    Both strands, clean and noisy fragments, fragments with an internal repeat; every pair against the oracle."""
    rng = np.random.default_rng(900 + k)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for _ in range(6):
        reads.append(acgt[rng.integers(0, 4, int(rng.integers(3000, 3900)))].tobytes())
    n_long = len(reads)
    for li in range(n_long):
        L = reads[li]
        for flen in (300, 340, 395, 430, 450):
            a = int(rng.integers(0, len(L) - flen))
            for err in (0.0, 0.02):
                f = np.frombuffer(L[a:a + flen], np.uint8).copy()
                r = rng.random(flen)
                f[r < err] = acgt[rng.integers(0, 4, int((r < err).sum()))]
                b = f.tobytes()
                if len(reads) % 4 == 1:
                    b = b[:150] + b[60:150] + b[150:]                    # an internal repeat: cross products of equal hashes
                reads.append(b if len(reads) % 2 else b.translate(comp)[::-1])
    gpu_ctx.load_reads(reads, k, True)
    n = len(reads)
    ii = np.repeat(np.arange(n_long), n - n_long); jj = np.tile(np.arange(n_long, n), n_long)
    ii = np.concatenate([ii, ii]); jj = np.concatenate([jj, jj])
    ss = np.concatenate([np.zeros(len(ii) // 2, np.int64), np.ones(len(ii) // 2, np.int64)])
    bases, hc, nd, var, nm = gpu_ctx.pair_score(ii, jj, ss)
    big = 0
    for t in range(len(ii)):
        assert len(reads[ii[t]]) - k > 4 * (len(reads[jj[t]]) - k) + 256       # the swapped walk is what runs
        b, h, d, v, m, _ = oracle.pair_score(reads[ii[t]], reads[jj[t]], k, int(ss[t]), dist_cap=1)
        assert (bases[t], nd[t], nm[t]) == (b, d, m), (t, ii[t], jj[t], ss[t], (bases[t], nd[t], nm[t]), (b, d, m))
        if m > 0:
            assert hc[t] == h, t
        assert _same_float(var[t], v), (t, var[t], v)
        big += 256 < m <= 400
    assert big >= 20


def _as_oracle_list(cl):
    return cl.as_list()


@pytest.mark.parametrize("is_rna", [False, True])
def test_cluster_reads_synthetic_matches_oracle(gpu_ctx, oracle, is_rna):
    seqs, _, _, _ = synth.reads(1500, 12, 2, not is_rna, seed=21)
    order = sorted(range(len(seqs)), key=lambda i: -len(seqs[i]))
    reads = [seqs[i] for i in order]
    gpu_ctx.load_reads(reads, 10, not is_rna)
    got = gpu_ctx.cluster_reads(is_rna=is_rna)
    want, _ = oracle.cluster_reads(reads, k=10, is_rna=is_rna)
    assert got.as_list() == want
    # iso parameters (k=11, 0.3, 25) exercise the variance threshold
    gpu_ctx.load_reads(reads, 11, not is_rna)
    got = gpu_ctx.cluster_reads(t_s=0.3, t_v=25.0, is_rna=is_rna)
    want, _ = oracle.cluster_reads(reads, k=11, t_s=0.3, t_v=25.0, is_rna=is_rna)
    assert got.as_list() == want
    if not is_rna:
        assert any(s[1] for _, mem in want for s in mem)          # reverse-strand members exist


def test_cluster_toyset_matches_reference_fixture(gpu_ctx, toyset, toyset_clusters):
    """`cluster --rna` on the recovered toyset == toyset/rna/output/clusters.out (546 clusters)."""
    seqs = [r[1] for r in toyset]
    got, counters = cluster_command(gpu_ctx, seqs, list(range(len(seqs))), k=10, is_rna=True)
    want = [((m[0], m[1], -1), [(s[0], s[1], -1) for s in mem]) for m, mem in toyset_clusters]
    assert len(got) == 546 and got == want


def test_cluster_subset_and_iso_flow(gpu_ctx, oracle):
    seqs, _, _, _ = synth.reads(900, 6, 3, True, seed=33)
    got, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))), iso=True)
    # oracle: same flow through orc::cluster_reads per gene cluster
    order = sorted(range(len(seqs)), key=lambda i: -len(seqs[i]))
    reads = [seqs[i] for i in order]
    gene, _ = oracle.cluster_reads(reads, k=10)
    want = []
    for gi, (m, mem) in enumerate(gene):
        ids = sorted([s[0] for s in mem], key=lambda x: -x)
        ids.sort(key=lambda x: -len(reads[x]))
        sub, _ = oracle.cluster_reads([reads[i] for i in ids], k=11, t_s=0.3, t_v=25.0)
        for im, imem in sub:
            want.append(((order[ids[im[0]]], im[1], gi), [(order[ids[s[0]]], s[1], gi) for s in imem]))
    assert got == want


def test_cluster_unsorted_matches_cluster_command(gpu_ctx, oracle):
    """rattle_hip_cluster_unsorted == sort + cluster_reads + id translation (main.cpp:254-277)."""
    from rattle_amd.api import pack_reads
    seqs, _, _, _ = synth.reads(800, 7, 1, True, seed=44)
    want, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))))
    cat, off = pack_reads(seqs)
    got = gpu_ctx.cluster_unsorted_packed(cat, off).as_list()
    assert got == want


@pytest.mark.parametrize("is_rna", [False, True])
def test_cluster_iso_unsorted_matches_iso_flow(gpu_ctx, is_rna):
    """rattle_hip_cluster_iso_unsorted == the two-level flow of main.cpp:254-323 as cluster_command runs it
    (itself checked against the oracle above): same transcript clusters, gene ids and order; also with staged reads."""
    from rattle_amd.api import pack_reads
    seqs, _, _, _ = synth.reads(1200, 8, 3, not is_rna, seed=35)
    want, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))), iso=True, is_rna=is_rna)
    cat, off = pack_reads(seqs)
    for staged in (False, True):
        if staged:
            gpu_ctx.stage_reads(cat, None, off)
        try:
            cl, gid, ng = gpu_ctx.cluster_iso_unsorted_packed(cat, off, is_rna=is_rna)
        finally:
            if staged:
                gpu_ctx.unstage_reads()
        got = [((m[0], m[1], int(g)), [(s[0], s[1], int(g)) for s in mem]) for (m, mem), g in zip(cl.as_list(), gid)]
        assert got == want
        assert ng == len(set(g for (_, _, g), _ in want))


def test_rna_mode_rejects_a_both_strand_index(gpu_ctx):
    """cluster.cpp:42 returns before the reverse test in --rna mode; a both-strand index would let reverse hits through."""
    from rattle_amd._lib import RattleError
    seqs, _, _, _ = synth.reads(50, 2, 1, True, seed=1)
    gpu_ctx.load_reads(sorted(seqs, key=lambda s: -len(s)), 10, True)
    with pytest.raises(RattleError):
        gpu_ctx.cluster_reads(is_rna=True)
