"""Kernel B's full pass (rattle_amd/csrc/pair_score.hip, round 3) does not walk the match list and the chain one element at a
time: it takes RUNS -- lane t of a 64-element step tests its element against element t-1, one ballot gives the length of the
leading run of passes, the run is applied at once, the element behind it takes the sequential step.  This file restates that
decomposition in plain Python (64 "lanes" per step, the same carried state) and checks it against the oracle's
calc_similarity (oracle/orc_cluster.hpp, pinned on the reference's own similarity.cpp) on reads whose match lists are full of
non-extending elements and whose chains are full of elements that are not kept.  No GPU: it pins the ALGORITHM the kernel
implements; the kernel itself is compared with the oracle in tests/test_gpu_cluster.py."""
import numpy as np
import pytest


def _matches(a: bytes, b: bytes, k: int):
    """get_common_kmers (kmer.cpp:45-67): every (pos1, pos2) with equal k-mers, sorted by (pos1, pos2)."""
    where = {}
    for p in range(len(b) - k):                 # extract_kmers_from_read stops at L - k (kmer.cpp:10)
        where.setdefault(b[p:p + k], []).append(p)
    out = []
    for p in range(len(a) - k):
        for q in where.get(a[p:p + k], ()):
            out.append((p, q))
    return out


def _by_runs(pos1, pos2, k):
    """The kernel's two loops, 64 lanes per step."""
    M = len(pos2)
    l = 0
    tv = [0] * (M + 2); m = [0] * (M + 2); pp = [0] * max(M, 1)
    tail_last = m_last = 0
    searched = False
    for base in range(0, M, 64):
        nb = min(64, M - base)
        xv = [pos2[base + t] if t < nb else 0 for t in range(64)]
        xsh = [0] + xv[:63]
        t0 = 0
        while t0 < nb:
            ext = [t0 <= t < nb and (xv[t] > (tail_last if t == t0 else xsh[t]) or (l == 0 and t == t0)) for t in range(64)]
            run = 0
            while t0 + run < 64 and ext[t0 + run]:
                run += 1
            if run:
                for t in range(t0, t0 + run):
                    i = base + t; rk = l + 1 + t - t0
                    pp[i] = m_last if t == t0 else i - 1; m[rk] = i; tv[rk] = xv[t]
                l += run; tail_last = xv[t0 + run - 1]; m_last = base + t0 + run - 1; t0 += run
            if t0 < nb:
                i = base + t0; x = xv[t0]; searched = True
                lo = 1 + sum(1 for idx in range(1, l + 1) if tv[idx] < x)
                pp[i] = m[lo - 1]; m[lo] = i; tv[lo] = x
                if lo == l:
                    tail_last = x; m_last = i
                t0 += 1
    if l == 0:
        return 0, 0, []
    if not searched:
        chain = list(range(l))
    else:
        chain = [0] * l; cur = m[l]
        for i in range(l - 1, -1, -1):
            chain[i] = cur; cur = pp[cur]
    kf, ks = pos1[chain[0]], pos2[chain[0]]
    prev_s = ks
    vb = vh = 0
    dists = []
    for base in range(1, l, 64):
        nb = min(64, l - base)
        fv = [pos1[chain[base + t]] if t < nb else 0 for t in range(64)]
        sv = [pos2[chain[base + t]] if t < nb else 0 for t in range(64)]
        fsh = [0] + fv[:63]; ssh = [0] + sv[:63]
        t0 = 0
        while t0 < nb:
            D = [(fv[t] - (kf if t == t0 else fsh[t]), sv[t] - (ks if t == t0 else ssh[t])) for t in range(64)]
            keep = [t0 <= t < nb and ((D[t][0] < k and D[t][1] < k) or (D[t][0] >= k and D[t][1] >= k)) for t in range(64)]
            run = 0
            while t0 + run < 64 and keep[t0 + run]:
                run += 1
            if run:
                for t in range(t0, t0 + run):
                    ex = k - (sv[t] - (prev_s if t == 0 else ssh[t]))
                    cb = k - ex if ex > 0 else k
                    dist = D[t][1] - D[t][0]
                    dists.append(dist); vb += cb
                    if dist < 10:
                        vh += cb
                kf, ks = fv[t0 + run - 1], sv[t0 + run - 1]; t0 += run
            if t0 < nb:
                t0 += 1
        prev_s = sv[nb - 1]
    return k + vb, k + vh, dists


@pytest.mark.parametrize("k", [10, 11])
def test_runs_equal_the_reference_walks(oracle, k):
    rng = np.random.default_rng(7 + k)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seg = [acgt[rng.integers(0, 4, n)] for n in (210, 95, 160, 64, 220, 33, 130)]
    orders = [(0, 1, 2, 3, 4, 5, 6), (0, 2, 1, 3, 4, 6, 5), (0, 1, 1, 2, 3, 4, 5, 6), (0, 2, 4, 6), (4, 5, 6, 0, 1, 2, 3),
              (0, 1, 2, 2, 2, 3, 4), (0, 3, 4), (1, 3, 5, 1, 3, 5, 1, 3, 5)]
    reads = []
    for o in orders:
        tx = np.concatenate([seg[i] for i in o])
        for err in (0.0, 0.06):
            r = rng.random(len(tx))
            sq = tx.copy()
            sub = r < err * 0.5
            sq[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
            reads.append((sq[r >= err * 0.25] if err else sq).tobytes())
    searched = dropped = 0
    for a in reads:
        for b in reads:
            ms = _matches(a, b, k)
            bases, hc, dists = _by_runs([p for p, _ in ms], [q for _, q in ms], k)
            wb, wh, wn, _, wm, wd = oracle.pair_score(a, b, k, 0, dist_cap=1 << 14)
            assert wm == len(ms)
            assert (bases, len(dists)) == (wb, wn) and list(wd) == dists
            if wm:
                assert hc == wh
            searched += wm > 0 and wn + 1 < wm
            dropped += any(abs(d) >= k for d in dists)
    assert searched > 50 and dropped > 20


def _swapped_walk_network(n):
    """Comparator schedule of the sort behind kernel B's swapped walk (pair_score.hip: a long seed against a short candidate
    walks the SHORT list, the matches come in the candidate's hash order and go back into (pos1, pos2) order): the all-ascending
    bitonic form, comparators whose partner lies beyond n skipped -- so nothing beyond n is ever touched."""
    p2 = 2
    while p2 < n:
        p2 <<= 1
    k2 = 2
    while k2 <= p2:
        j2 = k2 >> 1
        while j2 > 0:
            yield k2 - 1 if j2 == (k2 >> 1) else j2
            j2 >>= 1
        k2 <<= 1


@pytest.mark.parametrize("n", list(range(2, 70)) + [127, 128, 129, 255, 256, 257, 300, 399, 400])
def test_swapped_walk_sort_network_sorts_any_length(n):
    """The advisor's round-3 finding: the power-of-two padded network wrote past the `cap` = 400 entries of pos1 / pos2 for
    256 < n <= 400.  The network the kernel runs now is restated here and must sort every length without padding."""
    rng = np.random.default_rng(n)
    key = rng.integers(0, 40, n) * 100000 + rng.permutation(n)          # distinct (pos1, pos2) pairs as one key
    a = key.copy()
    for flip in _swapped_walk_network(n):
        t = np.arange(n)
        u = t ^ flip
        sel = (u > t) & (u < n)
        lo, hi = t[sel], u[sel]
        sw = a[lo] > a[hi]
        a[lo[sw]], a[hi[sw]] = a[hi[sw]].copy(), a[lo[sw]].copy()
    assert np.array_equal(a, np.sort(key))
