"""The lemma behind DESIGN.md §8's "exact band for the chains" (an idea for kernel C's POA #3 groups, NOT in the product): for spoa's local
alignment scores (match 5, mismatch -4, gap of length k: -8 - 6 (k - 1)), sequence against sequence, a band of half-width w around the
diagonals that a gapless alignment of the full shorter sequence can use is EXACT whenever the best score found inside it exceeds
5 (L - w - 1), L = the shorter length: a path that visits a cell t > w diagonals outside has at most L - t diagonal steps or pays for
2 t gap positions, so it scores at most 5 (L - t) <= 5 (L - w - 1).  Checked here on random pairs with plain Python DP matrices: same best
score, same set of best cells, same traceback from the first of them.  A pair with an indel longer than the band must fail the condition
(the caller then runs the full rows)."""
import numpy as np

M, N, G, E = 5, -4, -8, -6
NEG = -10 ** 9


def dp(a, b, band=None):
    """H, F, E matrices of spoa's affine local alignment (rows: a, columns: b); band = (lo, hi): only cells with lo <= i - j <= hi exist."""
    n, m = len(a), len(b)
    H = [[0] * (m + 1) for _ in range(n + 1)]
    F = [[NEG] * (m + 1) for _ in range(n + 1)]
    Ee = [[NEG] * (m + 1) for _ in range(n + 1)]
    inside = (lambda i, j: True) if band is None else (lambda i, j: band[0] <= i - j <= band[1])
    if band is not None:
        for i in range(n + 1):
            for j in range(m + 1):
                if not inside(i, j):
                    H[i][j] = NEG
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            if not inside(i, j):
                continue
            F[i][j] = max(H[i - 1][j] + G, F[i - 1][j] + E)
            Ee[i][j] = max(H[i][j - 1] + G, Ee[i][j - 1] + E)
            H[i][j] = max(0, H[i - 1][j - 1] + (M if a[i - 1] == b[j - 1] else N), F[i][j], Ee[i][j])
    return H, F, Ee


def best_cells(H):
    s = max(max(r) for r in H)
    return s, [(i, j) for i, r in enumerate(H) for j, v in enumerate(r) if v == s]


def traceback(a, b, H, F, Ee, start):
    """Diagonal first, then the vertical gap, then the horizontal one; stops where H is 0."""
    i, j = start
    path = []
    while i > 0 and j > 0 and H[i][j] > 0:
        path.append((i, j))
        if H[i][j] == H[i - 1][j - 1] + (M if a[i - 1] == b[j - 1] else N):
            i, j = i - 1, j - 1
        elif H[i][j] == F[i][j]:
            while F[i][j] != H[i - 1][j] + G:
                i -= 1
                path.append((i, j))
            i -= 1
        else:
            assert H[i][j] == Ee[i][j]
            while Ee[i][j] != H[i][j - 1] + G:
                j -= 1
                path.append((i, j))
            j -= 1
    return path


def mutate(rng, a, rate):
    out = []
    for c in a:
        r = rng.random()
        if r < rate * 0.3:
            continue
        out.append(int(rng.integers(0, 4)) if r < rate * 0.7 else c)
        if rng.random() < rate * 0.3:
            out.append(int(rng.integers(0, 4)))
    return out


def band_of(n, m, w):
    d = n - m
    return (min(0, d) - w, max(0, d) + w)


def test_band_is_exact_when_its_best_score_clears_the_bound():
    rng = np.random.default_rng(11)
    held = 0
    for case in range(120):
        n = int(rng.integers(40, 90))
        a = [int(x) for x in rng.integers(0, 4, n)]
        b = mutate(rng, a, float(rng.choice([0.02, 0.05, 0.10])))
        w = int(rng.choice([4, 6, 8, 12]))
        L = min(len(a), len(b))
        Hb, Fb, Eb = dp(a, b, band_of(len(a), len(b), w))
        sb, cb = best_cells(Hb)
        if sb <= 5 * (L - w - 1):
            continue
        held += 1
        Hf, Ff, Ef = dp(a, b)
        sf, cf = best_cells(Hf)
        assert sf == sb and cf == cb, (case, sf, sb)
        assert traceback(a, b, Hf, Ff, Ef, cf[0]) == traceback(a, b, Hb, Fb, Eb, cb[0]), case
    assert held >= 40      # the condition is the common case for near-identical sequences, not a vacuous one


def test_an_indel_longer_than_the_band_fails_the_condition():
    rng = np.random.default_rng(12)
    a = [int(x) for x in rng.integers(0, 4, 80)]
    w = 6
    b = a[:40] + [int(x) for x in rng.integers(0, 4, 3 * w)] + a[40:]
    L = min(len(a), len(b))
    sb, _ = best_cells(dp(a, b, band_of(len(a), len(b), w))[0])
    sf, _ = best_cells(dp(a, b)[0])
    assert sb <= 5 * (L - w - 1)                                  # the band cannot vouch for itself: full rows
    assert sf >= sb
