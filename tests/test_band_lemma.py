"""The lemma behind kernel C's exact band for near-chain graphs (dp_rows_band, DESIGN.md §4: POA #2 / #3 of `rattle correct`,
correct.cpp:427-436,520-532): for spoa's local alignment scores (match 5, mismatch -4, gap of length k: -8 - 6 (k - 1)) a band around the
diagonals a path with at least tau = L - t diagonal moves can use is EXACT whenever the best score found inside it is >= 5 tau - 4: a path
that touches a cell outside has fewer than tau diagonal moves and scores at most 5 (tau - 1).  Checked here on plain Python DP matrices --
sequence against sequence first (round 5's groundwork), then sequence against a DAG with aligned groups, insertions and skip edges, the
band placed by the rows' MSA columns as the kernel does: same best score, same set of best cells, same traceback from them.  A pair with
an indel longer than the band must fail the condition (the kernel then runs the full rows)."""
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


# ---- the same for a DAG (kernel C's dp_rows_band, round 6): rows in block order, the rows of an aligned group share an MSA column ----
# A path through cell (row i, column j) has at most M(i, j) = min(c_i, j) + min(C - c_i, L - j) diagonal moves (c_i = the row's MSA
# column, C = columns of the graph), provided every edge leads to a strictly later column.  With tau = L - t the band is
# {M >= tau} = {max(1, c_i - (C - L + t)) <= j <= min(L, c_i + t)}, cells outside read H = 0 / F = E = -inf, and the certificate is
# best_in_band >= 5 tau - 4.
def random_near_chain_dag(rng, length, subs, inss, skips):
    """rows (block order): dict(letter, preds [rows], col); a backbone chain, substitution siblings (same column), insertion nodes
    (a column of their own between two backbone columns), skip edges (deletions)."""
    cols = []                                            # list of columns, each a list of (letter, kind)
    for k in range(length):
        cols.append([int(rng.integers(0, 4))])
        if rng.random() < subs:
            cols[-1].append((cols[-1][0] + 1 + int(rng.integers(0, 3))) % 4)
        if rng.random() < inss:
            cols.append([int(rng.integers(0, 4)), "ins"])
    rows, col_rows = [], []
    for c, members in enumerate(cols):
        is_ins = members[-1] == "ins"
        letters = [m for m in members if m != "ins"]
        here = []
        for letter in letters:
            rows.append({"letter": letter, "preds": [], "col": c + 1, "ins": is_ins})
            here.append(len(rows))                       # 1-based row
        col_rows.append(here)
    # edges: every row from every row of the previous non-insertion column and from a preceding insertion column; insertion rows from the column before
    for c in range(1, len(cols)):
        prev = c - 1
        for r in col_rows[c]:
            for p in col_rows[prev]:
                rows[r - 1]["preds"].append(p)
            if rows[col_rows[prev][0] - 1]["ins"] and prev >= 1:          # the insertion is optional: also straight from the column before it
                for p in col_rows[prev - 1]:
                    rows[r - 1]["preds"].append(p)
            if c >= 3 and rng.random() < skips:                            # a deletion seen before: skip two columns
                rows[r - 1]["preds"].append(col_rows[c - 3][0])
    return rows, len(cols)


def dag_dp(rows, b, band=None):
    n, m = len(rows), len(b)
    H = [[0] * (m + 1) for _ in range(n + 1)]
    F = [[NEG] * (m + 1) for _ in range(n + 1)]
    Ee = [[NEG] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        preds = rows[i - 1]["preds"] or [0]
        lo, hi = (1, m) if band is None else band(i)
        for j in range(1, m + 1):
            if j < lo or j > hi:
                continue                                  # outside: H = 0, F = E = -inf
            s = M if rows[i - 1]["letter"] == b[j - 1] else N
            d = max(H[p][j - 1] + s for p in preds)
            F[i][j] = max(max(H[p][j] + G, F[p][j] + E) for p in preds)
            Ee[i][j] = max(H[i][j - 1] + G, Ee[i][j - 1] + E)
            H[i][j] = max(0, d, F[i][j], Ee[i][j])
    return H, F, Ee


def dag_trace(rows, b, H, F, Ee, start):
    """spoa's order: diagonal (in-edges in order), vertical (F extension before opening), horizontal; affine runs followed inside F / E"""
    i, j = start
    out = []
    while H[i][j] != 0:
        preds = rows[i - 1]["preds"] or [0]
        s = M if rows[i - 1]["letter"] == b[j - 1] else N
        nxt, ext_up, ext_left = None, False, False
        for p in preds:
            if H[i][j] == H[p][j - 1] + s:
                nxt = (p, j - 1); break
        if nxt is None:
            for p in preds:
                if H[i][j] == F[p][j] + E:
                    nxt, ext_up = (p, j), True; break
                if H[i][j] == H[p][j] + G:
                    nxt = (p, j); break
        if nxt is None:
            if H[i][j] == Ee[i][j - 1] + E:
                nxt, ext_left = (i, j - 1), True
            elif H[i][j] == H[i][j - 1] + G:
                nxt = (i, j - 1)
        assert nxt is not None
        out.append((i if nxt[0] != i else -1, j if nxt[1] != j else -1))
        i, j = nxt
        if ext_left:
            while True:
                out.append((-1, j)); j -= 1
                if Ee[i][j] + E != Ee[i][j + 1]:
                    break
        elif ext_up:
            while i != 0:
                stop, np_ = False, 0
                for p in (rows[i - 1]["preds"] or [0]):
                    if F[i][j] == H[p][j] + G:
                        stop, np_ = True, p; break
                    if F[i][j] == F[p][j] + E:
                        np_ = p; break
                out.append((i, -1)); i = np_
                if stop:
                    break
    return out


def test_band_on_a_dag_is_exact_when_its_best_score_clears_the_bound():
    rng = np.random.default_rng(21)
    held = failed = 0
    for case in range(80):
        rows, C = random_near_chain_dag(rng, int(rng.integers(40, 80)), 0.15, 0.08, 0.1)
        for r in rows:                                    # the premise the kernel checks while it builds the rows' records
            assert all(rows[p - 1]["col"] < r["col"] for p in r["preds"])
        # a sequence along one path of the graph, cut at the 5' end, with a few errors
        path, c = [], 0
        by_col = {}
        for i, r in enumerate(rows):
            by_col.setdefault(r["col"], []).append(i)
        for col in sorted(by_col):
            if rows[by_col[col][0]]["ins"] and rng.random() < 0.7:
                continue
            path.append(rows[int(rng.choice(by_col[col]))]["letter"])
        b = mutate(rng, path[int(rng.integers(0, 8)):], float(rng.choice([0.0, 0.03, 0.08])))
        L = len(b)
        if L < 10:
            continue
        t = int(rng.choice([2, 4, 8, 12]))
        tau, s = L - t, C - L + t
        if C < tau:
            continue
        band = lambda i: (max(1, rows[i - 1]["col"] - s), min(L, rows[i - 1]["col"] + t))
        Hb, Fb, Eb = dag_dp(rows, b, band)
        sb, cb = best_cells(Hb)
        Hf, Ff, Ef = dag_dp(rows, b)
        sf, cf = best_cells(Hf)
        if sb < 5 * tau - 4:
            failed += 1
            assert sf >= sb                               # a failed certificate only means: run the full rows
            continue
        held += 1
        assert sf == sb and cf == cb, (case, sf, sb)
        # spoa takes the first best cell in ITS rank order; whichever it is, the traceback from it is the same
        for start in cf[:3]:
            assert dag_trace(rows, b, Hf, Ff, Ef, start) == dag_trace(rows, b, Hb, Fb, Eb, start), case
    assert held >= 25 and failed >= 3                     # both branches of the kernel's logic are exercised


def longest_paths(rows):
    """a[i] = nodes on the longest path ENDING at row i (i included), b[i] = nodes on the longest path AFTER row i (rows 1-based, in
    topological order)."""
    n = len(rows)
    a, b = [0] * (n + 1), [0] * (n + 2)
    for i in range(1, n + 1):
        a[i] = 1 + max((a[p] for p in rows[i - 1]["preds"]), default=0)
    for i in range(n, 0, -1):
        for p in rows[i - 1]["preds"]:
            b[p] = max(b[p], b[i] + 1)
    return a, b


def add_ragged_ends(rng, rows, C, starts, ends):
    """What the POA #3 of many pack consensi looks like: alternative START chains that join the backbone a few
    columns in and alternative END chains that leave it a few columns before its end.  Every such chain takes columns of its own in the
    block order -- the MSA has 10-20 columns more per chain -- while no path visits two of them: the longest path barely grows."""
    by_col = {}
    for i, r in enumerate(rows):
        by_col.setdefault(r["col"], []).append(i + 1)
    out = [dict(r, preds=list(r["preds"])) for r in rows]
    # rebuild in block order with the new chains spliced in: a start chain right before the column it joins, an end chain at the very end
    joins = sorted(int(rng.integers(2, 8)) for _ in range(starts))
    leaves = sorted(C - int(rng.integers(2, 8)) for _ in range(ends))
    new_rows, remap, col = [], {}, 0
    cols_sorted = sorted(by_col)
    for c in cols_sorted:
        for jn in [j for j in joins if j == c]:
            prev = 0
            for _ in range(int(rng.integers(4, 12))):
                col += 1
                new_rows.append({"letter": int(rng.integers(0, 4)), "preds": [prev] if prev else [], "col": col, "ins": False})
                prev = len(new_rows)
            remap.setdefault(("join", c), []).append(prev)
        col += 1
        for r in by_col[c]:
            new_rows.append({"letter": out[r - 1]["letter"], "preds": [("old", p) for p in out[r - 1]["preds"]] + remap.get(("join", c), []), "col": col, "ins": out[r - 1]["ins"]})
            remap[("old", r)] = len(new_rows)
    for lv in leaves:
        prev = remap[("old", by_col[lv][0])]
        for _ in range(int(rng.integers(4, 12))):
            col += 1
            new_rows.append({"letter": int(rng.integers(0, 4)), "preds": [prev], "col": col, "ins": False})
            prev = len(new_rows)
    for r in new_rows:
        r["preds"] = [remap[p] if isinstance(p, tuple) else p for p in r["preds"]]
    return new_rows, col


def test_band_by_longest_paths_is_exact_and_narrower_than_the_band_by_columns():
    """The same lemma with the tightest bounds it admits (not built into the kernel: DESIGN.md section 8 item 2): a path through cell (i, j) has
    at most min(a_i, j) + min(b_i, L - j) diagonal moves, a_i / b_i the longest paths ending at / leaving row i.  The band
    L - t - b_i <= j <= a_i + t is exact under the same certificate (best score inside >= 5 (L - t) - 4) -- same best cells, same traceback --
    and on graphs with ragged start / end chains it is narrower than the band by MSA columns, which pays for every chain's own columns."""
    rng = np.random.default_rng(33)
    held = failed = narrower = 0
    for case in range(60):
        rows, C = random_near_chain_dag(rng, int(rng.integers(40, 70)), 0.12, 0.05, 0.08)
        rows, C = add_ragged_ends(rng, rows, C, int(rng.integers(2, 6)), int(rng.integers(2, 6)))
        for r in rows:
            assert all(rows[p - 1]["col"] < r["col"] for p in r["preds"])
        a, bb = longest_paths(rows)
        P = max(a)
        assert P < C                                       # the chains cost columns, not path length
        # a sequence along ONE longest path, cut at the 5' end, with a few errors
        i, path = a.index(P), []
        while i:
            path.append(rows[i - 1]["letter"])
            i = next((p for p in rows[i - 1]["preds"] if a[p] == a[i] - 1), 0)
        path.reverse()
        b = mutate(rng, path[int(rng.integers(0, 6)):], float(rng.choice([0.0, 0.03, 0.06])))
        L = len(b)
        if L < 10:
            continue
        t = int(rng.choice([2, 4, 8]))
        tau = L - t
        band = lambda i: (max(1, tau - bb[i]), min(L, a[i] + t))
        width_paths = max(hi - lo + 1 for lo, hi in (band(i) for i in range(1, len(rows) + 1)) if hi >= lo)
        width_cols = C - L + 2 * t + 1
        narrower += width_paths < width_cols
        Hb, Fb, Eb = dag_dp(rows, b, band)
        sb, cb = best_cells(Hb)
        Hf, Ff, Ef = dag_dp(rows, b)
        sf, cf = best_cells(Hf)
        if sb < 5 * tau - 4:
            failed += 1
            assert sf >= sb
            continue
        held += 1
        assert sf == sb and cf == cb, (case, sf, sb)
        for start in cf[:3]:
            assert dag_trace(rows, b, Hf, Ff, Ef, start) == dag_trace(rows, b, Hb, Fb, Eb, start), case
    assert held >= 20 and narrower >= 40, (held, failed, narrower)
