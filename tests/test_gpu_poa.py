"""Parity of the HIP POA kernel (kernel C) with the oracle's spoa restatement: MSA rows must be
byte-identical (same width, same column for every base)."""
import numpy as np
import pytest

from rattle_amd import synth

pytestmark = pytest.mark.gpu


def _packs_from_synth(n, genes, seed, both=False, max_pack=40):
    seqs, _, tid, _ = synth.reads(n, genes, 1, both, seed=seed)
    packs = []
    for g in range(genes):
        mem = [seqs[i] for i in range(n) if tid[i] == g]
        mem.sort(key=lambda s: -len(s))
        if len(mem) >= 2:
            packs.append(mem[:max_pack])
    return packs


def test_tiny_hand_cases(gpu_ctx, oracle):
    packs = [
        [b"ACGTACGTAC"],                                               # single sequence
        [b"ACGTACGTACGTTTGACA", b"ACGTACGTACGTTTGACA"],                # identical
        [b"ACGTACGTACGTTTGACA", b"ACGTACCTACGTTGACA", b"TTTTTTTT"],    # mismatch, deletion, unalignable
        [b"AAAAAAAAAACCCCCCCCCC", b"GGGGGGGGGG", b"AAAAAAAAAAGGGGGGGGGGCCCCCCCCCC"],
        [b"ACGT" * 30, b"ACGT" * 28 + b"AC", b"CGT" + b"ACGT" * 29, b"ACGA" * 30],
    ]
    rows, width, counters = gpu_ctx.poa_msa(packs)
    for p, pack in enumerate(packs):
        want, _ = oracle.poa_msa(pack)
        assert rows[p] == want, p
        assert width[p] == len(want[0])


def test_synthetic_packs_match_oracle(gpu_ctx, oracle):
    packs = _packs_from_synth(400, 10, seed=5)
    assert len(packs) >= 8
    rows, width, counters = gpu_ctx.poa_msa(packs)
    cells = 0
    for p, pack in enumerate(packs):
        want, c = oracle.poa_msa(pack)
        cells += c
        assert width[p] == len(want[0]), p
        assert rows[p] == want, p
    assert int(counters[0]) == cells                      # exact DP cell count


def test_toyset_clusters_match_oracle(gpu_ctx, oracle, toyset, toyset_clusters):
    """Real reads (long, noisy, up to 4.5 kb): clusters of 6..40 reads from the toyset."""
    cids = [c for c, (m, mem) in enumerate(toyset_clusters) if 6 <= len(mem) <= 40][:12]
    packs = [[toyset[s[0]][1] for s in toyset_clusters[c][1]] for c in cids]
    rows, width, _ = gpu_ctx.poa_msa(packs)
    for p, pack in enumerate(packs):
        want, _ = oracle.poa_msa(pack)
        assert rows[p] == want, cids[p]


def test_multi_segment_rows(gpu_ctx, oracle):
    """Sequences longer than one 1024-column segment and not a multiple of 16."""
    rng = np.random.default_rng(2)
    base = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 2500)]
    pack = []
    for i in range(6):
        s = base.copy()
        idx = rng.integers(0, len(s), 120)
        s[idx] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 120)]
        cut = int(rng.integers(0, 200))
        pack.append(s[cut:len(s) - int(rng.integers(0, 37))].tobytes())
    rows, width, _ = gpu_ctx.poa_msa([pack])
    want, _ = oracle.poa_msa(pack)
    assert rows[0] == want


@pytest.mark.parametrize("length,depth", [(700, 40), (1300, 30), (1800, 25), (2300, 20), (2900, 16), (4500, 10), (7000, 6)])
def test_length_classes_match_oracle(gpu_ctx, oracle, length, depth):
    """One pack per column class of kernel C (1024 / 1536 / 2048 / 2560 packed int16 rows, 4096 / 6144 / 8192 32-bit rows):
    noisy copies (sub / ins / del) of one random transcript, deep enough for bubbles, ties and far predecessors."""
    rng = np.random.default_rng(length)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    tx = acgt[rng.integers(0, 4, length)]
    pack = []
    for _ in range(depth):
        r = rng.random(len(tx))
        s = tx.copy()
        sub = (r >= 0.03) & (r < 0.06)
        s[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        s = s[r >= 0.03]
        pos = np.sort(rng.integers(0, len(s) + 1, int(0.03 * len(s))))
        s = np.insert(s, pos, acgt[rng.integers(0, 4, len(pos))])
        pack.append(s[int(rng.integers(0, 30)):].tobytes())
    pack.sort(key=lambda x: -len(x))
    rows, width, counters = gpu_ctx.poa_msa([pack, pack[::-1]])
    for got, p in zip(rows, (pack, pack[::-1])):
        want, _ = oracle.poa_msa(p)
        assert got == want


@pytest.mark.parametrize("mode", ["dense", "mt4", "mt2", "mt1"])
@pytest.mark.parametrize("length,depth", [(700, 40), (1300, 30), (1800, 25), (2300, 20)])
def test_packed_classes_in_every_form_of_the_row_loop(gpu_ctx, oracle, monkeypatch, length, depth, mode):
    """The four packed column classes under every form of the row loop (RATTLE_POA_MODE): barrier + record-word ring, and the teams of
    wavefronts with the ready-made ring (4 / 2 / 1 teams: rows that do not depend on each other run at the same time, poa.hip
    dp_rows_mt; one team = the lean pipeline without a row barrier) -- all byte-identical to the oracle."""
    monkeypatch.setenv("RATTLE_POA_MODE", mode)
    rng = np.random.default_rng(length + 1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    tx = acgt[rng.integers(0, 4, length)]
    pack = []
    for _ in range(depth):
        r = rng.random(len(tx))
        s = tx.copy()
        sub = (r >= 0.03) & (r < 0.07)
        s[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        s = s[r >= 0.03]
        pos = np.sort(rng.integers(0, len(s) + 1, int(0.03 * len(s))))
        s = np.insert(s, pos, acgt[rng.integers(0, 4, len(pos))])
        pack.append(s[int(rng.integers(0, 30)):].tobytes())
    pack.sort(key=lambda x: -len(x))
    rows, width, counters = gpu_ctx.poa_msa([pack, pack[::-1], pack[: depth // 2]])
    for got, p in zip(rows, (pack, pack[::-1], pack[: depth // 2])):
        want, _ = oracle.poa_msa(p)
        assert got == want


@pytest.mark.parametrize("env", [{"RATTLE_POA_DEBUG": "1"}, {"RATTLE_POA_DEBUG": "2"}, {"RATTLE_POA_DEBUG": "3"}, {"RATTLE_POA_DEBUG": "4"},
                                 {"RATTLE_POA_DEBUG": "4", "RATTLE_POA_MODE": "dense"},
                                 {"RATTLE_POA_NODE_CAP": "700"}, {"RATTLE_POA_MODE": "dense"},
                                 {"RATTLE_POA_MODE": "mt4"}, {"RATTLE_POA_MODE": "mt2"}, {"RATTLE_POA_MODE": "mt1"},
                                 {"RATTLE_POA_MODE": "mt4", "RATTLE_POA_DEBUG": "3"}, {"RATTLE_POA_MODE": "mt2", "RATTLE_POA_NODE_CAP": "700"},
                                 {"RATTLE_POA_MODE": "mt4", "RATTLE_POA_MT_SLOTS": "11"}, {"RATTLE_POA_MODE": "mt2", "RATTLE_POA_MT_SLOTS": "6"},
                                 {"RATTLE_POA_MODE": "mt1", "RATTLE_POA_MT_SLOTS": "3"}])
def test_fallback_paths_match_oracle(gpu_ctx, oracle, monkeypatch, env):
    """The slow paths behind the fast ones stay exact: full topological sort for ties (bit 0), traceback without
    the LDS chain (bit 1), packs re-run with a larger arena after a node-capacity overflow; and every form of the row loop
    (RATTLE_POA_MODE: the barrier form, teams of wavefronts on 4 / 2 / 1 teams -- the last with
    rings so short that many predecessors come from the record in HBM)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    packs = _packs_from_synth(400, 10, seed=11)
    rows, width, counters = gpu_ctx.poa_msa(packs)
    for p, pack in enumerate(packs):
        want, _ = oracle.poa_msa(pack)
        assert rows[p] == want, p


@pytest.mark.parametrize("env", [{}, {"RATTLE_POA_MODE": "dense"}, {"RATTLE_POA_MODE": "mt4"},
                                 {"RATTLE_POA_MODE": "mt2"}, {"RATTLE_POA_MODE": "mt1"}, {"RATTLE_POA_MODE": "mt4", "RATTLE_POA_MT_SLOTS": "11"}])
def test_predecessors_hundreds_of_rows_back_and_many_in_edges(gpu_ctx, oracle, monkeypatch, env):
    """The row loop reads a COMPACT plan record: the distances to a row's first eight predecessor rows in a byte each, saturated
    at 255, the in-degree capped at 255 (poa.hip, round 4).  Reads that skip 300-600 bases of the others (an exon left out) give
    nodes whose predecessor lies far more than 255 rows back; reads that resume at many different places give one node more than
    eight in-edges (the edge-list walk); both next to ordinary rows, in the barrier form and the team forms of the loop."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    tx = acgt[rng.integers(0, 4, 1400)]

    def noisy(a, err=0.06):
        r = rng.random(len(a))
        b = a.copy()
        sub = r < err * 0.4
        b[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        return b[(r >= err * 0.7) | (r < err * 0.4)]          # a few deletions too

    pack = [noisy(tx).tobytes() for _ in range(10)]
    # exon skipping: 300 .. 620 bases left out at different places
    for a, n in ((200, 300), (450, 620), (800, 410), (150, 505)):
        pack.append(noisy(np.concatenate([tx[:a], tx[a + n:]])).tobytes())
    # many different resume points into the same downstream node: prefixes of different lengths glued to the common tail from 1000 on
    for cut in range(300, 960, 55):
        pack.append(noisy(np.concatenate([tx[:cut], tx[1000:]]), 0.03).tobytes())
    pack.sort(key=lambda s: -len(s))
    rows, width, counters = gpu_ctx.poa_msa([pack, pack[::-1]])
    for got, p in zip(rows, (pack, pack[::-1])):
        want, _ = oracle.poa_msa(p)
        assert got == want


# ---- the exact band for near-chain graphs (dp_rows_band; RATTLE_POA_BAND=1 turns it on for this entry point) ----
def _mutate(rng, s, sub=0.0, ins=0.0, dele=0.0):
    out = bytearray()
    for c in s:
        r = rng.random()
        if r < dele:
            continue
        out.append(int(rng.choice(list(b"ACGT"))) if r < dele + sub else c)
        if rng.random() < ins:
            out.append(int(rng.choice(list(b"ACGT"))))
    return bytes(out)


def _near_identical_pack(rng, length, depth, err, trunc=0.10):
    base = bytes(rng.choice(list(b"ACGT"), length).astype(np.uint8))
    pack = []
    for _ in range(depth):
        cut = int(rng.random() * trunc * length)
        pack.append(_mutate(rng, base[cut:], err * 0.4, err * 0.3, err * 0.3))
    pack.sort(key=lambda s: -len(s))
    return pack


@pytest.mark.parametrize("length,err", [(700, 0.0), (1000, 0.004), (1450, 0.01), (2300, 0.003), (520, 0.02)])
def test_band_on_near_identical_packs_matches_oracle(gpu_ctx, oracle, monkeypatch, length, err):
    """What POA #2 / #3 of `rattle correct` see (correct.cpp:427-436,520-532): near-identical sequences, 5'-truncated.  With the band
    on, the MSA is byte-identical to the oracle's full matrices, most alignments are certified, and the device computes a fraction
    of the reference's cells."""
    monkeypatch.setenv("RATTLE_POA_BAND", "1")
    rng = np.random.default_rng(length)
    packs = [_near_identical_pack(rng, length, 24, err) for _ in range(3)]
    rows, width, counters = gpu_ctx.poa_msa(packs)
    cells = 0
    for p, pack in enumerate(packs):
        want, c = oracle.poa_msa(pack)
        cells += c
        assert rows[p] == want, p
    assert int(counters[0]) == cells                      # the reference's count, whatever was computed
    n_aln = sum(len(p) - 1 for p in packs)
    # alignments with a certified band: nearly all while the graph stays a chain; with more errors the graph outgrows the band (its
    # width is rows - columns + 2 t + 1: every bubble node counts) and the full rows take over -- never a wrong answer, never a failed call
    # (a pack with ONE alignment that has no certified band is run again over the full rows, from the start: its computed cells then exceed the reference's)
    if err <= 0.004 and length < 2000:
        assert int(counters[5]) >= 0.8 * n_aln, counters
        assert int(counters[4]) < 0.5 * cells, counters   # cells really computed
    else:
        assert int(counters[5]) >= 1, counters


def test_band_off_computes_every_cell(gpu_ctx, oracle, monkeypatch):
    monkeypatch.setenv("RATTLE_POA_BAND", "0")
    rng = np.random.default_rng(3)
    packs = [_near_identical_pack(rng, 800, 12, 0.003)]
    rows, _, counters = gpu_ctx.poa_msa(packs)
    want, c = oracle.poa_msa(packs[0])
    assert rows[0] == want
    assert int(counters[4]) == c == int(counters[0]) and int(counters[5]) == 0 and int(counters[6]) == 0


def _band_failure_packs():
    rng = np.random.default_rng(77)
    base = bytes(rng.choice(list(b"ACGT"), 1200).astype(np.uint8))
    other = bytes(rng.choice(list(b"ACGT"), 1150).astype(np.uint8))
    cases = {}
    # an indel longer than any band: the alignment jumps by 300 columns
    cases["long_deletion"] = [base, base, base[:500] + base[800:], base[:450] + base[700:], base]
    cases["long_insertion"] = [base, base[:600] + other[:260] + base[600:], base, base[:600] + other[:260] + base[600:]]
    # a sequence that has nothing to do with the pack: score far below 5 (L - t)
    cases["unrelated"] = [base, base, other, base[30:], other[10:]]
    # a bubble right at the band's edge: the same insertion in many members widens the graph until n - L eats the room
    ins = [base[:300] + other[k * 40:k * 40 + 38] + base[300:] for k in range(6)]
    cases["growing_graph"] = [base] + ins + [base[40:], base[80:], base]
    # two rows that tie for the best score: a tandem duplication (the end of the sequence matches two places equally well)
    rep = base[:400] + base[300:400] + base[400:900]
    cases["tied_rows"] = [rep, rep, base[:400] + base[400:900], base[:400], rep[50:]]
    # everything shifted to the far corner of the band: maximal 5' truncation against a long first read
    cases["band_corner"] = [base, base[118:], base[119:], base[120:], base[:1080], base[60:1100]]
    return cases


@pytest.mark.parametrize("strips", [False, True])
@pytest.mark.parametrize("name", sorted(_band_failure_packs()))
def test_band_certificate_failures_fall_back_to_the_full_rows(gpu_ctx, oracle, monkeypatch, name, strips):
    """Where the certificate must fail (or the band cannot be tried at all) the full rows run and the result is the oracle's: over the
    full-row kernels after the pack was handed back (strips = False: such a young pack has not earned the strips), or in place as
    vertical strips on the band's row loop (strips = True)."""
    monkeypatch.setenv("RATTLE_POA_BAND", "1")
    if strips:
        monkeypatch.setenv("RATTLE_POA_DEBUG", "8")
    pack = _band_failure_packs()[name]
    rows, _, counters = gpu_ctx.poa_msa([pack])
    want, c = oracle.poa_msa(pack)
    assert rows[0] == want, name
    assert int(counters[0]) == c
    if name == "unrelated":
        assert int(counters[6]) >= 1, counters            # a failed certificate ...
    if name in ("long_deletion", "long_insertion", "unrelated") and not strips:
        assert int(counters[4]) >= int(counters[0])       # ... the pack lost its band at some alignment and was run again over the full rows


@pytest.mark.parametrize("mode", ["band", "band+strips", "dense", "mt2"])
def test_band_noisy_packs_every_form(gpu_ctx, oracle, monkeypatch, mode):
    """Noisy reads (POA #1's input): the band cannot hold (graphs of 4-5 nodes per base) and must never be wrong about that.  `band+strips`:
    every alignment that gets no band runs the full rows as vertical strips of 512 columns on the band's own row loop (RATTLE_POA_DEBUG bit 3:
    also for packs that never had a band) -- the path a near-chain pack takes for the odd alignment that does not fit."""
    if mode == "band+strips":
        mode = "band"
        monkeypatch.setenv("RATTLE_POA_DEBUG", "8")
    monkeypatch.setenv("RATTLE_POA_MODE", mode)
    monkeypatch.setenv("RATTLE_POA_BAND", "1")
    packs = _packs_from_synth(300, 8, seed=23, max_pack=24)
    rows, _, counters = gpu_ctx.poa_msa(packs)
    for p, pack in enumerate(packs):
        want, _ = oracle.poa_msa(pack)
        assert rows[p] == want, p
