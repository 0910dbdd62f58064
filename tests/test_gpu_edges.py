"""Edge cases of the hot path through the C ABI, against the oracle."""
import numpy as np
import pytest

from rattle_amd import hps, synth
from rattle_amd.api import correct_command

pytestmark = pytest.mark.gpu


def test_empty_and_single_read_sets(gpu_ctx, oracle):
    gpu_ctx.load_reads([], 10, True)
    assert gpu_ctx.cluster_reads().as_list() == []
    one = [b"ACGTTGCAAGGCTTACGATCGATCGGATCCATGCAAGTCCATG" * 5]
    gpu_ctx.load_reads(one, 10, True)
    assert gpu_ctx.cluster_reads().as_list() == oracle.cluster_reads(one, k=10)[0] == [((0, 0, -1), [(0, 0, -1)])]


def test_a_big_correct_takes_the_room_of_the_cluster_index(gpu_ctx, oracle, monkeypatch):
    """A context that clustered its reads keeps their k-mer index (12-20 bytes per base); `correct` needs none of it and sizes its arena by
    the free memory, so beyond 24 GB the index is released at its entry (abi.hip; config 5 at 3e6 reads ran out of memory in stage 2 with
    72 GB of index beside the arena).  With the bound lowered to nothing: `correct` gives the oracle's bytes, the next cluster call says the
    reads are gone instead of reading freed memory, and after loading them again the clusters are the same."""
    from rattle_amd import synth
    from rattle_amd._lib import RattleError
    from rattle_amd.api import cluster_command, correct_command
    seqs, quals, _, _ = synth.reads(400, 4, 1, True, seed=5)
    order = sorted(range(len(seqs)), key=lambda i: -len(seqs[i]))
    reads = [seqs[i] for i in order]
    headers = [b"@r%d" % i for i in range(len(seqs))]
    clusters, _ = cluster_command(gpu_ctx, seqs, list(range(len(seqs))))
    gpu_ctx.load_reads(reads, 10, True)
    before = gpu_ctx.cluster_reads().as_list()
    monkeypatch.setenv("RATTLE_INDEX_KEEP_MB", "0")
    got = correct_command(gpu_ctx, headers, seqs, quals, clusters)
    monkeypatch.delenv("RATTLE_INDEX_KEEP_MB")
    from rattle_amd import hps
    want = oracle.correct(headers, seqs, quals, hps.encode(clusters))
    assert (got[0], got[1], got[2]) == (want[0], want[1], want[2])
    with pytest.raises(RattleError, match="no reads loaded"):
        gpu_ctx.cluster_reads()
    gpu_ctx.load_reads(reads, 10, True)
    assert gpu_ctx.cluster_reads().as_list() == before


def test_correct_gives_up_its_cached_arena_when_a_stage_needs_the_room():
    """`correct` keeps the POA arena between stages (85 % of what was free when a pass began); the stages' own buffers -- MSA rows, the
    compacted corrected reads -- grow with the job, and at 5e6 mixed reads they no longer fitted beside it.  correct_driver.hip: make_room
    frees the idle arena before a large allocation that would not fit; the next POA pass allocates one that does.  Here with
    RATTLE_MAKE_ROOM_ALWAYS=1 (a fresh process: the switch is read once): every stage re-allocates, the three outputs are the oracle's."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'oracle'))
import oracle as orc_mod
from rattle_amd import synth, hps
from rattle_amd.api import Context, cluster_command, correct_command
seqs, quals, _, _ = synth.reads(500, 4, 1, True, seed=12)
headers = [b'@r%%d' %% i for i in range(len(seqs))]
ctx = Context(0)
clusters, _ = cluster_command(ctx, seqs, list(range(len(seqs))))
got = correct_command(ctx, headers, seqs, quals, clusters, split=40)
want = orc_mod.Oracle().correct(headers, seqs, quals, hps.encode(clusters), split=40)
assert (got[0], got[1], got[2]) == (want[0], want[1], want[2])
print('same')
""" % (root, root)
    env = dict(os.environ, RATTLE_MAKE_ROOM_ALWAYS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("same"), r.stderr[-2000:]


def test_identical_reverse_and_short_reads(gpu_ctx, oracle):
    rng = np.random.default_rng(3)
    base = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 600)].tobytes()
    rc = base.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]
    reads = [base, base, rc, base[:300], rc[:200], b"ACGTACGTAC", b"ACGTACG", b"T" * 40]
    reads.sort(key=lambda s: -len(s))
    for is_rna in (False, True):
        gpu_ctx.load_reads(reads, 10, not is_rna)
        got = gpu_ctx.cluster_reads(is_rna=is_rna).as_list()
        want, _ = oracle.cluster_reads(reads, k=10, is_rna=is_rna)
        assert got == want
        if not is_rna:
            assert any(s[1] for _, mem in want for s in mem)          # the reverse-complement copies join on the reverse strand


def test_all_unrelated_reads_stay_singletons(gpu_ctx, oracle):
    rng = np.random.default_rng(11)
    reads = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(rng.integers(200, 500)))].tobytes() for _ in range(300)]
    reads.sort(key=lambda s: -len(s))
    gpu_ctx.load_reads(reads, 10, True)
    got = gpu_ctx.cluster_reads().as_list()
    want, _ = oracle.cluster_reads(reads, k=10)
    assert got == want and len(got) == 300


def test_polish_like_parameters_no_merge_pass(gpu_ctx, oracle):
    """-B == -b (main.cpp:669): the merge loop never runs (cluster.cpp:171-173)."""
    seqs, _, _, _ = synth.reads(300, 4, 1, False, seed=2)
    seqs.sort(key=lambda s: -len(s))
    gpu_ctx.load_reads(seqs, 6, False)
    got = gpu_ctx.cluster_reads(t_s=0.5, t_v=25.0, bv_threshold=0.4, min_bv_threshold=0.4, bv_falloff=0.05, is_rna=True).as_list()
    want, _ = oracle.cluster_reads(seqs, k=6, t_s=0.5, t_v=25.0, bvB=0.4, bvb=0.4, bvf=0.05, is_rna=True)
    assert got == want


def test_correct_small_clusters_and_u_bases(gpu_ctx, oracle):
    """Clusters at / below min_reads go to uncorrected untouched; RNA 'U' reads go through POA."""
    seqs, quals, _, _ = synth.reads(60, 2, 1, False, seed=9)
    seqs = [s.replace(b"T", b"U") for s in seqs]
    headers = [b"@u%d" % i for i in range(len(seqs))]
    clusters = [((0, 0, -1), [(i, 0, -1) for i in range(0, 5)]),            # 5 reads: not > min_reads
                ((5, 0, -1), [(i, 0, -1) for i in range(5, 11)]),           # 6 reads: corrected
                ((11, 0, -1), [(11, 0, -1)]),
                ((12, 0, -1), [(i, 0, -1) for i in range(12, 60)])]
    got = correct_command(gpu_ctx, headers, seqs, quals, clusters)
    want = oracle.correct(headers, seqs, quals, hps.encode(clusters))
    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
    assert got[1].count(b"@u") >= 6


def test_poa_unalignable_and_single_sequence_packs(gpu_ctx, oracle):
    packs = [[b"ACGTTGCA" * 20], [b"A" * 100, b"C" * 80, b"G" * 60], [b"ACGTTGCA" * 20, b"TGCAACGT" * 20, b"ACGTTGCA" * 19]]
    rows, width, _ = gpu_ctx.poa_msa(packs)
    for p, pack in enumerate(packs):
        want, _ = oracle.poa_msa(pack)
        assert rows[p] == want


def test_reads_longer_than_the_register_classes(gpu_ctx, oracle):
    """Reads beyond 6144 nt take the segmented int32 rows of kernel C (rare tail of a cDNA run); mixed with
    short packs in one call, and a pack whose first read alone exceeds the first-round node capacity."""
    rng = np.random.default_rng(17)
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def noisy(tx, n):
        out = []
        for _ in range(n):
            r = rng.random(len(tx))
            s = tx.copy()
            sub = (r >= 0.02) & (r < 0.05)
            s[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
            s = s[r >= 0.02]
            pos = np.sort(rng.integers(0, len(s) + 1, int(0.02 * len(s))))
            out.append(np.insert(s, pos, acgt[rng.integers(0, 4, len(pos))]).tobytes())
        return sorted(out, key=lambda x: -len(x))

    packs = [noisy(acgt[rng.integers(0, 4, 7000)], 5), noisy(acgt[rng.integers(0, 4, 900)], 12),
             noisy(acgt[rng.integers(0, 4, 12500)], 3)]
    rows, width, _ = gpu_ctx.poa_msa(packs)
    for p, pack in enumerate(packs):
        want, _ = oracle.poa_msa(pack)
        assert rows[p] == want, p


def test_reads_of_three_segments(gpu_ctx, oracle):
    """Reads beyond 16 384 nt: three segments of 8192 columns per row in the segmented int32 rows (config 5's long tail; the
    reference aligns whatever passes --upper-length, correct.cpp:395-405).  Byte for byte against the oracle's full matrices."""
    rng = np.random.default_rng(23)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    tx = acgt[rng.integers(0, 4, 18600)]
    pack = []
    for _ in range(3):
        r = rng.random(len(tx))
        s = tx.copy()
        sub = (r >= 0.01) & (r < 0.03)
        s[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        s = s[r >= 0.01]
        pos = np.sort(rng.integers(0, len(s) + 1, int(0.01 * len(s))))
        pack.append(np.insert(s, pos, acgt[rng.integers(0, 4, len(pos))]).tobytes())
    pack.sort(key=lambda x: -len(x))
    assert min(len(x) for x in pack) > 2 * 8192 + 1000
    rows, width, counters = gpu_ctx.poa_msa([pack])
    want, cells = oracle.poa_msa(pack)
    assert rows[0] == want
    assert int(counters[0]) == cells


def test_absurdly_long_sequence_is_an_error(gpu_ctx):
    from rattle_amd._lib import RattleError
    with pytest.raises(RattleError):
        gpu_ctx.poa_msa([[b"ACGT" * 300000, b"ACGT" * 300000]])      # 1.2 Mnt > POA_MAX_LEN
