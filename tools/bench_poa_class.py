"""POA-only throughput of one length class: PACKS copies of a few distinct 200-read packs of reads
around LEN nt, timed by the library's own HIP events.  usage: bench_poa_class.py LEN PACKS [DEPTH]"""
import sys, time
sys.path.insert(0, '/root/repo')
import ctypes as C
import numpy as np
from rattle_amd.api import Context, K_POA, MsaSet, _ptr, check, pack_reads

LEN = int(sys.argv[1]); PACKS = int(sys.argv[2]); DEPTH = int(sys.argv[3]) if len(sys.argv) > 3 else 200
ERR = float(sys.argv[4]) if len(sys.argv) > 4 else 0.10      # total error rate (del : sub : ins = 3 : 4 : 3)
rng = np.random.default_rng(5)
ACGT = np.frombuffer(b"ACGT", np.uint8)
distinct = []
for t in range(16):
    tx = ACGT[rng.integers(0, 4, LEN - 20 * t)]
    mem = []
    for _ in range(DEPTH):
        r = rng.random(len(tx)); keep = r >= 0.3 * ERR
        s = tx.copy(); sub = (r >= 0.3 * ERR) & (r < 0.7 * ERR); s[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
        out = []
        ins = rng.random(len(tx)) < 0.3 * ERR
        pieces = np.where(ins)[0]
        s2 = s[keep]
        # insertions: splice random bases at random places
        pos = np.sort(rng.integers(0, len(s2) + 1, int(ins.sum())))
        s3 = np.insert(s2, pos, ACGT[rng.integers(0, 4, len(pos))])
        mem.append(s3.tobytes())
    mem.sort(key=lambda x: -len(x))
    distinct.append(mem)
flat = [s for p in range(PACKS) for s in distinct[p % 16]]
cat, off = pack_reads(flat)
first = (np.arange(PACKS + 1) * DEPTH).astype(np.uint32)
ctx = Context(0)
for it in range(int(sys.argv[5]) if len(sys.argv) > 5 else 2):
    ctx.reset_stats()
    out = C.POINTER(MsaSet)()
    t = time.time()
    check(ctx.lib.rattle_hip_poa_msa(ctx.h, _ptr(cat, C.c_uint8), _ptr(off, C.c_uint64), len(flat), _ptr(first, C.c_uint32), PACKS, C.byref(out)))
    dt = time.time() - t
    cells = int(out.contents.counters[0]); rows = int(out.contents.counters[3]); alns = int(out.contents.counters[1])
    c = [int(out.contents.counters[i]) for i in range(8)]
    ctx.lib.rattle_hip_msa_set_free(out)
    if c[5]:
        tick = 1e-8
        print(f"  profile: ties {c[4] >> 32} sorted {c[4] & 0xffffffff} of {alns} | per row us: dp {c[5] * tick / rows * 1e6:.2f} | per aln ms: dp {c[5] * tick / alns * 1e3:.2f} tb {c[6] * tick / alns * 1e3:.2f} (ties {c[2] * tick / alns * 1e3:.2f}) add {c[7] * tick / alns * 1e3:.2f}")
    ms, launches, _ = ctx.kernel_stats(K_POA)
    print(f"iter {it}: maxlen {max(len(s) for s in flat)} packs {PACKS} kernel {ms:.0f} ms  {cells / ms / 1e6:.1f} GCUPS  rows/aln {rows / max(alns, 1):.0f}  wall {dt:.1f} s", flush=True)
