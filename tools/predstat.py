"""Predecessor statistics of the POA row plans (measurement build -DPOA_PREDSTAT): in-edges per row, share beyond the ring,
share from the previous row.  usage: RATTLE_HIP_LIB=.../librattle_hip_predstat.so python tools/predstat.py LEN PACKS"""
import sys, subprocess, os, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from rattle_amd.api import Context, MsaSet, _ptr, check, pack_reads
LEN = int(sys.argv[1]); PACKS = int(sys.argv[2]); DEPTH = 200; ERR = 0.10
rng = np.random.default_rng(5)
ACGT = np.frombuffer(b"ACGT", np.uint8)
distinct = []
for t in range(4):
    tx = ACGT[rng.integers(0, 4, LEN - 20 * t)]
    mem = []
    for _ in range(DEPTH):
        r = rng.random(len(tx)); keep = r >= 0.3 * ERR
        s = tx.copy(); sub = (r >= 0.3 * ERR) & (r < 0.7 * ERR); s[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
        ins = rng.random(len(tx)) < 0.3 * ERR
        s2 = s[keep]
        pos = np.sort(rng.integers(0, len(s2) + 1, int(ins.sum())))
        mem.append(np.insert(s2, pos, ACGT[rng.integers(0, 4, len(pos))]).tobytes())
    mem.sort(key=lambda x: -len(x))
    distinct.append(mem)
flat = [s for p in range(PACKS) for s in distinct[p % 4]]
cat, off = pack_reads(flat)
first = (np.arange(PACKS + 1) * DEPTH).astype(np.uint32)
ctx = Context(0)
out = C.POINTER(MsaSet)()
check(ctx.lib.rattle_hip_poa_msa(ctx.h, _ptr(cat, C.c_uint8), _ptr(off, C.c_uint64), len(flat), _ptr(first, C.c_uint32), PACKS, C.byref(out)))
c = [int(out.contents.counters[i]) for i in range(8)]
rows = c[3]
print(f"len {LEN}: rows {rows}, in-edges per row {c[4] / rows:.3f}, beyond the ring {c[5] / c[4]:.3f} of the in-edges, from the previous row {c[6] / c[4]:.3f}")
