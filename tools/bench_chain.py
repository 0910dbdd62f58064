"""Near-chain POA passes (what stages 2a / 2b / 3a of `rattle correct` run): PACKS packs of DEPTH near-identical sequences around LEN nt
(5'-truncated by up to 10 %, ERR residual errors), timed by the library's own HIP events, with the exact band on and off.
usage: bench_chain.py LEN PACKS DEPTH [ERR] ; RATTLE_HIP_LIB=.../librattle_hip_prof.so + RATTLE_POA_PROFILE_JSON=<file> for the phase shares"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from rattle_amd.api import Context, K_POA, MsaSet, _ptr, check, pack_reads

LEN = int(sys.argv[1]); PACKS = int(sys.argv[2]); DEPTH = int(sys.argv[3]); ERR = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0005
rng = np.random.default_rng(9)
ACGT = np.frombuffer(b"ACGT", np.uint8)
distinct = []
for t in range(8):
    tx = ACGT[rng.integers(0, 4, LEN - 25 * t)]
    mem = []
    for _ in range(DEPTH):
        s = tx[int(rng.random() * 0.1 * len(tx)):].copy()
        hit = rng.random(len(s)) < ERR
        s[hit] = ACGT[rng.integers(0, 4, int(hit.sum()))]
        mem.append(s.tobytes())
    mem.sort(key=lambda x: -len(x))
    distinct.append(mem)
flat = [s for p in range(PACKS) for s in distinct[p % 8]]
cat, off = pack_reads(flat)
first = (np.arange(PACKS + 1) * DEPTH).astype(np.uint32)
ctx = Context(0)
for band in ("1", "0", "1", "0"):
    os.environ["RATTLE_POA_BAND"] = band
    ctx.reset_stats()
    out = C.POINTER(MsaSet)()
    t = time.time()
    check(ctx.lib.rattle_hip_poa_msa(ctx.h, _ptr(cat, C.c_uint8), _ptr(off, C.c_uint64), len(flat), _ptr(first, C.c_uint32), PACKS, C.byref(out)))
    dt = time.time() - t
    c = [int(x) for x in out.contents.counters]
    ctx.lib.rattle_hip_msa_set_free(out)
    ms = ctx.kernel_stats(K_POA)[0]
    print(f"band {band}: LEN {LEN} PACKS {PACKS} DEPTH {DEPTH}: kernel {ms:.1f} ms (call {dt * 1e3:.0f} ms), {ms / DEPTH * 1e3:.0f} us per alignment of a pack, "
          f"cells ref {c[0]:.3e} computed {c[4]:.3e}, certified {c[5]} failed {c[6]} of {c[1] - PACKS}, GCUPS(ref cells) {c[0] / ms / 1e6:.1f}", flush=True)
