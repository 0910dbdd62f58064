"""Mixed-length robustness run: usage run_mixed.py READS  (transcripts of 8 exons of 20..1200 nt)"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from rattle_amd import synth
from rattle_amd.api import Context
n = int(sys.argv[1])
cat, qcat, off, tid, flip = synth.reads_packed(n, max(5, n // 100), 1, True, seed=3, exon=(20, 1200))
lens = np.diff(off.astype(np.int64))
print("reads", n, "len min/median/max", lens.min(), int(np.median(lens)), lens.max(), "> 6144:", int((lens > 6144).sum()), flush=True)
ctx = Context(0)
t = time.time(); cl = ctx.cluster_unsorted_packed(cat, off); t1 = time.time()
res = ctx.correct_packed(cat, qcat, off, cl); t2 = time.time()
print(f"cluster {t1 - t:.2f} s ({len(cl.main_id)} clusters), correct {t2 - t1:.2f} s -> corrected {res[0]} uncorrected {res[1]} consensi {res[2]}, DP cells {int(res[3][0]):.3e}")
