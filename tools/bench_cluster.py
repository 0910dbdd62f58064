"""cluster-only timing at bench size: usage bench_cluster.py READS"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from rattle_amd import synth
from rattle_amd.api import Context
n = int(sys.argv[1])
cat, qcat, off, tid, _ = synth.reads_packed(n, max(5, n // 200), 1, True, seed=20260929, exon=(50, 210))
ctx = Context(0)
ctx.stage_reads(cat, None, off)
for it in range(3):
    ctx.reset_stats()
    t = time.time(); cl = ctx.cluster_unsorted_packed(cat, off); dt = time.time() - t
    print(f"iter {it}: {dt:.3f} s, {len(cl.main_id)} clusters, kernels ms:", [round(ctx.kernel_stats(k)[0]) for k in range(3)], "counters", cl.counters[:5].tolist(), flush=True)
