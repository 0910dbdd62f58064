"""Time of the two-level `cluster --iso` flow on synthetic reads: usage bench_iso.py READS"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from rattle_amd import synth
from rattle_amd.api import Context, cluster_command
n = int(sys.argv[1])
t = time.time()
seqs, quals, tid, _ = synth.reads(n, max(5, n // 600), 3, True, seed=7, exon=(50, 210))
print("generated", n, "reads in", round(time.time() - t, 1), "s", flush=True)
ctx = Context(0)
for it in range(2):
    t = time.time()
    gene, _ = cluster_command(ctx, seqs, list(range(n)))
    t1 = time.time()
    iso, _ = cluster_command(ctx, seqs, list(range(n)), iso=True)
    t2 = time.time()
    print(f"iter {it}: gene level {t1 - t:.2f} s ({len(gene)} clusters)   --iso {t2 - t1:.2f} s ({len(iso)} clusters)", flush=True)
