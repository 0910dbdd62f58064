"""`cluster` of N mixed-length --rna reads (config 5's first phase) with the package of ANOTHER checkout of this repository: the bisect of the
22.8 -> 33.8 s change between round 3d and round 4.  usage: cluster_mixed_tree.py ROOT READS"""
import os
import sys
import time
root = os.path.abspath(sys.argv[1]); n = int(sys.argv[2])
sys.path.insert(0, root)
os.environ["RATTLE_HIP_LIB"] = os.path.join(root, "rattle_amd", "csrc", "librattle_hip.so")
import numpy as np
from rattle_amd import synth
from rattle_amd.api import Context
tx = synth.mixed_transcriptome(max(50, n // 50), seed=5)
cat, qcat, off, tid, flip = synth.reads_packed(n, 0, 1, False, seed=6, tx=tx, chunk=50)
ctx = Context(0)
if hasattr(ctx, "stage_reads"):
    ctx.stage_reads(cat, qcat, off)
t0 = time.time(); cl = ctx.cluster_unsorted_packed(cat, off, is_rna=True); t1 = time.time()
print(os.path.basename(root), "reads", n, "bases", int(off[-1]), "cluster_s", round(t1 - t0, 2), "clusters", int(len(cl.main_id)),
      "counters", [int(x) for x in cl.counters[:2]], flush=True)
