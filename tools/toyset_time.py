"""`cluster --rna` + `correct` of the reference's toyset (tests/golden) through the HIP path, three times; RATTLE_TIMING=1 for the stages.
usage: toyset_time.py"""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rattle_amd.api import Context, pack_reads
lines = gzip.open(os.path.join(ROOT, "tests", "golden", "toyset_rna.fastq.gz"), "rb").read().split(b"\n")
seqs, quals = lines[1::4], lines[3::4]
seqs = [s for s in seqs if s]; quals = quals[:len(seqs)]
cat, off = pack_reads(seqs)
qcat = np.frombuffer(b"".join(quals), np.uint8).copy()
ctx = Context(0)
for it in range(3):
    t0 = time.time()
    cl = ctx.cluster_unsorted_packed(cat, off, k=10, is_rna=True)
    t1 = time.time()
    res = ctx.correct_packed(cat, qcat, off, cl, vote_order=b"U-GTAC")
    t2 = time.time()
    print(f"toyset pass {it}: cluster {t1 - t0:.3f} s, correct {t2 - t1:.3f} s, counters {[int(x) for x in res[3]]}", flush=True)
    sys.stderr.write(f"==== end of pass {it}\n")
