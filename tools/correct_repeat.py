"""`correct` repeated in ONE process on the bench's batch (same arena, same clusters): wall time of every run, to separate
run-to-run variation inside a process from variation between processes / boxes.  usage: python tools/correct_repeat.py [reads] [runs]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from rattle_amd.api import Context

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cat, qcat, off, tid, _ = bench.make_workload(reads, max(5, reads // 200), seed=20260929)
ctx = Context(0)
ctx.stage_reads(cat, qcat, off)
cl = ctx.cluster_unsorted_packed(cat, off)
for i in range(runs):
    t0 = time.time()
    h = ctx.correct_packed(cat, qcat, off, cl, keep=True)
    dt = time.time() - t0
    print(f"run {i}: {1e3 * dt:.0f} ms digest {h.digest()}", flush=True)
    h.free()
