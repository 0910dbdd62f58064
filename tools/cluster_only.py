"""Gene-level `rattle cluster` alone on the bench's synthetic batch: wall time per run (RATTLE_TIMING=1 adds the driver's split).
usage: python tools/cluster_only.py [reads] [runs]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from rattle_amd.api import Context

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cat, qcat, off, tid, _ = bench.make_workload(reads, max(5, reads // 200), seed=20260929)
ctx = Context(0)
ctx.stage_reads(cat, qcat, off)
for i in range(runs):
    t0 = time.time()
    cl = ctx.cluster_unsorted_packed(cat, off)
    print(f"run {i}: {1e3 * (time.time() - t0):.1f} ms, {len(cl.main_id)} clusters, digest {bench.cluster_digest(cl)}, rounds {int(cl.counters[3])}, launches {int(cl.counters[4])}", flush=True)
