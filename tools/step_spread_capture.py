"""The bench's step (cluster, then correct, results of the step before still held) repeated in one process with RATTLE_POA_TIMELINE set:
per step the wall times and the stage times, so that a step with the intermittent slow stage 1 can be matched with
its timeline (tools/timeline_summary.py: pass 4 k is stage 1 of step k).  usage: RATTLE_POA_TIMELINE=/tmp/tl.txt python tools/step_spread_capture.py [reads] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from rattle_amd.api import Context

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cat, qcat, off, tid, _ = bench.make_workload(reads, max(5, reads // 200), seed=20260929)
ctx = Context(0)
ctx.stage_reads(cat, qcat, off)
held = None
for i in range(steps):
    ctx.stage_ms(True)
    t0 = time.time()
    cl = ctx.cluster_unsorted_packed(cat, off)
    t1 = time.time()
    h = ctx.correct_packed(cat, qcat, off, cl, keep=True)
    t2 = time.time()
    ms = ctx.stage_ms(True)
    if held is not None:
        held.free()
    held = h
    print(f"step {i}: {1e3 * (t2 - t0):.0f} ms = cluster {1e3 * (t1 - t0):.0f} + correct {1e3 * (t2 - t1):.0f} (stages { {k: round(v) for k, v in ms.items()} })", flush=True)
