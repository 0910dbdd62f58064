"""BASELINE configs[4] on one GPU: mixed-length --rna reads (150 .. 100 000 nt) through
cluster -> correct (with the DP budget rule) -> polish-style re-clustering of the consensi.
usage: run_mixed.py READS [MAX_READ_LEN_FOR_CORRECT]"""
import json
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rattle_amd import synth
from rattle_amd.api import Context, pack_reads
import ctypes
import resource
import threading
n = int(sys.argv[1])
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 9000


class HbmPeak(threading.Thread):
    """Polls hipMemGetInfo (device-wide) five times a second: the largest amount in use per phase."""
    def __init__(self):
        super().__init__(daemon=True)
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.phase, self.peak, self.total, self.stop = "start", {}, 0, False

    def run(self):
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        while not self.stop:
            if self.hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0:
                self.total = t.value
                self.peak[self.phase] = max(self.peak.get(self.phase, 0), t.value - f.value)
            time.sleep(0.2)
t = time.time()
tx = synth.mixed_transcriptome(max(50, n // 50), seed=5)
cat, qcat, off, tid, flip = synth.reads_packed(n, 0, 1, False, seed=6, tx=tx, chunk=50)
lens = np.diff(off.astype(np.int64))
print(f"generated {n} reads in {time.time() - t:.0f} s: len min/mean/median/max {lens.min()} {lens.mean():.0f} {int(np.median(lens))} {lens.max()}, > 6144 nt: {int((lens > 6144).sum())}, "
      f"> {cap} nt: {int((lens > cap).sum())}, bases {int(lens.sum()):.3e}", flush=True)
ctx = Context(0)
mem = HbmPeak(); mem.start()
ctx.stage_reads(cat, qcat, off)
mem.phase = "cluster"
t0 = time.time(); cl = ctx.cluster_unsorted_packed(cat, off, is_rna=True); t1 = time.time()
print(f"cluster: {t1 - t0:.1f} s, {len(cl.main_id)} clusters, HBM in use at most {mem.peak.get('cluster', 0) / 1e9:.1f} GB", flush=True)
mem.phase = "correct"
res = ctx.correct_packed(cat, qcat, off, cl, max_pack_cells=(6 * cap + 64) * cap, keep=True); t2 = time.time()
n_cor, n_unc, n_cons, counters = res.counts()
print(f"correct: {t2 - t1:.1f} s, {n_cor} corrected, {n_unc} uncorrected, {n_cons} consensi, HBM in use at most {mem.peak.get('correct', 0) / 1e9:.1f} GB", flush=True)
assert n_cor + n_unc == n
R = res.ptr.contents
cons = [R.consensi.seq[int(R.consensi.off[i]):int(R.consensi.off[i + 1])] for i in range(R.consensi.n)]
res.free()
# polish (main.cpp:612-762): consensi clustered with k=6, 0.5 / 25, no merge pass, corrected with min_reads = 0
order = sorted(range(len(cons)), key=lambda i: -len(cons[i]))
pc, po = pack_reads([cons[i] for i in order])
t3 = time.time()
mem.phase = "polish"
ctx.unstage_reads()
ctx.load_packed(pc, po, 6, False)
pcl = ctx.cluster_reads(t_s=0.5, t_v=25.0, bv_threshold=0.4, min_bv_threshold=0.4, bv_falloff=0.05, is_rna=True)
pq = np.full(len(pc), ord('K'), np.uint8)
pres = ctx.correct_packed(pc, pq, po, pcl, min_reads=0)
t4 = time.time()
mem.stop = True
out = {"hbm_in_use_peak_gb": {k: round(v / 1e9, 1) for k, v in mem.peak.items()}, "hbm_total_gb": round(mem.total / 1e9, 1),
       "host_rss_peak_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1),
       "config": f"{n} mixed-length --rna reads (log-uniform 150..8000 nt body + 1 % tail to 100 000 nt, mean {lens.mean():.0f} nt), cluster -> correct -> polish on one MI355X",
       "reads": n, "bases": int(lens.sum()), "max_read": int(lens.max()), "reads_over_6144": int((lens > 6144).sum()),
       "cluster_s": t1 - t0, "correct_s": t2 - t1, "polish_s": t4 - t3, "reads_per_s_cluster_correct": n / (t2 - t0),
       "clusters": int(len(cl.main_id)), "corrected": int(n_cor), "uncorrected": int(n_unc), "consensi": int(n_cons),
       "packs_queued": int(counters[2]), "packs_skipped": int(counters[3]), "reads_in_skipped_packs": int(counters[4]),
       "correct_budget": f"packs whose longest read exceeds {cap} nt are left uncorrected (max_pack_cells = (6 L + 64) L)",
       "dp_cells": int(counters[0]), "polish_clusters": int(len(pcl.main_id)), "transcriptome_records": int(pres[2]),
       "cluster_counters": {"bv_pair_tests": int(cl.counters[0]), "full_comparisons": int(cl.counters[1])}}
print(json.dumps(out))
