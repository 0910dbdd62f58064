"""SURVEY 8(d), reference CPU path timing, step (1): the REAL reference translation units (kmer.cpp, similarity.cpp, utils.cpp
compiled in place into oracle/_ref/libref.so) timed beside the restatement (liboracle.so) on the same inputs, in the build
container (the reference sources do not travel to the GPU box).  cluster.cpp / correct.cpp do not compile here (absent hps /
spoa headers), so the comparison covers the units the cluster path spends its time in: k-mer extraction + bit-vectors
(extract_kmers_from_read) and the full pair comparison (get_common_kmers + calc_similarity + var).
usage: python tools/ref_vs_port.py [N_READS] [N_PAIRS] -> one JSON line (kept under profiles/)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc_mod  # noqa: E402
from rattle_amd import synth  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
orc_mod.build(ref=True)
port, ref = orc_mod.Oracle(), orc_mod.Ref()
seqs, _, tid, _ = synth.reads(n_reads, 10, 1, True, seed=3, exon=(50, 210))
rng = np.random.default_rng(0)
same = [(i, j) for i in range(n_reads) for j in range(i + 1, min(i + 40, n_reads)) if tid[i] == tid[j]][:n_pairs // 2]
diff = [(int(a), int(b)) for a, b in rng.integers(0, n_reads, (n_pairs, 2)) if tid[a] != tid[b]][:n_pairs - len(same)]
pairs = same + diff
out = {"host": "build container", "cores_used": 1, "reads": n_reads, "pairs": len(pairs), "related_pairs": len(same),
       "workload": "synthetic cDNA reads (mean 1 kb, 10 % error), k = 10, both strands"}
for name, lib in (("reference", ref), ("port", port)):
    t0 = time.time()
    for s in seqs:
        lib.extract_kmers(s, 10, True)
    t1 = time.time()
    acc = 0
    for i, j in pairs:
        acc += lib.pair_score(seqs[i], seqs[j], 10, 0)[0]
    t2 = time.time()
    out[name] = {"extract_kmers_us_per_read": (t1 - t0) / n_reads * 1e6, "pair_score_us_per_pair": (t2 - t1) / len(pairs) * 1e6, "checksum_bases": int(acc)}
assert out["reference"]["checksum_bases"] == out["port"]["checksum_bases"]
out["port_over_reference"] = {k: out["port"][k] / out["reference"][k] for k in ("extract_kmers_us_per_read", "pair_score_us_per_pair")}
out["note"] = ("pair_score includes the k-mer extraction of both reads (the shim's call shape, ctypes overhead ~10 us included on both sides); "
               "a ratio <= 1 means the restatement bench.py times as cpu_baseline is not slower than the reference's own code")
print(json.dumps(out))
