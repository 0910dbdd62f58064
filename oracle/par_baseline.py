"""ORACLE-side helper of bench.py's cpu_baseline_all_cores leg (test / measurement infrastructure, not shipped):
the oracle (cluster + correct, POA rows in AVX2 int16 where available) over a sample of whole transcripts, one task per transcript, on a
process pool.  usage: par_baseline.py SAMPLE.npz WORKERS  ->  one JSON line."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

_G = {}


def _run(task):
    import oracle as orc_mod
    from rattle_amd import hps
    orc = _G.get("orc")
    if orc is None:
        orc = _G["orc"] = orc_mod.Oracle()
        _G["simd"] = orc.set_poa_simd(True)              # AVX2 int16 POA rows where the CPU has them (what spoa's SIMD engine runs)
    s, q = task
    order = sorted(range(len(s)), key=lambda i: -len(s[i]))
    cl, _ = orc.cluster_reads([s[i] for i in order], k=10)
    clusters = [((order[m[0]], m[1], -1), [(order[x[0]], x[1], -1) for x in mem]) for m, mem in cl]
    orc.correct([b"@r%d" % i for i in range(len(s))], s, q, hps.encode(clusters))
    return len(s)


def main():
    z = np.load(sys.argv[1])
    workers = int(sys.argv[2])
    cat, qcat, off, grp = z["cat"], z["qcat"], z["off"], z["grp"]
    tasks = []
    for g in np.unique(grp):
        idx = np.nonzero(grp == g)[0]
        tasks.append(([cat[int(off[i]):int(off[i + 1])].tobytes() for i in idx], [qcat[int(off[i]):int(off[i + 1])].tobytes() for i in idx]))
    tasks.sort(key=lambda t: -len(t[0]))                 # longest first
    t0 = time.time()
    with mp.get_context("fork").Pool(workers) as pool:
        done = pool.map(_run, tasks, chunksize=1)
    dt = time.time() - t0
    import oracle as orc_mod
    print(json.dumps({"reads": int(sum(done)), "seconds": dt, "workers": workers, "tasks": len(tasks), "avx2": bool(orc_mod.Oracle().set_poa_simd(True))}))


if __name__ == "__main__":
    main()
