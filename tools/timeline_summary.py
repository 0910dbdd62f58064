"""Summarise RATTLE_POA_TIMELINE dumps: per pass, how many packs' workgroups are active over time.
usage: timeline_summary.py FILE [bins]"""
import sys
import numpy as np
passes, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur: passes.append(np.array(cur, np.float64))
        cur = []
        continue
    cur.append([float(x) for x in line.split()])
if cur: passes.append(np.array(cur, np.float64))
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for i, P in enumerate(passes):
    P = P[P[:, 5] > 0]
    if not len(P): continue
    t0, t1 = P[:, 4].min(), P[:, 5].max()
    dur = (t1 - t0) * 1e-8
    busy = ((P[:, 5] - P[:, 4]) * 1e-8).sum()
    edges = np.linspace(t0, t1, bins + 1)
    act = [float(np.clip(np.minimum(P[:, 5], edges[b + 1]) - np.maximum(P[:, 4], edges[b]), 0, None).sum() / (edges[b + 1] - edges[b])) for b in range(bins)]
    big = P[P[:, 2] >= 150]
    print(f"pass {i}: {len(P)} packs, wall {dur:.3f} s, block-seconds {busy:.1f}, mean active blocks {busy / dur:.0f}, peak {max(act):.0f}; "
          f"packs of >=150 reads: {len(big)}, their mean duration {((big[:,5]-big[:,4])*1e-8).mean() if len(big) else 0:.3f} s, max {((big[:,5]-big[:,4])*1e-8).max() if len(big) else 0:.3f} s")
    for c in sorted(set(P[:, 1].astype(int))):
        Q = P[P[:, 1] == c]
        print(f"   class {c}: {len(Q)} packs, first start {(Q[:,4].min()-t0)*1e-8:.3f} s, last end {(Q[:,5].max()-t0)*1e-8:.3f} s, block-seconds {((Q[:,5]-Q[:,4])*1e-8).sum():.1f}")
    print("   active blocks per time bin:", " ".join(f"{a:.0f}" for a in act))
