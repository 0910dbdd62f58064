"""The multi-GPU curve of the sharded job, MEASURED on one GPU (round 5).

A sharded job (rank r of R scores every R-th candidate of a seed batch, runs the packs LPT gave it, ...) exchanges small payloads
at fixed points: accepted hits per greedy round of `cluster`, pack / cluster consensi after the stages of `correct`, and at the end
the corrected reads travel to the root.  What the ranks TOGETHER put into an exchange is exactly what the single-rank job holds at
that point, so:
  1. `record`: the single-rank job runs with RATTLE_XCHG_RECORD=<file> and appends its whole payload at every exchange point
     (this run is also the 1-GPU time of the curve);
  2. `replay r`: a context configured as rank r of R (rattle_hip_set_exchange with no callback, RATTLE_XCHG_REPLAY=<file>) runs
     ITS share alone on the device and, at every exchange, takes its peers' part from the record; its pieces for the final gather
     go to files, the root runs last and merges them.
Per rank: wall time of `cluster` and `correct` (+ the root's gather); the R-GPU step without the links = max over the ranks.
What is NOT in it: the xGMI transfers themselves (KBs per greedy round, a few MB per stage, 1 / R of 2 GB to the root).

usage: rank_replay.py --reads N --worlds 2,4,8 [--out "profiles/round5_rank_replay_{}.json"]      (runs the record + every rank of every world as subprocesses)
"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(a):
    import ctypes as C
    import numpy as np
    import bench
    from rattle_amd.api import Context, K_POA
    genes = max(1, a.reads // 200)
    cache = f"/tmp/rattle_replay_workload_{a.reads}.npz"       # every rank's process reads the same reads: generate them once
    if os.path.exists(cache):
        z = np.load(cache)
        cat, qcat, off = z["cat"], z["qcat"], z["off"]
    else:
        cat, qcat, off, tid, _ = bench.make_workload(a.reads, genes, seed=20260929)[:5]
        np.savez(cache + ".tmp.npz", cat=cat, qcat=qcat, off=off)
        os.replace(cache + ".tmp.npz", cache)
    ctx = Context(0)
    if a.rank >= 0:
        from rattle_amd.api import check
        from rattle_amd import _lib
        check(ctx.lib.rattle_hip_set_exchange(ctx.h, a.rank, a.world, C.cast(None, _lib.ALLGATHERV_FN), None))
    ctx.stage_reads(cat, qcat, off)
    out = []
    for step in range(2):                      # the first step pays the arena allocation: the second is the measurement
        ctx.reset_stats()
        t0 = time.time()
        cl = ctx.cluster_unsorted_packed(cat, off, k=10)
        t1 = time.time()
        h = ctx.correct_packed(cat, qcat, off, cl, gather_root=None, keep=True)
        t2 = time.time()
        g = 0.0
        counts = None
        if a.rank >= 0:
            import ctypes
            from rattle_amd.api import Correction, CorrectionHandle, check
            merged = ctypes.POINTER(Correction)()
            tg = time.time()
            check(ctx.lib.rattle_hip_correction_gather(ctx.h, h.ptr, 0, ctypes.byref(merged)))
            g = time.time() - tg
            if a.rank == 0:
                hm = CorrectionHandle(ctx.lib, merged)
                counts = hm.counts()[:3]
                digest = hm.digest()
                hm.free()
        else:
            counts = h.counts()[:3]
            digest = h.digest()
        ms, launches, _ = ctx.kernel_stats(K_POA)
        h.free()
        out = {"rank": a.rank, "world": a.world, "cluster_s": round(t1 - t0, 4), "correct_s": round(t2 - t1, 4), "gather_s": round(g, 4),
               "kernel_c_ms": round(ms, 1), "clusters": int(len(cl.main_id))}
        if counts is not None:
            out["counts"] = [int(x) for x in counts]
            out["digest"] = int(digest)
    print("RANK_REPLAY " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1000000)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--worlds", default=None, help="several worlds from ONE record, e.g. 2,4,8 (--out then takes a {} for the world)")
    ap.add_argument("--rank", type=int, default=None, help="(internal) -1: the recording single-rank run; r: replay rank r")
    ap.add_argument("--record", default="/tmp/rattle_xchg_record.bin")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.rank is not None:
        return one(a)
    rows = []

    def run(rank, env_extra):
        env = dict(os.environ, RATTLE_TIMING="1", **env_extra)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--reads", str(a.reads), "--world", str(a.world), "--rank", str(rank), "--record", a.record],
                           capture_output=True, text=True, env=env, timeout=3000)
        with open(f"{a.record}.w{a.world}.rank{rank}.err", "w") as f:            # the library's stage timings of this rank, for the curious
            f.write(r.stderr)
        line = [l for l in r.stdout.splitlines() if l.startswith("RANK_REPLAY ")]
        if r.returncode != 0 or not line:
            raise SystemExit(f"rank {rank} failed: {r.stdout[-1500:]} {r.stderr[-3000:]}")
        row = json.loads(line[-1][len("RANK_REPLAY "):])
        # the gather, split (the library's own timers, last step): building this rank's piece, moving it (files here: NOT what links cost),
        # merging on the root
        import re
        for key, pat in (("gather_serialise_s", r"gather: serialise this rank's share\s+([0-9.]+) ms"), ("gather_transport_files_s", r"gather: transport\s+([0-9.]+) ms"),
                         ("gather_merge_s", r"gather: merge on the root\s+([0-9.]+) ms")):
            m = re.findall(pat, r.stderr)
            row[key] = round(float(m[-1]) / 1e3, 4) if m else 0.0
        stages = {}
        for name in ("stage 1", "stage 2a", "stage 2b+3a", "stage 3b"):
            m = re.findall(r"correct: " + re.escape(name) + r"\s+([0-9.]+) ms", r.stderr)
            if m:
                stages[name] = round(float(m[-1]) / 1e3, 4)
        row["correct_stages_s"] = stages
        return row

    single = run(-1, {"RATTLE_XCHG_RECORD": a.record})
    print("single rank:", single, flush=True)
    bad = False
    for world in ([int(x) for x in a.worlds.split(",")] if a.worlds else [a.world]):
        a.world = world
        rows = []
        for rank in list(range(1, world)) + [0]:                   # the root last: it picks up the others' gather pieces
            rows.append(run(rank, {"RATTLE_XCHG_REPLAY": a.record}))
            print(rows[-1], flush=True)
        root = rows[-1]
        ok = root.get("counts") == single.get("counts") and root.get("digest") == single.get("digest")
        # the step without the links: slowest rank of each phase + the root's share of the gather (its own piece + the merge)
        step = max(r["cluster_s"] for r in rows) + max(r["correct_s"] for r in rows) + root["gather_serialise_s"] + root["gather_merge_s"]
        res = {"what": "sharded job replayed rank by rank on ONE MI355X (tools/rank_replay.py): per-rank wall times with the peers' exchange payloads from a record; no link time",
               "reads": a.reads, "world": world, "single_rank": single, "ranks": sorted(rows, key=lambda r: r["rank"]),
               "root_result_equals_single_rank": ok,
               "step_s_without_links": round(step, 4), "single_rank_step_s": round(single["cluster_s"] + single["correct_s"], 4),
               "speedup": round((single["cluster_s"] + single["correct_s"]) / step, 3),
               "cluster_max_s": max(r["cluster_s"] for r in rows), "correct_max_s": max(r["correct_s"] for r in rows),
               "root_gather_s": round(root["gather_serialise_s"] + root["gather_merge_s"], 4), "root_gather_with_file_transport_s": root["gather_s"]}
        print(json.dumps(res), flush=True)
        if a.out:
            with open(a.out.format(world), "w") as f:
                json.dump(res, f, indent=1)
        bad |= not ok
    if bad:
        raise SystemExit("the root's merged result differs from the single-rank result")


if __name__ == "__main__":
    main()
