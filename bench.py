#!/usr/bin/env python3
"""Headline benchmark: reads/sec for `cluster` + `correct` on 1e6 synthetic ONT cDNA reads, 1 -> 8 GPUs.

One step = one full pass of the hot path over the SAME 1e6-read batch:
  sort + k-mer index (kernel K) -> greedy clustering (kernels A + B) -> correct (kernel C x3, kernel D).
N GPUs = N ranks (launched by torch.distributed.run) working on ONE job (strong scaling): every rank holds
the reads in HBM, the candidates of a seed batch / the `correct` packs are sharded over the ranks inside
librattle_hip.so, the exchange is RCCL all-gather(v) of small byte strings (hit lists, pack consensi), and
the corrected reads are gathered on rank 0, whose result must carry the same digest as the 1-GPU run.
`--weak` keeps the reads per GPU fixed instead (each rank its own data set, no exchange).
`--iso` benchmarks config 3 instead: the two-level `cluster --iso` flow (k=10 then k=11).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` for the dominant kernel
(poa_align) and `cpu_baseline` (the oracle timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from rattle_amd import synth  # noqa: E402
from rattle_amd.api import K_FILTER, K_KMER, K_POA, K_POST, K_SCORE, Context  # noqa: E402

METRIC = "reads/sec for `cluster`+`correct` on 1e6×1kb synthetic ONT reads, 1→8 GPU"
VALU_PEAK_TINSTR = 256 * 4 * 2.4e9 / 2 / 1e12      # wave64 VALU instructions per second: 256 CUs x 4 SIMDs, 2 cycles each at 2.4 GHz
VALU_PRACTICAL_TINSTR = 256 * 4 * 2.4e9 / 4.3 / 1e12      # ... at the issue rate MEASURED for this kernel's mix (v_pk_*_i16, v_max_i32, DPP, v_perm: 4.3 cycles, tools/ubench_valu.hip)
SALU_PEAK_TINSTR = 256 * 2.4e9 / 1e12               # one scalar unit per CU, one instruction per cycle


def device_state(light=False):
    """What the box says about the device's clocks and power right now (sysfs; None where it says nothing): boxes and consecutive runs
    differ by +-8 % for one binary, so the line carries the state it was measured in, before and after the timed region.
    light = True: the clock and one power reading only (two files instead of six) -- what the sampler reads WHILE the timed region runs:
    every one of these files is a query to the device's power-management firmware."""
    import glob
    st = {}
    try:
        for d in sorted(glob.glob("/sys/class/drm/card*/device")):
            if not os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                continue
            for key, f in ((("sclk", "pp_dpm_sclk"),) if light else (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"))):
                try:
                    cur = [l.split(":")[1].strip().rstrip("*").strip() for l in open(os.path.join(d, f)) if "*" in l]
                    st[key] = cur[0] if cur else None
                except Exception:
                    st[key] = None
            if not light:
                try:
                    st["busy_percent"] = int(open(os.path.join(d, "gpu_busy_percent")).read())
                except Exception:
                    pass
            for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
                for key, f, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6), ("temp_c", "temp1_input", 1e-3)):
                    if key in st or (light and key != "power_w"):
                        continue                                       # (one power file is enough: the second is the fallback)
                    try:
                        st[key] = round(int(open(os.path.join(hw, f)).read()) * scale, 1)
                    except Exception:
                        pass
            break
    except Exception:
        pass
    return st or None


class DeviceSampler:
    """Samples device_state() every 100 ms on a thread: the state before and after a run is an idle device's (sclk ~100 MHz) and says nothing
    about the clocks the kernels ran at.  summary(): mean / min / max of sclk, power, temperature.
    It runs during the WARM-UP steps only (same reads, same kernels as the timed ones): every one of these files is a query to the device's
    power-management firmware, and the timed region is better left without them.  (Round 6 suspected this polling -- six files ten times a
    second -- of the intermittent slow stage 1 of single steps, 4.5-4.9 s instead of 3.8 s at the same clock and lower power: 6 of 29 timed
    steps with it, none in 26 steps of the same loop in processes without it, none in 16 steps with it throttled.  Then a step WITHOUT any
    polling was slow on yet another box.  The slow steps follow the box, not the sampler; DESIGN.md section 5.  Note also that the first card
    in sysfs is not always the device the process runs on.)"""

    def __init__(self, period=0.1, light=False):
        import threading
        self.period, self.light, self.samples, self._stop = period, light, [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            st = device_state(light=self.light)
            if st:
                st["t"] = time.time()
                self.samples.append(st)
            self._stop.wait(self.period)

    def start(self):
        self._t.start()
        return self

    def stop(self):
        self._stop.set()
        self._t.join(timeout=2)

    def summary(self, t0=None, t1=None, keys=("sclk", "mclk", "power_w", "temp_c", "busy_percent")):
        """mean / min / max of what sysfs said while the timed region ran, or (t0, t1 given) while one step of it ran"""
        if t0 is None:
            self.stop()
        sel = [st for st in self.samples if t0 is None or t0 <= st.get("t", 0) <= t1]
        out = {"samples": len(sel)}
        for key in keys:
            vals = []
            for st in sel:
                v = st.get(key)
                if isinstance(v, str):
                    v = "".join(ch for ch in v if ch.isdigit() or ch == ".")
                    v = float(v) if v else None
                if v is not None:
                    vals.append(float(v))
            if vals:
                out[key] = {"mean": round(sum(vals) / len(vals), 1), "min": min(vals), "max": max(vals)}
        return out


def make_workload(n_reads, genes, seed, isoforms=1):
    # mean ~1 kb transcripts (8 exons of U[50,210]), 10 % error, both strands (cDNA); packed arrays
    return synth.reads_packed(n_reads, genes, isoforms, True, seed=seed, exon=(50, 210))


PHASES = {"cluster": 0.0, "correct": 0.0}      # host wall time of the two calls, summed over the timed steps


def run_step(ctx, cat, qcat, off, root, k=10):
    """`rattle cluster` (sort + index + gene-level cluster_reads + id translation, main.cpp:254-277)
    then `rattle correct` (correct_reads, main.cpp:405) on the same reads; several ranks: the sharded result
    is reassembled on `root`."""
    t0 = time.time()
    cl = ctx.cluster_unsorted_packed(cat, off, k=k)
    t1 = time.time()
    res = ctx.correct_packed(cat, qcat, off, cl, gather_root=root, keep=True)
    t2 = time.time()
    PHASES["cluster"] += t1 - t0
    PHASES["correct"] += t2 - t1
    return cl, res


def expected_consensi(cl, split=200, min_reads=5):
    """clusters owning at least one pack of more than min_reads reads (correct.cpp:331-360)"""
    sizes = np.diff(cl.offsets.astype(np.int64))
    nf = (sizes - 1) // split + 1
    return int(((sizes - 1) // np.maximum(nf, 1) + 1 > min_reads).sum())


def cluster_digest(cl):
    import zlib
    crc = 0
    for a in (cl.main_id, cl.main_rev, cl.offsets, cl.member_id, cl.member_rev):
        crc = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc)
    return crc


def cpu_baseline(cat, qcat, off, tid, correct_reads=1800, cluster_reads=5000):
    """Oracle (CPU restatement, 1 thread), timed per phase on bounded samples of the same workload: `cluster` on
    every read of randomly chosen transcripts up to ~cluster_reads, `correct` on a ~correct_reads subset of whole
    transcripts (so per-cluster depth matches the full workload).  The POA matrices are filled by the oracle's AVX2 int16
    row kernel (16 columns per instruction, what spoa's own SIMD engine does; tests/test_oracle_correct.py checks it against
    the scalar loops); the scalar rate is measured beside it on a third of the sample.  The combined rate is the harmonic sum."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc_mod
    from rattle_amd import hps
    orc = orc_mod.Oracle()
    rng = np.random.default_rng(1)
    counts = np.bincount(tid)
    perm = rng.permutation(int(tid.max()) + 1)

    def sample(target, lo, hi):
        ids = []
        for g in perm:
            if counts[g] > hi or counts[g] < lo:
                continue
            ids += [int(i) for i in np.nonzero(tid == g)[0]]
            if len(ids) >= target:
                break
        return ids

    ids_c = sample(cluster_reads, 6, cluster_reads // 4)
    s = [cat[int(off[i]):int(off[i + 1])].tobytes() for i in ids_c]
    t0 = time.time()
    order = sorted(range(len(s)), key=lambda i: -len(s[i]))
    cl, cnt = orc.cluster_reads([s[i] for i in order], k=10)
    dt_cluster = time.time() - t0

    def time_correct(n_target, simd):
        ids_k = sample(n_target, 6, max(12, n_target // 2))
        s2 = [cat[int(off[i]):int(off[i + 1])].tobytes() for i in ids_k]
        q2 = [qcat[int(off[i]):int(off[i + 1])].tobytes() for i in ids_k]
        order2 = sorted(range(len(s2)), key=lambda i: -len(s2[i]))
        cl2, _ = orc.cluster_reads([s2[i] for i in order2], k=10)
        clusters = [((order2[m[0]], m[1], -1), [(order2[x[0]], x[1], -1) for x in mem]) for m, mem in cl2]
        have = orc.set_poa_simd(simd)
        t1 = time.time()
        try:
            orc.correct([b"@r%d" % i for i in range(len(s2))], s2, q2, hps.encode(clusters))
        finally:
            orc.set_poa_simd(False)
        return len(s2), time.time() - t1, have

    n_v, dt_v, avx2 = time_correct(correct_reads, True)
    n_s, dt_s, _ = time_correct(correct_reads // 3, False)
    r_cluster, r_correct, r_scalar = len(s) / dt_cluster, n_v / dt_v, n_s / dt_s
    return {"value": 1.0 / (1.0 / r_cluster + 1.0 / r_correct), "unit": "reads/s", "cores": 1, "kind": "port",
            "poa_rows": "AVX2 int16" if avx2 else "scalar (no AVX2 on this CPU)",
            "cluster_reads_per_s": r_cluster, "correct_reads_per_s": r_correct, "correct_reads_per_s_scalar_rows": r_scalar,
            "value_scalar_rows": 1.0 / (1.0 / r_cluster + 1.0 / r_scalar),
            "sample": f"oracle, one thread, per phase on whole transcripts of the same workload: cluster {len(s)} reads in {dt_cluster:.1f} s "
                      f"({int(cnt[0])} pair tests, {int(cnt[1])} full comparisons), correct {n_v} reads in {dt_v:.1f} s with AVX2 int16 POA rows "
                      f"({n_s} reads in {dt_s:.1f} s with scalar rows); value = harmonic sum"}


def cpu_baseline_all_cores(cat, qcat, off, tid, whole=False, what="transcripts (6..100 reads each)"):
    """The same oracle on every host core: one task per whole transcript on a process pool
    (oracle/par_baseline.py, run as a separate process so the pool forks without a HIP runtime).
    Transcripts are clustered separately, which spares the CPU the cross-transcript filter tests.
    whole = True: every group of `tid` (the toyset: one task per cluster of the reference's own clusters.out)."""
    import subprocess
    import tempfile
    cores = os.cpu_count() or 1
    counts = np.bincount(tid)
    rng = np.random.default_rng(2)
    keep = [int(g) for g in rng.permutation(len(counts)) if 6 <= counts[g] <= 100][:max(8, cores)]      # bounded: ~20-30 s of wall time
    if whole:
        keep = list(range(len(counts)))
    sel = np.nonzero(np.isin(tid, keep))[0]
    lens = (off[sel + 1] - off[sel]).astype(np.int64)
    o2 = np.zeros(len(sel) + 1, np.uint64)
    o2[1:] = np.cumsum(lens)
    idx = np.concatenate([np.arange(int(off[i]), int(off[i + 1])) for i in sel]) if len(sel) else np.zeros(0, np.int64)
    model = ""
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sample.npz")
        np.savez(path, cat=cat[idx], qcat=qcat[idx], off=o2, grp=tid[sel])
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "par_baseline.py"), path, str(cores)], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": r.stderr[-300:]}
    j = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": j["reads"] / j["seconds"], "unit": "reads/s", "cores": int(min(cores, j["tasks"])), "nproc": cores, "cpu": model, "kind": "port", "poa_rows": "AVX2 int16" if j.get("avx2") else "scalar",
            "sample": f"{j['reads']} reads = every read of {j['tasks']} {what}, one oracle task per group, {j['seconds']:.1f} s"}


def _latest(*names):
    for n in names:
        if os.path.exists(os.path.join(ROOT, "profiles", n)):
            return os.path.join("profiles", n)
    return os.path.join("profiles", names[-1])


PMC_FILE = _latest("round6_pmc_poa.json", "round5_pmc_poa.json", "round4_pmc_poa.json", "round3_pmc_poa.json")      # kernel C (tools/gpu_pmc_only.sh); `pmc_stale` says whether it matches the tree
PMC_100K_FILE = _latest("round6_pmc_poa_100k.json", "round5_pmc_poa_100k.json", "round4_pmc_poa.json")      # ... at 1e5 reads: the under-filled device runs other forms of the row loop (teams of wavefronts)
PMC_ISO_FILE = _latest("round6_pmc_iso.json", "round5_pmc_iso.json", "round4_pmc_iso.json", "round3_pmc_iso.json")      # kernel B in the --iso flow (tools/gpu_pmc_iso.sh)


def toyset_line(ctx_cls, device, cpu_too=True):
    """The reference's own data set (toyset/rna, 8306 real ONT reads, recovered into tests/golden/) through the HIP path:
    `cluster --rna` + `correct`, timed beside the synthetic headline (the only published reference timings are for this set:
    README.md:400-403, 16 s / 76 s on 1 thread, 759 reads/s for `correct` on 24 threads).  The clusters are checked against
    the shipped clusters.out fixture."""
    import gzip
    from rattle_amd import hps
    from rattle_amd.api import pack_reads
    path = os.path.join(ROOT, "tests", "golden", "toyset_rna.fastq.gz")
    if not os.path.exists(path):
        return None
    lines = gzip.open(path, "rb").read().split(b"\n")
    seqs, quals = lines[1::4], lines[3::4]
    seqs = [s for s in seqs if s]; quals = quals[:len(seqs)]
    cat, off = pack_reads(seqs)
    qcat = np.frombuffer(b"".join(quals), np.uint8).copy()
    ctx = ctx_cls(device)
    best = None
    for _ in range(2):                                  # second pass: arena and kernels warm
        t0 = time.time()
        cl = ctx.cluster_unsorted_packed(cat, off, k=10, is_rna=True)
        t1 = time.time()
        res = ctx.correct_packed(cat, qcat, off, cl, vote_order=b"U-GTAC")
        t2 = time.time()
        best = (t1 - t0, t2 - t1, cl, res)
    want = hps.decode(open(os.path.join(ROOT, "tests", "golden", "toyset_rna.clusters.out"), "rb").read(), fields=2)
    same = [((m[0], m[1]), [(x[0], x[1]) for x in mem]) for m, mem in best[2].as_list()] == [((m[0], m[1]), [(x[0], x[1]) for x in mem]) for m, mem in want]
    ctx.close()
    n = len(seqs)
    # the same 8306 reads through the oracle on every host core of THIS box: one task per cluster of the reference's clusters.out
    # (cluster_reads + correct of its members; the cross-cluster filter tests are spared the CPU, so the ratio is conservative)
    cpu = None
    if cpu_too:
        try:
            grp = np.zeros(n, np.int64)
            for ci, (m, mem) in enumerate(want):
                for x in mem:
                    grp[x[0]] = ci
            cpu = cpu_baseline_all_cores(cat, qcat, off, grp, whole=True, what="clusters of the reference's clusters.out")
            if "value" in cpu:
                cpu["gpu_over_cpu"] = (n / (best[0] + best[1])) / cpu["value"]
        except Exception as e:
            cpu = {"error": str(e)[:200]}
    return {"reads": n, "cpu_all_cores": cpu, "cluster_s": best[0], "correct_s": best[1], "poa_dp_cells": int(best[3][3][0]), "cluster_reads_per_s": n / best[0], "correct_reads_per_s": n / best[1],
            "reads_per_s": n / (best[0] + best[1]), "clusters": int(len(best[2].main_id)), "clusters_equal_reference_fixture": bool(same),
            "consensi": int(best[3][2]), "reference_published": "README.md:400-403: cluster 16 s, correct 76 s on 1 thread (516 / 109 reads/s); correct 759 reads/s on 24 threads",
            "note": "small input: ~550 packs do not fill one MI355X (a pass lasts as long as its largest pack)"}


def pmc_reference(path=None):
    """Counter-derived constants of kernel C, measured in separate rocprofv3 --pmc passes (tools/gpu_pmc_only.sh) and
    committed under profiles/ (the counters cannot be read from inside the benchmark process).  The file records a hash of
    the kernel's sources; `stale` says whether the tree this benchmark runs in still has those sources."""
    import hashlib
    try:
        d = json.load(open(os.path.join(ROOT, path or PMC_FILE)))
    except Exception:
        return None
    h = hashlib.sha256()
    try:
        for f in d.get("kernel_sources", []):
            h.update(open(os.path.join(ROOT, f), "rb").read())
        d["stale"] = h.hexdigest() != d.get("kernel_sources_sha256")
    except Exception:
        d["stale"] = True
    return d


ALG_VALU_PER_CELL = (19 + 2 * 2.18) / 128 + 10 / (64 * 5)      # see poa_roofline
KNAMES = {K_KMER: "kmer_extract", K_FILTER: "bv_filter", K_SCORE: "pair_score", K_POA: "poa_align", K_POST: "post_msa"}


def iso_roofline(kst):
    """kernel B in the `--iso` flow: algorithmic bytes (8 B x (nK_i + nK_j) per comparison) over its HIP-event time"""
    ms, launches, alg = kst["pair_score"]
    ach = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    pmc_iso = pmc_reference(PMC_ISO_FILE)
    ratio = pmc_iso.get("hbm_bytes_per_algorithmic_byte") if pmc_iso else None
    return {"bound": "hbm", "kernel": "pair_score", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
            "traffic": ratio * alg / max(launches, 1) if ratio else None,
            "hbm_bytes_per_algorithmic_byte": ratio, "pmc_source": PMC_ISO_FILE if pmc_iso else None,
            "pmc_stale": pmc_iso["stale"] if pmc_iso else None,
            "alg_bytes_per_launch": alg / max(launches, 1), "avg_launch_ms": ms / max(launches, 1), "launches": launches,
            "note": "8 B x (nK_i + nK_j) per comparison (SURVEY 8d) / kernel time from HIP events; traffic = FETCH_SIZE(x2) + WRITE_SIZE of "
                    "kernel B from the committed PMC passes of the same flow (tools/gpu_pmc_iso.sh), per launch"}


def poa_roofline(kst, cells, steps, per_gpu=1, copy_gbs=None, pmc_file=None, cells_reference=None):
    """kernel C: exact DP cells over its HIP-event time, priced in wave64 VALU instructions per second (it is bound by VALU issue /
    per-row latency, not by HBM: SURVEY 8d, profiles/) against 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles; `cells` = the cells the device
    COMPUTED per step (the PMC constants are per computed cell too), `cells_reference` = rows x columns of every alignment"""
    ms, launches, alg = kst["poa_align"]
    gcups = cells / per_gpu / (ms / steps * 1e-3) / 1e9 if ms > 0 else 0.0
    hbm6 = 6.0 * gcups
    pmc = pmc_reference(pmc_file)
    ipc = pmc.get("valu_wave_instr_per_cell") if pmc else None
    spc = pmc.get("salu_wave_instr_per_cell") if pmc else None
    ach = gcups * 1e9 * ipc / 1e12 if ipc else None
    return {
        "bound": "valu_issue", "kernel": "poa_align", "achieved": ach, "peak": VALU_PEAK_TINSTR, "unit": "Tinstr/s",
        "frac": ach / VALU_PEAK_TINSTR if ach else None,
        # `frac` falls when instructions are cut at equal GCUPS; the two gauges that do not mislead: the fraction of the issue rate this
        # instruction mix can reach at all, and how busy the CU's one scalar unit is (round 4: it was the busier of the two)
        "frac_practical": ach / VALU_PRACTICAL_TINSTR if ach else None, "peak_practical": VALU_PRACTICAL_TINSTR,
        "salu_wave_instr_per_cell": spc, "salu_frac": gcups * 1e9 * spc / 1e12 / SALU_PEAK_TINSTR if spc else None,
        "valu_wave_instr_per_cell": ipc, "gcups": gcups, "avg_launch_ms": ms / max(launches, 1), "launches": launches,
        "gcups_reference_cells": (cells_reference / per_gpu / (ms / steps * 1e-3) / 1e9) if cells_reference and ms > 0 else None,
        # the recurrence's own instruction count: per column PAIR and row 19 packed operations + 2 per in-edge (ready-made
        # terms), i.e. (19 + 2 x 2.18) / 128 wave64 instructions per cell, + the row's scan and bookkeeping (~10 per wavefront and row over
        # 64 x 5 columns): what a kernel with NO overhead would issue.  frac_alg prices the achieved cell rate with THAT count, so it
        # falls when the kernel gets slower and does not rise when it wastes instructions
        "alg_valu_per_cell": ALG_VALU_PER_CELL, "frac_alg": gcups * 1e9 * ALG_VALU_PER_CELL / 1e12 / VALU_PEAK_TINSTR,
        "cells_per_launch": cells * steps / max(launches, 1),
        # HBM view: SURVEY 8(d)'s 6 B per DP cell (three int16 matrices) and what the kernel really stores
        "hbm": {"achieved_6B_per_cell_gbs": hbm6, "peak_gbs": 8000.0, "frac_6B_per_cell": hbm6 / 8000.0, "measured_copy_gbs": copy_gbs,
                "stored_bytes_per_cell": pmc.get("hbm_bytes_per_cell") if pmc else None,
                "achieved_stored_gbs": gcups * pmc["hbm_bytes_per_cell"] if pmc and pmc.get("hbm_bytes_per_cell") else None},
        "traffic": (pmc["hbm_bytes_per_cell"] * cells * steps / max(launches, 1)) if pmc and pmc.get("hbm_bytes_per_cell") else None,
        "pmc_source": (pmc_file or PMC_FILE) if pmc else None,
        # true: poa.hip / common.h changed after the counter passes were collected -- the per-cell constants (and `frac`,
        # `traffic`) then describe an older kernel; GCUPS and the timings are always live
        "pmc_stale": bool(pmc.get("stale")) if pmc else None,
        "note": "achieved = exact DP cells / kernel time (HIP events on the library's streams) x VALU wave-instructions per cell from the "
                "committed SQ_INSTS_VALU pass of this tree; traffic = FETCH_SIZE(x2) + WRITE_SIZE per cell from the committed PMC passes x cells per launch"}


def side_config(device, n_reads, iso, steps=2):
    """One of BASELINE's other single-GPU configs, timed beside the headline on the same box and in the same process so that the
    driver's one default command measures it too: configs[1] (1e5 cDNA reads, gene-level `cluster`; `correct` follows because the
    under-filled device is what that size probes) and configs[2] (1e6 reads, `cluster --iso`).  Inputs resident in HBM, one
    warm-up, `steps` timed passes between device synchronisations, digests equal across the passes."""
    import torch
    genes = max(5, n_reads // (600 if iso else 200))
    cat, qcat, off, tid, _ = make_workload(n_reads, genes, seed=20260929, isoforms=3 if iso else 1)
    n = len(off) - 1
    ctx = Context(device)
    ctx.stage_reads(cat, qcat, off)
    t_cluster = t_correct = 0.0

    def one():
        nonlocal t_cluster, t_correct
        t0 = time.time()
        if iso:
            cl, gid, ng = ctx.cluster_iso_unsorted_packed(cat, off)
            t_cluster += time.time() - t0
            return cl, int(ng), None
        cl = ctx.cluster_unsorted_packed(cat, off)
        t1 = time.time()
        res = ctx.correct_packed(cat, qcat, off, cl, keep=True)
        t_cluster += t1 - t0
        t_correct += time.time() - t1
        return cl, None, res

    w = one()
    dg_w = (cluster_digest(w[0]), w[2].digest() if w[2] is not None else None)
    if w[2] is not None:
        w[2].free()
    ctx.reset_stats()
    t_cluster = t_correct = 0.0
    torch.cuda.synchronize()
    t0 = time.time()
    outs = [one() for _ in range(steps)]
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    kst = {KNAMES[k]: ctx.kernel_stats(k) for k in KNAMES}
    cl, ng, res = outs[-1]
    rec = {"reads": n, "value": n / dt, "unit": "reads/s", "ms_per_step": dt * 1e3, "steps": steps,
           "kernels_ms_per_step": {k: v[0] / steps for k, v in kst.items()},
           "phases_ms_per_step": {"cluster": t_cluster / steps * 1e3, **({"correct": t_correct / steps * 1e3} if not iso else {})}}
    if iso:
        assert len(cl.member_id) == n and np.array_equal(np.sort(cl.member_id), np.arange(n)), "--iso: not a partition of the reads"
        rec["workload"] = f"{n} synthetic cDNA reads, {genes} genes x 3 isoforms, `rattle cluster --iso` k=10 / k=11 (BASELINE configs[2])"
        rec["gene_clusters"], rec["transcript_clusters"] = ng, int(len(cl.main_id))
        rec["roofline"] = iso_roofline(kst)
        assert cluster_digest(cl) == dg_w[0], "--iso: result differs between passes"
    else:
        n_cor, n_unc, n_cons, counters = res.counts()
        assert n_cor + n_unc == n and n_cons == expected_consensi(cl)
        assert (cluster_digest(cl), res.digest()) == dg_w, "result differs between passes"
        rec["workload"] = f"{n} synthetic cDNA reads, {genes} transcripts, `rattle cluster` k=10 gene level + `rattle correct` (BASELINE configs[1] size)"
        rec["clusters"], rec["packs"], rec["cluster_reads_per_s"] = int(len(cl.main_id)), int(counters[2]), n / (t_cluster / steps)
        rec["poa_dp_cells_reference"], rec["poa_dp_cells_computed"] = int(counters[0]), int(counters[5]) or int(counters[0])
        rec["roofline"] = poa_roofline(kst, int(counters[5]) or int(counters[0]), steps, pmc_file=PMC_100K_FILE, cells_reference=int(counters[0]))
    rec["digest_equal_across_steps"] = True
    for o in outs:
        if o[2] is not None:
            o[2].free()
    ctx.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("RATTLE_BENCH_READS", 1000000)), help="reads of the job (per GPU with --weak)")
    ap.add_argument("--genes", type=int, default=0, help="transcripts (default reads/200; --iso: genes = reads/600 x 3 isoforms)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: every rank its own data set of --reads reads, no exchange")
    ap.add_argument("--iso", action="store_true", help="config 3: two-level `cluster --iso` (k=10, then k=11 per gene cluster) instead of cluster+correct")
    ap.add_argument("--transport", choices=["rccl", "gloo"], default="rccl", help="exchange of the sharded job: RCCL on device buffers, or host buffers through gloo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the side measurements of BASELINE configs[1] and configs[2]")
    ap.add_argument("--no-reference-digest", action="store_true", help="several ranks: skip the unsharded reference run the result is checked against")
    ap.add_argument("--no-stage", action="store_true", help="hand the reads over as host buffers every step (PCIe-inclusive rate)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become N ranks (one per GPU) under torch.distributed.run.  The reference
        # parallelises the same axes with in-process threads and needs no launcher (cluster.cpp:138-158, correct.cpp:377-392).
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); nothing was timed")
    # ranks of one node share the host cores for pack planning / result assembly
    os.environ.setdefault("RATTLE_HOST_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    same_device = bool(os.environ.get("RATTLE_BENCH_ONE_DEVICE"))      # tests: several ranks on one GPU (gloo transport only)
    if not same_device and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP device(s) on this node")
    if same_device:
        local = 0
    torch.cuda.set_device(local)
    use_dist = world > 1 or bool(os.environ.get("RATTLE_BENCH_FORCE_DIST"))      # the flag exercises the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if a.transport == "gloo" or same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    sharded = use_dist and not a.weak
    if a.iso:
        genes = a.genes or max(5, a.reads // 600)
        cat, qcat, off, tid, _ = make_workload(a.reads, genes, seed=20260929 + (rank if a.weak else 0), isoforms=3)
    else:
        genes = a.genes or max(5, a.reads // 200)
        cat, qcat, off, tid, _ = make_workload(a.reads, genes, seed=20260929 + (rank if a.weak else 0))
    n_reads = len(off) - 1

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # several ranks on one job: the result must equal the unsharded one -> rank 0 runs it once, untimed, first
    ref_digest = None
    if sharded and world > 1 and not a.no_reference_digest and not a.iso:
        if rank == 0:
            ref = Context(local)
            cl0 = ref.cluster_unsorted_packed(cat, off)
            h0 = ref.correct_packed(cat, qcat, off, cl0, keep=True)
            ref_digest = (cluster_digest(cl0), h0.digest())
            h0.free()
            ref.close()
        barrier()

    ctx = Context(local)
    transport = None
    if sharded:
        if a.transport == "gloo" or same_device:
            ctx.set_exchange_gloo()
            ctx.comm_probe()
            transport = "gloo(host)"
        else:
            # RCCL on device buffers; if the communicator does not come up or its self-test fails on ANY rank, all
            # ranks fall back together to host buffers over a gloo group (and the JSON line says so)
            why = ""
            try:
                ctx.comm_init_rccl()
                ctx.comm_probe()
            except Exception as e:
                why = str(e)
            bad = torch.tensor([1 if why else 0], device="cuda")
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            transport = "rccl"
            if int(bad.item()):
                try:
                    ctx.comm_destroy()
                except Exception:
                    pass
                host_group = dist.new_group(backend="gloo")
                ctx.set_exchange_gloo(host_group)
                ctx.comm_probe()
                transport = "gloo(host) -- RCCL transport failed its self-test" + (f" on this rank: {why}" if why else " on another rank")
    root = 0 if sharded else None

    def step():
        if a.iso:
            t0 = time.time()
            cl, gid, ng = ctx.cluster_iso_unsorted_packed(cat, off)
            PHASES["cluster"] += time.time() - t0
            return cl, (gid, ng)
        return run_step(ctx, cat, qcat, off, root)

    if not a.no_stage:     # inputs resident in HBM before the timed region (BASELINE metric definition)
        ctx.stage_reads(cat, qcat, off)
    warm = None
    state0 = device_state() if rank == 0 else None
    sampler = DeviceSampler().start() if rank == 0 and a.warmup else None       # (the warm-up steps only: see DeviceSampler)
    for _ in range(a.warmup):
        if warm is not None and not a.iso:
            warm[1].free()
        warm = step()
    state_during = sampler.summary() if sampler is not None else None           # (stops the thread)
    ctx.reset_stats()
    PHASES["cluster"] = PHASES["correct"] = 0.0
    # A step returns the corrected reads as ~2 GB of host buffers the CALLER frees (the reference returns std::vectors).  Giving
    # them back to the OS costs ~0.18 s per step of pure harness time (round 3's verdict), so the results of the timed steps are
    # kept and freed after the timed region when the host has the memory for it (psutil), else freed between steps as before.
    held = []
    hold = False
    if not a.iso and warm is not None:
        try:
            import psutil
            per = warm[1].host_bytes()
            hold = psutil.virtual_memory().available > 3 * per * (a.steps + 2)
        except Exception:
            hold = False
    barrier()
    t0 = time.time()
    last = None
    step_ms = []
    step_detail = []
    if not a.iso:
        ctx.stage_ms()                          # (reset: the stage times below are the timed steps')
    for _ in range(a.steps):
        if last is not None and not a.iso:
            if hold:
                held.append(last[1])
            else:
                last[1].free()
        ts = time.time()
        c0 = PHASES["cluster"]
        last = step()
        te = time.time()
        step_ms.append((te - ts) * 1e3)
        # which stage a slow step was slow in (the library keeps its stages' host wall time), and what the device reported meanwhile
        step_detail.append({"ms": round((te - ts) * 1e3, 1), "cluster_ms": round((PHASES["cluster"] - c0) * 1e3, 1),
                            "correct_stage_ms": {} if a.iso else {k: round(v, 1) for k, v in ctx.stage_ms().items()}, "t": (ts, te)})
    barrier()
    dt = time.time() - t0
    for d in step_detail:
        d.pop("t")
    state1 = device_state() if rank == 0 else None
    tf = time.time()
    for h in held:
        h.free()
    free_ms = (time.time() - tf) * 1e3 / max(1, len(held)) if held else None
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    total_reads = n_reads * (world if a.weak else 1)
    value = total_reads / (dt / a.steps)
    cl, res = last

    # ---- correctness of what was timed (outside the timed region) ----
    checks = {}
    if a.iso:
        gid, ng = res
        assert len(cl.member_id) == n_reads and np.array_equal(np.sort(cl.member_id), np.arange(n_reads)), "--iso: not a partition of the reads"
        assert np.all(np.diff(gid) >= 0) and (len(gid) == 0 or gid[-1] == ng - 1), "--iso: gene ids not in gene order"
        checks = {"partition": True, "gene_clusters": int(ng), "transcript_clusters": int(len(cl.main_id)), "cluster_digest": cluster_digest(cl)}
        if warm is not None:
            assert cluster_digest(warm[0]) == checks["cluster_digest"], "--iso: result differs between steps"
            checks["digest_equal_across_steps"] = True
    else:
        if rank == 0 or not sharded:
            n_cor, n_unc, n_cons, counters = res.counts()
            assert n_cor + n_unc == n_reads, f"accounting: {n_cor} corrected + {n_unc} uncorrected != {n_reads} reads"
            assert n_cons == expected_consensi(cl), f"consensus count {n_cons} != {expected_consensi(cl)}"
            assert len(cl.member_id) == n_reads, "cluster: not a partition of the reads"
            dg = (cluster_digest(cl), res.digest())
            checks = {"n_corrected": int(n_cor), "n_uncorrected": int(n_unc), "n_consensi": int(n_cons), "packs_skipped": int(counters[3]),
                      "cluster_digest": dg[0], "correct_digest": dg[1]}
            if warm is not None:
                assert (cluster_digest(warm[0]), warm[1].digest()) == dg, "result differs between steps"
                checks["digest_equal_across_steps"] = True
            if ref_digest is not None:
                assert ref_digest == dg, f"sharded result {dg} != single-GPU result {ref_digest}"
                checks["digest_equal_to_single_gpu"] = True

    if rank == 0:
        kst = {KNAMES[k]: ctx.kernel_stats(k) for k in KNAMES}
        # what this GPU sustains on a plain device-to-device copy (read + write bytes), SURVEY 8(d)
        buf = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(buf)
        dst.copy_(buf); torch.cuda.synchronize()
        tc = time.time()
        for _ in range(10):
            dst.copy_(buf)
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * buf.numel() / (time.time() - tc) / 1e9
        del buf, dst
        par = ("replicas%d" % world) if a.weak else ("shard%d" % world)
        out = {
            "metric": METRIC if not a.iso else "reads/sec for `cluster --iso` (two-level, k=10 then k=11) on 1e6x1kb synthetic ONT cDNA reads",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if a.weak else "strong", "vs_baseline": None,
            "dtype": "int16", "data": "synthetic", "inputs": "host buffers per step (PCIe inclusive)" if a.no_stage else "resident in HBM (rattle_hip_stage_reads)",
            "checks": checks,
            # every timed step on its own (first-step vs steady state), the state of the device around the timed region, and what the
            # harness kept OUT of the timed region: a step's ~2 GB result is freed after the timer stops when the host has the memory
            # (`results_held`), which costs `result_free_ms` per step when it is done between steps instead
            "step_ms": [round(x, 1) for x in step_ms], "step_detail": step_detail, "device_state": {"before": state0, "during_warmup": state_during, "after": state1,
                             "note": "sysfs (first card that has a clock file: not necessarily this process's device) is sampled during the warm-up steps only; the timed region runs without firmware queries"},
            "results_held": bool(hold), "result_free_ms": free_ms,
            "kernels_ms_per_step": {k: v[0] / a.steps for k, v in kst.items()},
            "phases_ms_per_step": {k: v / a.steps * 1e3 for k, v in PHASES.items() if v > 0},
            "phase_reads_per_s": {k: n_reads / (v / a.steps) for k, v in PHASES.items() if v > 0},
            "cluster_counters": {"bv_pair_tests": int(cl.counters[0]), "full_comparisons": int(cl.counters[1]), "kmer_matches": int(cl.counters[2]), "pairs_past_the_count_bound": int(cl.counters[5]),
                                 "seed_rounds": int(cl.counters[3]), "kernel_launches": int(cl.counters[4])},
        }
        if sharded:
            calls, xbytes = ctx.comm_stats()
            out["exchange"] = {"transport": transport, "ranks": world, "collectives": calls, "bytes_received": xbytes}
        if a.iso:
            out["config"] = {"workload": f"{n_reads} synthetic cDNA reads (mean 1 kb, 10% err, both strands, {genes} genes x 3 isoforms, Zipf abundance), "
                                         "`rattle cluster --iso` k=10 / iso k=11 (BASELINE configs[2])",
                             "reads": n_reads, "gene_clusters": checks["gene_clusters"], "transcript_clusters": checks["transcript_clusters"], "parallelism": par}
            out["roofline"] = iso_roofline(kst)
        else:
            n_cor, n_unc, n_cons, counters = res.counts()
            cells = int(counters[0])
            cells_done = int(counters[5]) or cells   # what the device computed: fewer where the exact band of POA #2 / #3 was certified
            per_gpu = world if sharded else 1      # several ranks on one job: `cells` is the job's total, the kernel time this GPU's
            out["config"] = {"workload": f"{n_reads} synthetic cDNA reads (mean 1 kb, 10% err, both strands, {genes} transcripts, Zipf abundance), "
                                         "`rattle cluster` k=10 gene level + `rattle correct` (BASELINE metric size; configs[1]/[3] shape)"
                                         + (" per GPU" if a.weak else ", ONE job over all GPUs"),
                             "reads": n_reads, "clusters": int(len(cl.main_id)), "poa_dp_cells_per_step": cells, "poa_dp_cells_reference": cells,
                             "poa_dp_cells_computed": cells_done, "poa_alignments_per_step": int(counters[1]),
                             "poa_band": {"alignments_certified": int(counters[6]), "failed_certificates": int(counters[7]),
                                          "note": "POA #2 / #3 (near-identical sequences) run inside an exact band on one wavefront, certified per alignment; "
                                                  "`reference` = rows x columns of every alignment, what the reference's engine fills"},
                             "packs": int(counters[2]), "parallelism": par}
            out["roofline"] = poa_roofline(kst, cells_done, a.steps, per_gpu, copy_gbs, cells_reference=cells)
            if not a.no_cpu_baseline:
                try:
                    out["toyset"] = toyset_line(Context, local)
                except Exception as e:       # the headline must not depend on the side measurement
                    out["toyset"] = {"error": str(e)[:300]}
                if world == 1 and not a.no_configs:
                    # BASELINE's other single-GPU configs on the same box, same process (each with its own kernel's roofline)
                    out["configs"] = {}
                    for name, nr, iso in (("config2_100k_cluster_correct", 100000, False), ("config3_1M_cluster_iso", 1000000, True)):
                        try:
                            out["configs"][name] = side_config(local, nr, iso)
                        except Exception as e:
                            out["configs"][name] = {"error": str(e)[:300]}
                if world == 1 and not a.no_configs and not a.no_stage:
                    # the PCIe-inclusive rate (never `value`): the same step with the reads handed over as host buffers in every call
                    try:
                        c2 = Context(local)
                        w2 = run_step(c2, cat, qcat, off, None); w2[1].free()
                        torch.cuda.synchronize(); tn = time.time()
                        w2 = run_step(c2, cat, qcat, off, None)
                        torch.cuda.synchronize(); dtn = time.time() - tn
                        ok = (cluster_digest(w2[0]), w2[1].digest()) == (checks["cluster_digest"], checks["correct_digest"])
                        w2[1].free(); c2.close()
                        out["no_stage"] = {"value": n_reads / dtn, "unit": "reads/s", "ms_per_step": dtn * 1e3, "steps": 1, "digest_equal_to_staged_run": bool(ok),
                                           "note": "reads and qualities (2 x ~1 GB) uploaded inside the timed step; the headline has them resident in HBM"}
                    except Exception as e:
                        out["no_stage"] = {"error": str(e)[:300]}
                out["cpu_baseline"] = cpu_baseline(cat, qcat, off, tid)
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(cat, qcat, off, tid)
        print(json.dumps(out))
    if not a.iso:
        res.free()
        if warm is not None:
            warm[1].free()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
