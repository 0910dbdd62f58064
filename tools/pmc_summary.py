"""Sum rocprofv3 --pmc counter values by kernel and turn the kernel C numbers into per-cell constants.

usage: pmc_summary.py OUT.json BENCH.json DIR [DIR ...]
  DIR       = output directories of `rocprofv3 --kernel-trace --pmc <counters> -d DIR --output-format csv -- python bench.py ...`
              (one pass per counter group; every pass runs the same deterministic workload)
  BENCH.json = the bench line of one of those runs (for the exact DP cell count of the workload)
Writes the per-kernel sums and, for the poa kernels together: VALU wave-instructions per DP cell, HBM bytes per cell
(FETCH_SIZE doubled as the gfx950 guide prescribes for wide reads, WRITE_SIZE as reported; both counters are in KB)."""
import csv
import glob
import json
import os
import sys

out_path, bench_path, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
sums = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            c = row["Counter_Name"]
            e = sums.setdefault(c, {}).setdefault(k, {"dispatches": 0, "sum": 0.0})
            e["dispatches"] += 1
            e["sum"] += float(row["Counter_Value"])
bench = json.loads([l for l in open(bench_path) if l.startswith("{")][-1])
# per COMPUTED cell (round 6: POA #2 / #3 run inside an exact band; bench.py prices its roofline with the computed cells as well)
cells = bench["config"].get("poa_dp_cells_computed", bench["config"]["poa_dp_cells_per_step"]) * (bench["steps"] + bench["warmup"])
poa = {c: sum(v["sum"] for k, v in ks.items() if "poa_kernel" in k) for c, ks in sums.items()}
res = {"workload": bench["config"]["workload"], "dp_cells_profiled": cells, "dp_cells_reference": bench["config"]["poa_dp_cells_per_step"] * (bench["steps"] + bench["warmup"]), "counters_by_kernel": sums, "poa_kernels_total": poa}
if "SQ_INSTS_VALU" in poa:
    res["valu_wave_instr_per_cell"] = poa["SQ_INSTS_VALU"] / cells
    res["salu_wave_instr_per_cell"] = poa.get("SQ_INSTS_SALU", 0) / cells
    res["lds_wave_instr_per_cell"] = poa.get("SQ_INSTS_LDS", 0) / cells
if "SQ_WAVE_CYCLES" in poa and "SQ_ACTIVE_INST_VALU" in poa:
    res["valu_active_fraction_of_wave_cycles"] = poa["SQ_ACTIVE_INST_VALU"] / poa["SQ_WAVE_CYCLES"]
if "SQ_WAVE_CYCLES" in poa and "SQ_WAIT_ANY" in poa:
    res["wait_any_fraction_of_wave_cycles"] = poa["SQ_WAIT_ANY"] / poa["SQ_WAVE_CYCLES"]
# what the waves wait for: issue stalls on the LDS queue, the scalar unit's share, scalar memory instructions
for name, num in (("lds_issue_stall_fraction_of_wave_cycles", "SQ_WAIT_INST_LDS"), ("scalar_active_fraction_of_wave_cycles", "SQ_ACTIVE_INST_SCA"),
                  ("lds_active_fraction_of_wave_cycles", "SQ_ACTIVE_INST_LDS"), ("issue_stall_fraction_of_wave_cycles", "SQ_WAIT_INST_ANY")):
    if num in poa and poa.get("SQ_WAVE_CYCLES"):
        res[name] = poa[num] / poa["SQ_WAVE_CYCLES"]
if "SQ_INSTS_SMEM" in poa:
    res["smem_wave_instr_per_cell"] = poa["SQ_INSTS_SMEM"] / cells
if "FETCH_SIZE" in poa and "WRITE_SIZE" in poa:
    res["hbm_fetch_bytes_per_cell_x2"] = 2 * 1024 * poa["FETCH_SIZE"] / cells
    res["hbm_write_bytes_per_cell"] = 1024 * poa["WRITE_SIZE"] / cells
    res["hbm_bytes_per_cell"] = res["hbm_fetch_bytes_per_cell_x2"] + res["hbm_write_bytes_per_cell"]
    res["note"] = "FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B); WRITE_SIZE as reported"
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in ("rattle_amd/csrc/poa.hip", "rattle_amd/csrc/common.h"):
    h.update(open(os.path.join(root, f), "rb").read())
res["kernel_sources_sha256"] = h.hexdigest()          # bench.py compares it with the tree it runs in (pmc_stale)
res["kernel_sources"] = ["rattle_amd/csrc/poa.hip", "rattle_amd/csrc/common.h"]
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ("counters_by_kernel",)}, indent=1))
