#!/bin/bash
# SQ counters of kernel C's 32-bit classes (dp_rows_wide: 4096 / 6144 / 8192 columns; dp_rows_long: segments beyond 8192), one pass per class on
# the POA microbench (tools/bench_poa_class.py LEN PACKS DEPTH): instructions per DP cell, what the waves wait for.  usage: tools/pmc_wide.sh TAG
TAG=${1:-pmc_wide}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for spec in "1000 512 100" "3500 96 100" "5500 64 100" "7500 48 100" "12000 24 60"; do
  set -- $spec; n=L$1
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_$n -- python $GRAFT_REPO_ROOT/tools/bench_poa_class.py $1 $2 $3 0.10 1 > $O/pmc_$n.log 2>&1 )
  python - $O/pmc_$n "$spec" <<'PY'
import csv, glob, os, sys, re
sums, names = {}, set()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "poa_kernel" not in row["Kernel_Name"]: continue
        names.add(row["Kernel_Name"].split("(")[0].replace("void rattle::", ""))
        sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
log = open(sys.argv[1] + ".log").read()
m = re.search(r"kernel (\d+) ms\s+([\d.]+) GCUPS", log)
cells = float(m.group(1)) * 1e-3 * float(m.group(2)) * 1e9 if m else 0
wc = max(sums.get("SQ_WAVE_CYCLES", 1), 1)
print("LEN PACKS DEPTH =", sys.argv[2], sorted(names), "GCUPS (under the profiler)", m.group(2) if m else None,
      "| per DP cell:", {k.replace("SQ_INSTS_", "").lower(): round(v / cells, 4) for k, v in sums.items() if k.startswith("SQ_INSTS")} if cells else sums,
      "| of wave cycles: valu active", round(sums.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3), "wait_any", round(sums.get("SQ_WAIT_ANY", 0) / wc, 3), "wait_inst_any", round(sums.get("SQ_WAIT_INST_ANY", 0) / wc, 3))
PY
done | tee $O/summary.txt
