#!/bin/bash
# SQ counters of ONE kernel (name substring) over tools/cluster_only.py: usage tools/pmc_kernel.sh TAG KERNEL_SUBSTRING
TAG=$1; K=$2
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc -- python $GRAFT_REPO_ROOT/tools/cluster_only.py > $O/run.log 2>&1 )
python - $O/pmc "$K" <<'PY'
import csv, glob, os, sys
sums = {}; n = 0
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] not in row["Kernel_Name"]: continue
        sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
wc = max(sums.get("SQ_WAVE_CYCLES", 1), 1)
print(sys.argv[2], {k: f"{v:.4g}" for k, v in sums.items()})
print("valu_active/wave_cycles", round(sums.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3), "wait_any", round(sums.get("SQ_WAIT_ANY", 0) / wc, 3), "wait_inst", round(sums.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
      "valu per busy cycle", round(sums.get("SQ_INSTS_VALU", 0) / max(sums.get("SQ_BUSY_CYCLES", 1), 1), 3))
PY
