#!/bin/bash
# SQ counters of the POA microbench (tools/bench_poa_class.py LEN PACKS) for a list of library variants: usage tools/pmc_micro.sh TAG LEN name1 name2 ...
TAG=$1; LEN=$2; shift 2
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/rattle_amd/csrc/variants
for n in "$@"; do
  L=$V/librattle_hip_$n.so; [ $n = base ] && L=$GRAFT_REPO_ROOT/rattle_amd/csrc/librattle_hip.so
  ( cd /tmp && RATTLE_HIP_LIB=$L RATTLE_POA_EXP=$EXP timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_$n -- python $GRAFT_REPO_ROOT/tools/bench_poa_class.py $LEN 2560 200 0.10 1 > $O/pmc_$n.log 2>&1 )
  python - $O/pmc_$n $n <<'PY'
import csv, glob, os, sys
sums = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "poa_kernel" not in row["Kernel_Name"]: continue
        sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
log = open(sys.argv[1] + ".log").read()
import re
m = re.search(r"kernel (\d+) ms\s+([\d.]+) GCUPS", log)
cells = float(m.group(1)) * 1e-3 * float(m.group(2)) * 1e9 if m else 0
print(sys.argv[2], "GCUPS(under profiler)", m.group(2) if m else None, {k: round(v / cells, 4) for k, v in sums.items()} if cells else sums,
      "valu_active/wave_cycles", round(sums.get("SQ_ACTIVE_INST_VALU", 0) / max(sums.get("SQ_WAVE_CYCLES", 1), 1), 3),
      "wait_any", round(sums.get("SQ_WAIT_ANY", 0) / max(sums.get("SQ_WAVE_CYCLES", 1), 1), 3),
      "wait_inst", round(sums.get("SQ_WAIT_INST_ANY", 0) / max(sums.get("SQ_WAVE_CYCLES", 1), 1), 3))
PY
done
