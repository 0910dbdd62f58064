#!/bin/bash
# round 4, fourth GPU call: regression check of the dense forms after the LDS layout went back, SQ / instruction-cache counters of the
# POA microbench (barrier form vs skewed pipeline with ready-made terms), kernel A after the 16-dword banks, default bench.
TAG=${1:-r4d}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for cfg in "980 -1,-1,-1,-1" "1450 -1,-1,-1,-1" "980 1,-1,-1,-1" "1450 -1,1,-1,-1"; do
  set -- $cfg
  echo "== len $1 EXP=$2 dense: $(RATTLE_POA_MODE=dense RATTLE_POA_EXP=$2 RATTLE_TIMING=1 timeout 200 python tools/bench_poa_class.py $1 2560 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr '\n' ' ' | sed 's/\[rattle\]     poa class//')"
done 2>&1 | tee $O/micro.log
for cfg in "980 256" "1450 256" "980 1"; do
  set -- $cfg
  echo "== len $1 packs $2 auto: $(RATTLE_TIMING=1 timeout 200 python tools/bench_poa_class.py $1 $2 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr '\n' ' ' | sed 's/\[rattle\]     poa class//')"
done 2>&1 | tee -a $O/micro.log
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INSTS_[A-Z_]*" | sort -u | tr '\n' ' ' > $O/counters_avail.txt
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  for v in "-1" "1"; do
    ( cd /tmp && RATTLE_POA_MODE=dense RATTLE_POA_EXP=$v,-1,-1,-1 timeout 240 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_g${i}_v$v -- python $GRAFT_REPO_ROOT/tools/bench_poa_class.py 980 2560 200 0.10 1 > $GRAFT_REPO_ROOT/$O/pmc_g${i}_v$v.log 2>&1 )
  done
done
python - <<PY
import csv, glob, os, re
for v in ("-1", "1"):
    sums = {}
    cells = None
    for d in sorted(glob.glob("$O/pmc_g*_v%s" % v)):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "poa_kernel" not in row["Kernel_Name"]: continue
                sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        m = re.search(r"kernel (\d+) ms\s+([\d.]+) GCUPS", open(d + ".log").read())
        if m: cells = float(m.group(1)) * 1e-3 * float(m.group(2)) * 1e9
    wc = sums.get("SQ_WAVE_CYCLES", 1) or 1
    print("variant", v, "cells", cells)
    print("  per cell:", {k: round(x / cells, 4) for k, x in sums.items() if k.startswith(("SQ_INSTS", "SQC_", "SQ_IFETCH"))} if cells else sums)
    print("  of wave cycles:", {k: round(x / wc, 3) for k, x in sums.items() if k.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES", "SQ_BUSY"))})
PY
RATTLE_TIMING=1 timeout 300 python tools/cluster_only.py 1000000 3 2>&1 | grep -E "^run|filter|count pass|full pass|index" | tail -12 | tee $O/cluster_only.log
timeout 1500 python bench.py --steps 2 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
    print(round(d['value']), d['ms_per_step'], d.get('phases_ms_per_step'), d['kernels_ms_per_step'], d['roofline'].get('gcups'))
    for k,v in (d.get('configs') or {}).items(): print(k, {x: v.get(x) for x in ('value','ms_per_step','phases_ms_per_step','error')}, (v.get('roofline') or {}).get('gcups'), (v.get('roofline') or {}).get('frac'))
    print('toyset', {x: d['toyset'].get(x) for x in ('cluster_s','correct_s','reads_per_s','clusters_equal_reference_fixture')})
except Exception as e: print('bench failed', e)
PY
