#!/bin/bash
# What holds kernel A (bv_filter) at a third of its issue bound?  One SQ counter group per rocprofv3 pass (nine counters in one
# pass did not finish in round 3) over the gene-level `cluster` of the bench's batch, summed per kernel.
# usage: tools/gpu_pmc_cluster.sh TAG [reads]
TAG=${1:-pmc_cluster}; R=${2:-1000000}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $GRAFT_REPO_ROOT/$O/g$i -- python $GRAFT_REPO_ROOT/tools/cluster_only.py $R 1 > $GRAFT_REPO_ROOT/$O/g$i.log 2>&1 )
  tail -1 $O/g$i.log
done
python - <<PY
import csv, glob, json
sums = {}
for f in glob.glob('$O/g*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        e = sums.setdefault(k, {})
        e[row['Counter_Name']] = e.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
        e['dispatches_' + row['Counter_Name']] = e.get('dispatches_' + row['Counter_Name'], 0) + 1
json.dump(sums, open('$O/pmc_cluster.json', 'w'), indent=1)
for k, e in sums.items():
    if 'bv_filter' in k or 'pair_' in k or 'kmer' in k:
        wc = e.get('SQ_WAVE_CYCLES', 0) or 1
        print(k, {c: round(v / wc, 4) if c.startswith('SQ_') and c != 'SQ_WAVE_CYCLES' else v for c, v in e.items() if not c.startswith('dispatches')})
PY
