#!/bin/bash
# The PMC passes (HBM read, HBM write, instruction mix, what the waves wait for) of tools/gpu_profile_round2.sh alone (counters only, one group per run), summarised into OUT/pmc_poa.json
# usage: tools/gpu_pmc_only.sh TAG
TAG=${1:-pmc}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=${READS_PMC:-300000}
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  N=$(echo $G | cut -d' ' -f1)
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$N -- python $GRAFT_REPO_ROOT/bench.py --reads $R --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_$N.json 2> $GRAFT_REPO_ROOT/$O/pmc_$N.err )
done
python tools/pmc_summary.py $O/pmc_poa.json $O/pmc_FETCH_SIZE.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU $O/pmc_SQ_WAIT_INST_LDS > $O/pmc_summary.log 2>&1
python -c "
import json; d=json.load(open('$O/pmc_poa.json')); print({k:v for k,v in d.items() if k not in ('counters_by_kernel','workload','note')})"
