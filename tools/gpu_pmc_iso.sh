#!/bin/bash
# HBM traffic of kernel B in the --iso flow (config 3): two rocprofv3 --pmc passes (counters only, one per run) over the same
# deterministic workload, summarised into OUT/pmc_iso.json.  usage: tools/gpu_pmc_iso.sh TAG
TAG=${1:-pmc_iso}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R=${READS_PMC:-300000}
for G in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$G -- python $GRAFT_REPO_ROOT/bench.py --iso --reads $R --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_$G.json 2> $GRAFT_REPO_ROOT/$O/pmc_$G.err )
done
python tools/pmc_iso_summary.py $O/pmc_iso.json $O/pmc_FETCH_SIZE.json $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
