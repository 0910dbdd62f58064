#!/bin/bash
# interleaved A/B of library variants on ONE box: tools/ab.sh TAG REPS name1 name2 ... ("base" = the in-tree library)
TAG=$1; REPS=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
V=$PWD/rattle_amd/csrc/variants
for rep in $(seq $REPS); do
for n in "$@"; do
  L=$V/librattle_hip_$n.so; [ $n = base ] && L=$PWD/rattle_amd/csrc/librattle_hip.so
  RATTLE_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', round(d['value']), round(d['roofline']['gcups'],1), {k: round(v) for k, v in d['phases_ms_per_step'].items()}, round(d['kernels_ms_per_step']['poa_align']), d['checks']['correct_digest'])" | tee -a $O/ab.log
done; done
