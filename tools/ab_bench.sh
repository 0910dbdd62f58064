#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by +-5 %): usage tools/ab_bench.sh OLD.so [rounds] -- alternates old / new default benches
OLD=$1; R=${2:-2}
for i in $(seq $R); do
  for L in $OLD $PWD/rattle_amd/csrc/librattle_hip.so; do
    RATTLE_HIP_LIB=$L python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $L)', round(d['value']), round(d['roofline']['gcups'],1), d['phases_ms_per_step'], d['checks']['correct_digest'])"
  done
done
