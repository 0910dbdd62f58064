#!/bin/bash
# interleaved A/B on ONE box of dp_rows_v3 with (pf1) and without (pf0) the far in-edge of the next row requested a row ahead
# (tools/build_variant.sh pf0 -DPOA_FAR_PREFETCH=0 ; tools/build_variant.sh pf1 -DPOA_FAR_PREFETCH=1).  usage: tools/ab_far_prefetch.sh TAG
TAG=${1:-ab_pf}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/rattle_amd/csrc/variants
RATTLE_HIP_LIB=$V/librattle_hip_pf1.so RATTLE_POA_MODE=dense timeout 900 python -m pytest tests/test_gpu_poa.py -m gpu -q -x -k "not band" > $O/tests_pf1.log 2>&1; tail -1 $O/tests_pf1.log
for rep in 1 2; do for n in pf0 pf1; do
  echo -n "$n class 1536, 2048 packs: "; RATTLE_HIP_LIB=$V/librattle_hip_$n.so RATTLE_POA_MODE=dense python tools/bench_poa_class.py 1450 2048 200 2>&1 | grep GCUPS | tail -1
done; done | tee $O/micro.log
for rep in 1 2; do for n in pf0 pf1; do
  RATTLE_HIP_LIB=$V/librattle_hip_$n.so timeout 600 python bench.py --steps 2 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', round(d['value']), d['step_ms'], [x['correct_stage_ms']['1'] for x in d['step_detail']], d['checks']['correct_digest'])"
done; done | tee $O/bench.log
