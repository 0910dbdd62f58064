#!/bin/bash
# sparse threshold (packs per CU below which a pass takes the sparse form): 4 (default) against 7 on the whole step, interleaved; the new POA test
TAG=${1:-r4j}; O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_poa.py -x -q -m gpu -k "predecessors_hundreds or fallback" > $O/tests.log 2>&1; echo "tests: $(tail -1 $O/tests.log)"
for rep in 1 2; do for t in 4 7; do
  RATTLE_POA_SPARSE_PER_CU=$t RATTLE_TIMING=1 timeout 600 python bench.py --no-cpu-baseline 2> $O/err_$t.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sparse<$t/CU', round(d['value']), round(d['roofline']['gcups'],1), {k: round(v) for k, v in d['phases_ms_per_step'].items()}, round(d['kernels_ms_per_step']['poa_align']), d['checks']['correct_digest'])"
  grep "correct: stage" $O/err_$t.txt | tail -4 | tr '\n' ' '; echo
done; done 2>&1 | tee $O/ab.log
