#!/bin/bash
# round 4, second GPU call: canaries of the skewed-pipeline variants, the full GPU suite on the defaults (new correct flow with the
# big clusters' chain beside stage 1, world-8 tests), bench with and without the overlap, then the variants' parity + microbench.
TAG=${1:-r4b}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for exp in 0 1 2 3 4 5; do
  out=$(RATTLE_POA_EXP=$exp,$exp,0,0 timeout 90 python tools/bench_poa_class.py 980 64 12 0.10 1 2>&1 | tail -1)
  echo "canary EXP=$exp: $out"
done 2>&1 | tee $O/canary.log
timeout 2700 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed\|error" $O/tests.log | tail -3
for ov in 1 0 1 0; do
  RATTLE_CORRECT_OVERLAP=$ov RATTLE_TIMING=1 timeout 600 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > $O/bench_ov$ov.json 2> $O/bench_ov$ov.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_ov$ov.json').read().strip().splitlines()[-1])
    print('overlap=$ov', round(d['value']), round(d['ms_per_step']), {k: round(v) for k,v in d['phases_ms_per_step'].items()}, {k: round(v) for k,v in d['kernels_ms_per_step'].items()}, round(d['roofline']['gcups']), d['checks']['correct_digest'])
except Exception as e: print('bench overlap=$ov failed', e)
PY
  grep "correct: stage\|correct: total" $O/bench_ov$ov.err | tail -8
done 2>&1 | tee $O/bench_overlap.log
if grep -q "canary.*iter" $O/canary.log && ! grep -q "canary EXP=[0-9]: *$" $O/canary.log; then
for exp in "0,0,0,0" "1,1,1,1" "3,3,2,2" "4,4,1,2" "5,5,0,1" "2,2,0,0"; do
  RATTLE_POA_EXP=$exp timeout 420 python -m pytest tests/test_gpu_poa.py tests/test_gpu_correct.py -x -q -m gpu > $O/tests_$exp.log 2>&1
  echo "parity EXP=$exp: $(tail -1 $O/tests_$exp.log)"
done 2>&1 | tee $O/parity.log
for packs in 2560 256 1; do
  for exp in "-1" "0" "1" "2" "3" "4" "5"; do
    echo "== 1024 class, packs $packs, EXP=$exp: $(RATTLE_POA_EXP=$exp,-1,-1,-1 RATTLE_TIMING=1 timeout 240 python tools/bench_poa_class.py 980 $packs 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr '\n' ' ')"
  done
done 2>&1 | tee $O/micro_1024.log
for packs in 2560 256; do
  for exp in "-1" "0" "1" "2" "3" "4" "5"; do
    echo "== 1536 class, packs $packs, EXP=$exp: $(RATTLE_POA_EXP=-1,$exp,-1,-1 RATTLE_TIMING=1 timeout 240 python tools/bench_poa_class.py 1450 $packs 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr '\n' ' ')"
  done
done 2>&1 | tee $O/micro_1536.log
fi
