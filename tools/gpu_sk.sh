#!/bin/bash
# Round 4: the skewed wavefront pipeline of kernel C (dp_rows_sk) against the barrier form (dp_rows_v3): a canary per variant
# (a hang must cost a minute, not the call), parity of every candidate variant (POA + correct suites under RATTLE_POA_EXP),
# then the POA microbench full / one pack per CU / lone pack.   usage: tools/gpu_sk.sh TAG
TAG=${1:-sk}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for exp in 0 1 2 3 4 5; do
  out=$(RATTLE_POA_EXP=$exp,$exp,0,0 timeout 90 python tools/bench_poa_class.py 980 64 12 0.10 1 2>&1 | tail -1)
  echo "canary EXP=$exp: $out"
  case "$out" in iter*) ;; *) echo "canary failed: stop"; exit 1;; esac
done 2>&1 | tee $O/canary.log
grep -q "canary failed" $O/canary.log && exit 1
for exp in "0,0,0,0" "1,1,1,1" "3,3,2,2" "4,4,1,2" "5,2,0,1" "2,0,0,0"; do
  RATTLE_POA_EXP=$exp timeout 420 python -m pytest tests/test_gpu_poa.py tests/test_gpu_correct.py -x -q -m gpu > $O/tests_$exp.log 2>&1
  echo "parity EXP=$exp: $(tail -1 $O/tests_$exp.log)"
done 2>&1 | tee $O/parity.log
for packs in 2560 256 1; do
  for exp in "-1" "0" "1" "2" "3" "4" "5"; do
    echo "== 1024 class, packs $packs, EXP=$exp: $(RATTLE_POA_EXP=$exp,-1,-1,-1 RATTLE_TIMING=1 timeout 240 python tools/bench_poa_class.py 980 $packs 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr "\n" " ")"
  done
done 2>&1 | tee $O/micro_1024.log
for packs in 2560 256; do
  for exp in "-1" "0" "1" "2" "3" "4" "5"; do
    echo "== 1536 class, packs $packs, EXP=$exp: $(RATTLE_POA_EXP=-1,$exp,-1,-1 RATTLE_TIMING=1 timeout 240 python tools/bench_poa_class.py 1450 $packs 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr "\n" " ")"
  done
done 2>&1 | tee $O/micro_1536.log
