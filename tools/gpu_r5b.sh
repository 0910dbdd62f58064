#!/bin/bash
# Round 5: dp_rows_mt after a change -- canary, POA parity, microbench (lone pack / one per CU / three per CU).  usage: tools/gpu_r5b.sh TAG [full]
TAG=${1:-r5b}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for mode in mt1 mt2 mt4; do
  out=$(RATTLE_POA_MODE=$mode timeout 120 python tools/bench_poa_class.py 980 64 12 0.10 1 2>&1 | tail -1)
  echo "canary $mode: $out"
  case "$out" in iter*) ;; *) echo "canary failed: $mode";; esac
done 2>&1 | tee $O/canary.log
if grep -q "canary failed" $O/canary.log; then echo "stop: canary"; exit 1; fi
timeout 1500 python -m pytest tests/test_gpu_poa.py -x -q -m gpu > $O/tests_poa.log 2>&1; echo "poa tests: $(tail -1 $O/tests_poa.log)"
if [ "$2" = full ]; then timeout 900 python -m pytest tests/test_gpu_correct.py -x -q -m gpu > $O/tests_correct.log 2>&1; echo "correct tests: $(tail -1 $O/tests_correct.log)"; fi
for packs in 1 256 768; do
  for mode in sparse mt1 mt2 mt4; do
    if [ $packs = 768 ] && [ $mode = mt4 ]; then continue; fi
    echo "== 1024 class, packs $packs, $mode: $(RATTLE_POA_MODE=$mode RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py 980 $packs 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr "\n" " " | sed 's/\[rattle\]     poa class//')"
  done
done 2>&1 | tee $O/micro_1024.log
for packs in 1 256; do
  for mode in mt1 mt4; do
    echo "== 1536 class, packs $packs, $mode: $(RATTLE_POA_MODE=$mode RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py 1450 $packs 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr "\n" " " | sed 's/\[rattle\]     poa class//')"
  done
done 2>&1 | tee $O/micro_1536.log
for mode in mt1 mt4; do
  echo "== lone pack phases, $mode:"; RATTLE_HIP_LIB=$PWD/rattle_amd/csrc/librattle_hip_prof.so RATTLE_POA_MODE=$mode timeout 200 python tools/bench_poa_class.py 980 1 200 0.10 2 2>&1 | grep -E "phases|profile|^iter" | tail -4
done 2>&1 | tee $O/lone_phases.log
