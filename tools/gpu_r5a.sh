#!/bin/bash
# Round 5, first contact of dp_rows_mt (teams of wavefronts) with the device: a canary per form (a hang must cost a minute, not the
# call), parity of every form, then the POA microbench: lone pack / one per CU / three per CU / device full.  usage: tools/gpu_r5a.sh TAG
TAG=${1:-r5a}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for mode in mt1 mt2 mt4 sparse dense; do
  out=$(RATTLE_POA_MODE=$mode timeout 120 python tools/bench_poa_class.py 980 64 12 0.10 1 2>&1 | tail -1)
  echo "canary $mode: $out"
  case "$out" in iter*) ;; *) echo "canary failed: $mode"; RATTLE_POA_MODE=$mode timeout 120 python tools/bench_poa_class.py 980 64 12 0.10 1 2>&1 | tail -5;; esac
done 2>&1 | tee $O/canary.log
if grep -q "canary failed" $O/canary.log; then echo "stop: canary"; exit 1; fi
timeout 1500 python -m pytest tests/test_gpu_poa.py -x -q -m gpu > $O/tests_poa.log 2>&1; echo "poa tests: $(tail -1 $O/tests_poa.log)"
timeout 900 python -m pytest tests/test_gpu_correct.py -x -q -m gpu > $O/tests_correct.log 2>&1; echo "correct tests: $(tail -1 $O/tests_correct.log)"
for packs in 1 256 768 2560; do
  for mode in dense sparse mt1 mt2 mt4; do
    if [ $packs = 2560 ] && [ $mode = mt4 ]; then continue; fi
    echo "== 1024 class, packs $packs, $mode: $(RATTLE_POA_MODE=$mode RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py 980 $packs 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr "\n" " " | sed 's/\[rattle\]     poa class//')"
  done
done 2>&1 | tee $O/micro_1024.log
for packs in 1 256 768; do
  for mode in dense sparse mt1 mt2 mt4; do
    echo "== 1536 class, packs $packs, $mode: $(RATTLE_POA_MODE=$mode RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py 1450 $packs 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr "\n" " " | sed 's/\[rattle\]     poa class//')"
  done
done 2>&1 | tee $O/micro_1536.log
# where a lone pack's time goes (instrumented build): phases per form
for mode in sparse mt4; do
  echo "== lone pack phases, $mode:"; RATTLE_HIP_LIB=$PWD/rattle_amd/csrc/librattle_hip_prof.so RATTLE_POA_MODE=$mode timeout 200 python tools/bench_poa_class.py 980 1 200 0.10 2 2>&1 | grep -E "phases|profile|^iter" | tail -4
done 2>&1 | tee $O/lone_phases.log
