#!/bin/bash
TAG=${1:-r5g}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
run() { echo "== $1 packs $2 $3: $(RATTLE_HIP_LIB=$LIB RATTLE_POA_MODE=$3 RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py $1 $2 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter|rror" | tail -2 | tr "\n" " " | sed 's/\[rattle\]     poa class//' | cut -c1-230)"; }
timeout 900 python -m pytest tests/test_gpu_poa.py -x -q -m gpu --timeout 150 > $O/tests.log 2>&1; echo "poa tests: $(tail -1 $O/tests.log)"; grep -n "Error\|FAILED\|Timeout" $O/tests.log | head
for v in A w6; do
  if [ $v = A ]; then LIB=$PWD/rattle_amd/csrc/librattle_hip.so; else LIB=$PWD/rattle_amd/csrc/variants/librattle_hip_$v.so; fi
  echo "#### variant $v"
  run 980 1 mt2; run 980 256 mt2; run 980 512 mt2; run 980 768 mt2; run 1450 256 mt2; run 1450 512 mt2; run 1450 768 mt2
done 2>&1 | tee $O/mt2_regs.log
LIB=$PWD/rattle_amd/csrc/librattle_hip.so
run 980 1 mt4; run 980 256 mt4; run 980 1 mt1; run 980 768 mt1; run 980 1024 mt1; run 1450 1 mt4; run 1450 256 mt4; run 1450 768 mt1; run 1900 1 mt4; run 1900 256 mt2; run 1900 1 sparse; run 2400 1 mt4; run 2400 1 sparse
