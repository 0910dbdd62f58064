#!/bin/bash
TAG=${1:-r4k}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_poa.py tests/test_gpu_correct.py tests/test_gpu_edges.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests: $(tail -1 $O/tests.log)"
bash tools/ab.sh $TAG 3 compact1 base
for cfg in "980 2560 dense" "1450 2560 dense" "980 1 auto"; do
  set -- $cfg
  M=""; [ $3 = dense ] && M=dense
  echo "== len $1 packs $2 $3: $(RATTLE_POA_MODE=$M timeout 200 python tools/bench_poa_class.py $1 $2 200 0.10 2 2>&1 | grep -E "^iter" | tail -1)"
done 2>&1 | tee $O/micro.log
