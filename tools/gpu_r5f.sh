#!/bin/bash
# Round 5: dp_rows_mt without the zero slot and without the 80-register cap: parity, forms by load, the bench at 1e5.  usage: tools/gpu_r5f.sh TAG
TAG=${1:-r5f}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_poa.py -x -q -m gpu --timeout 150 > $O/tests.log 2>&1; echo "poa tests: $(tail -1 $O/tests.log)"; grep -n "Error\|FAILED\|Timeout" $O/tests.log | head
run() { echo "== $1 packs $2 $3: $(RATTLE_HIP_LIB=$LIB RATTLE_POA_MODE=$3 RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py $1 $2 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter|rror" | tail -2 | tr "\n" " " | sed 's/\[rattle\]     poa class//' | cut -c1-230)"; }
for v in A pF; do
  if [ $v = A ]; then LIB=$PWD/rattle_amd/csrc/librattle_hip.so; else LIB=$PWD/rattle_amd/csrc/variants/librattle_hip_$v.so; fi
  echo "#### variant $v"
  run 980 1 mt2; run 980 256 mt2; run 980 512 mt2; run 1450 512 mt2
done 2>&1 | tee $O/poll_variants.log
LIB=$PWD/rattle_amd/csrc/librattle_hip.so
for packs in 1 256 512 768 1024; do for mode in dense sparse mt1 mt2 mt4; do
  if [ $packs -gt 300 ] && [ $mode = mt4 ]; then continue; fi
  run 980 $packs $mode; done; done 2>&1 | tee $O/crossover_1024.log
for packs in 256 512 768 1024; do for mode in dense sparse mt1 mt2; do run 1450 $packs $mode; done; done 2>&1 | tee $O/crossover_1536.log
RATTLE_TIMING=1 timeout 600 python bench.py --no-cpu-baseline --no-configs --steps 2 --warmup 1 --reads 100000 > $O/bench_100k.json 2> $O/bench_100k.err; tail -1 $O/bench_100k.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1e5:', round(d['value']), d['ms_per_step'], d.get('phases_ms_per_step'), d['roofline'].get('gcups'))"
grep -E "correct: stage|poa class" $O/bench_100k.err | tail -8 | cut -c1-180
