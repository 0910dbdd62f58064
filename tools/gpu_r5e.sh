#!/bin/bash
# Round 5: how a waiting row polls (variants of dp_rows_mt), and where the forms cross over.  usage: tools/gpu_r5e.sh TAG
TAG=${1:-r5e}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
run() { echo "== $1 packs $2 $3: $(RATTLE_HIP_LIB=$LIB RATTLE_POA_MODE=$3 timeout 300 python tools/bench_poa_class.py $1 $2 200 0.10 2 2>&1 | grep -E "^iter|rror" | tail -1)"; }
for v in A pB pC pD pE; do
  if [ $v = A ]; then LIB=$PWD/rattle_amd/csrc/librattle_hip.so; else LIB=$PWD/rattle_amd/csrc/variants/librattle_hip_$v.so; fi
  echo "#### variant $v"
  run 980 1 mt2; run 980 1 mt4; run 980 256 mt2; run 980 256 mt4; run 980 768 mt2; run 1450 512 mt2
done 2>&1 | tee $O/poll_variants.log
LIB=$PWD/rattle_amd/csrc/librattle_hip.so
for packs in 1024 1280; do for mode in dense sparse mt1 mt2; do run 980 $packs $mode; done; done 2>&1 | tee $O/crossover.log
RATTLE_TIMING=1 timeout 600 python bench.py --no-cpu-baseline --no-configs --steps 2 --warmup 1 --reads 100000 > $O/bench_100k.json 2> $O/bench_100k.err; tail -1 $O/bench_100k.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1e5:', round(d['value']), d['ms_per_step'], d.get('phases_ms_per_step'), d['roofline'].get('gcups'))"
grep -E "correct: stage|poa class" $O/bench_100k.err | tail -8 | cut -c1-180
