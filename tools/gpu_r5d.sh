#!/bin/bash
# Round 5: dp_rows_mt with publish-first order: parity (POA + correct suites, every test bounded), microbench, the bench at 1e5 and 1e6 reads.  usage: tools/gpu_r5d.sh TAG
TAG=${1:-r5d}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_poa.py tests/test_gpu_correct.py -x -q -m gpu --timeout 150 > $O/tests.log 2>&1; echo "poa + correct tests: $(tail -1 $O/tests.log)"; grep -n "Error\|FAILED\|Timeout" $O/tests.log | head
for packs in 1 256 768; do
  for mode in sparse mt1 mt2 mt4 mt4w; do
    if [ $packs = 768 ] && [ ${mode:0:3} = mt4 ]; then continue; fi
    echo "== 1024 class, packs $packs, $mode: $(RATTLE_POA_MODE=$mode RATTLE_TIMING=1 timeout 300 python tools/bench_poa_class.py 980 $packs 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter|rror" | tail -2 | tr "\n" " " | sed 's/\[rattle\]     poa class//')"
  done
done 2>&1 | tee $O/micro_1024.log
for mode in mt1 mt4 mt4w; do
  echo "== lone pack phases, $mode:"; RATTLE_HIP_LIB=$PWD/rattle_amd/csrc/librattle_hip_prof.so RATTLE_POA_MODE=$mode timeout 200 python tools/bench_poa_class.py 980 1 200 0.10 2 2>&1 | grep -E "phases|profile|^iter" | tail -3
done 2>&1 | tee $O/lone_phases.log
echo "== chain-like packs (40 reads at 2 % error: POA #2 / #3 look like this), lone pack:"
for mode in sparse mt1 mt4; do
  echo "$mode: $(RATTLE_POA_MODE=$mode timeout 200 python tools/bench_poa_class.py 980 1 200 0.02 2 2>&1 | grep -E "^iter" | tail -1)"
done 2>&1 | tee $O/chain.log
RATTLE_TIMING=1 timeout 600 python bench.py --no-cpu-baseline --no-configs --steps 2 --warmup 1 --reads 100000 > $O/bench_100k.json 2> $O/bench_100k.err; tail -1 $O/bench_100k.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1e5:', round(d['value']), d['ms_per_step'], d.get('phases_ms_per_step'), d['roofline'].get('gcups'))"
RATTLE_TIMING=1 timeout 900 python bench.py --no-cpu-baseline --no-configs --steps 3 --warmup 1 > $O/bench_1M.json 2> $O/bench_1M.err; tail -1 $O/bench_1M.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1e6:', round(d['value']), d['ms_per_step'], d.get('phases_ms_per_step'), d['roofline'].get('gcups'))"
grep -E "stage|poa class|poa pass" $O/bench_1M.err | tail -40
