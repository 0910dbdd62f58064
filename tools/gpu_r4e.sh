#!/bin/bash
# round 4, fifth GPU call: ONE wavefront per pack (16 / 24 columns per lane, ready-made ring, no mailbox at all) for sparse passes:
# parity, microbench at 256 packs and a lone pack, 1e5 reads; config 5 at 1e5 mixed reads; CLI end to end.
TAG=${1:-r4e}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
out=$(RATTLE_POA_EXP=4,3,0,0 timeout 120 python tools/bench_poa_class.py 980 64 12 0.10 1 2>&1 | tail -1); echo "canary: $out" | tee $O/canary.log
RATTLE_POA_EXP=4,3,-1,-1 timeout 600 python -m pytest tests/test_gpu_poa.py tests/test_gpu_correct.py -x -q -m gpu > $O/tests_1wave.log 2>&1; echo "parity 1-wave: $(tail -1 $O/tests_1wave.log)" | tee -a $O/canary.log
for cfg in "980 256 4,-1,-1,-1" "980 1 4,-1,-1,-1" "1450 256 -1,3,-1,-1" "1450 1 -1,3,-1,-1" "1450 1 -1,-1,-1,-1" "980 2560 4,-1,-1,-1"; do
  set -- $cfg
  echo "== len $1 packs $2 EXP=$3: $(RATTLE_POA_EXP=$3 RATTLE_TIMING=1 timeout 200 python tools/bench_poa_class.py $1 $2 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr '\n' ' ' | sed 's/\[rattle\]     poa class//')"
done 2>&1 | tee $O/micro.log
for exp in "-1,-1,-1,-1" "4,3,-1,-1"; do
  RATTLE_POA_EXP=$exp timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --reads 100000 > $O/bench100k_$exp.json 2> $O/bench100k_$exp.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench100k_$exp.json').read().strip().splitlines()[-1])
    print('100k EXP=$exp', round(d['value']), round(d['ms_per_step']), {k: round(v) for k,v in d['phases_ms_per_step'].items()}, round(d['kernels_ms_per_step']['poa_align']), round(d['roofline']['gcups']), d['checks']['correct_digest'])
except Exception as e: print('bench 100k $exp failed', e)
PY
done 2>&1 | tee $O/bench_100k.log
timeout 900 python tools/run_mixed.py 100000 20000 > $O/mixed_100k.log 2> $O/mixed_100k.err; tail -1 $O/mixed_100k.log | cut -c1-900
RATTLE_TIMING=1 timeout 900 bash tools/cli_e2e.sh 1000000 > $O/cli_e2e.txt 2>&1; tail -30 $O/cli_e2e.txt
