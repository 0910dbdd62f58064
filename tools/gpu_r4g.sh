#!/bin/bash
# round 4: the compact plan record in both row loops: parity, POA microbench dense and sparse, whole step
TAG=${1:-r4g}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
out=$(timeout 120 python tools/bench_poa_class.py 980 64 12 0.10 1 2>&1 | tail -1); echo "canary: $out" | tee $O/canary.log
case "$out" in iter*) ;; *) exit 1;; esac
timeout 900 python -m pytest tests/test_gpu_poa.py tests/test_gpu_correct.py tests/test_gpu_edges.py -x -q -m gpu > $O/tests.log 2>&1; echo "parity: $(tail -1 $O/tests.log)" | tee -a $O/canary.log
for cfg in "980 2560 dense" "1450 2560 dense" "980 256 auto" "1450 256 auto" "980 1 auto"; do
  set -- $cfg
  M=""; [ $3 = dense ] && M=dense
  echo "== len $1 packs $2 $3: $(RATTLE_POA_MODE=$M RATTLE_TIMING=1 timeout 200 python tools/bench_poa_class.py $1 $2 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr '\n' ' ' | sed 's/\[rattle\]     poa class//')"
done 2>&1 | tee $O/micro.log
timeout 900 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --reads 100000 > $O/bench100k.json 2> $O/bench100k.err
python - <<PY
import json
for f in ('bench','bench100k'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step']), {k: round(v) for k,v in d['phases_ms_per_step'].items()}, round(d['kernels_ms_per_step']['poa_align']), round(d['roofline']['gcups']), d['checks']['correct_digest'])
    except Exception as e: print(f, 'failed', e)
PY
