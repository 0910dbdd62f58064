#!/bin/bash
# round 4, third GPU call: the skewed-pipeline variants with the mailbox in real LDS instructions (the first build's volatile generic
# pointers had become flat loads / stores with vmcnt(0) waits): POA microbench full / one pack per CU / lone pack, then the whole
# step with candidate defaults, then 1e5 reads dense vs sparse.
TAG=${1:-r4c}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for exp in 0 1 3; do
  out=$(RATTLE_POA_EXP=$exp,$exp,0,0 timeout 90 python tools/bench_poa_class.py 980 64 12 0.10 1 2>&1 | tail -1); echo "canary EXP=$exp: $out"
  case "$out" in iter*) ;; *) echo "canary failed"; exit 1;; esac
done 2>&1 | tee $O/canary.log
grep -q "canary failed" $O/canary.log && exit 1
for packs in 2560 256 1; do
  for exp in "-1" "0" "1" "2" "3" "5"; do
    echo "== 1024 class, packs $packs, EXP=$exp: $(RATTLE_POA_EXP=$exp,-1,-1,-1 RATTLE_TIMING=1 timeout 200 python tools/bench_poa_class.py 980 $packs 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr '\n' ' ' | sed 's/\[rattle\]     poa class//')"
  done
done 2>&1 | tee $O/micro_1024.log
for packs in 2560 256; do
  for exp in "-1" "0" "1" "2" "4" "5"; do
    echo "== 1536 class, packs $packs, EXP=$exp: $(RATTLE_POA_EXP=-1,$exp,-1,-1 RATTLE_TIMING=1 timeout 200 python tools/bench_poa_class.py 1450 $packs 200 0.10 2 2>&1 | grep -E "blocks/CU|^iter" | tail -2 | tr '\n' ' ' | sed 's/\[rattle\]     poa class//')"
  done
done 2>&1 | tee $O/micro_1536.log
for exp in "-1,-1,-1,-1" "0,0,0,0" "1,5,1,1" "0,4,0,0" "1,1,1,1"; do
  RATTLE_POA_EXP=$exp RATTLE_TIMING=1 timeout 400 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > $O/bench_$exp.json 2> $O/bench_$exp.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$exp.json').read().strip().splitlines()[-1])
    print('EXP=$exp', round(d['value']), round(d['ms_per_step']), {k: round(v) for k,v in d['phases_ms_per_step'].items()}, round(d['kernels_ms_per_step']['poa_align']), round(d['roofline']['gcups']), d['checks']['correct_digest'])
except Exception as e: print('bench EXP=$exp failed', e)
PY
  grep "correct: stage" $O/bench_$exp.err | tail -4 | tr '\n' ' '; echo
done 2>&1 | tee $O/bench_exp.log
for mode in dense sparse; do
  RATTLE_POA_MODE=$mode timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --reads 100000 > $O/bench100k_$mode.json 2> $O/bench100k_$mode.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/bench100k_$mode.json').read().strip().splitlines()[-1])
    print('100k $mode', round(d['value']), round(d['ms_per_step']), {k: round(v) for k,v in d['phases_ms_per_step'].items()}, round(d['kernels_ms_per_step']['poa_align']), round(d['roofline']['gcups']), d['checks']['correct_digest'])
except Exception as e: print('bench 100k $mode failed', e)
PY
done 2>&1 | tee $O/bench_100k.log
