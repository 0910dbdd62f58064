#!/bin/bash
# GPU suite + smoke + default bench on the tree as pushed.  usage: tools/gpu_check.sh TAG [bench args]
TAG=${1:-check}; shift
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -n "passed\|failed\|error" $O/tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python bench.py "$@" > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
python - <<PY
import json
try:
    d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
    print(round(d['value']), d['ms_per_step'], d.get('phases_ms_per_step'), d['kernels_ms_per_step'], d['roofline'].get('gcups'))
    for k,v in (d.get('configs') or {}).items(): print(k, {x: v.get(x) for x in ('value','ms_per_step','phases_ms_per_step','error')}, (v.get('roofline') or {}).get('gcups'), (v.get('roofline') or {}).get('frac'))
    print('toyset', d.get('toyset'))
except Exception as e: print('bench failed', e)
PY
